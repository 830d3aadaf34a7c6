import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select id, category, name, start, end, event_id, stack_id, corr_id, extdata from regions where category like 'MARKER%' limit 6").fetchall()
for r in rows: print(r)
ev = [r[5] for r in rows]
for e in ev[:3]:
    print(e, c.execute("select arg_position, arg_type, arg_name, arg_value from events_args where event_id=?", (e,)).fetchall())
print(c.execute("select distinct name from regions where category like 'MARKER%'").fetchall()[:20])
