import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for name, typ in c.execute("select name, type from sqlite_master where type in ('table','view') order by type, name"):
    cols = [r[1] for r in c.execute(f"pragma table_info('{name}')")]
    try:
        n = c.execute(f"select count(*) from '{name}'").fetchone()[0]
    except Exception as e:
        n = str(e)
    print(typ, name, n, cols)
for q in ("select * from regions limit 5", "select * from markers limit 5", "select * from kernels limit 2", "select * from regions_and_samples limit 5"):
    try:
        print(q); [print("   ", r) for r in c.execute(q).fetchall()]
    except Exception as e:
        print("   ERR", e)
