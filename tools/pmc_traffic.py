#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each with --kernel-trace) of bench.py.
usage: pmc_traffic.py <fetch.db> <write.db> <steps_in_run> [out.txt] [out.json]
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 64 B per 128-B request -> doubled here; WRITE_SIZE as reported."""
import json, sqlite3, sys
fdb, wdb, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
res = {}
for db, ctr in ((fdb, "FETCH_SIZE"), (wdb, "WRITE_SIZE")):
    c = sqlite3.connect(db)
    for kn, v, n, dur in c.execute("select kernel_name, sum(value), count(*), sum(duration) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
        short = kn.split("(")[0].replace("void ", "")
        d = res.setdefault(short, {})
        d[ctr] = d.get(ctr, 0) + v; d["n_" + ctr] = d.get("n_" + ctr, 0) + n; d["dur_" + ctr] = d.get("dur_" + ctr, 0) + dur
tf = sum(d.get("FETCH_SIZE", 0) for d in res.values()) * 2 / 1e6 / steps
tw = sum(d.get("WRITE_SIZE", 0) for d in res.values()) / 1e6 / steps
lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) over bench.py; per-launch averages in MB",
         "# FETCH_SIZE x 2 (gfx950: 64 B tallied per 128-B request; calibrated on ln_fwd: 103 MB algorithmic vs 2 x 49.3 MB), WRITE_SIZE as reported",
         f"# whole step ({steps} steps incl. settle/warm-up in the run): fetch {tf:.1f} GB + write {tw:.1f} GB = {tf + tw:.1f} GB per step",
         f"{'kernel':70s} {'launches':>8s} {'fetch_MB':>10s} {'write_MB':>10s} {'avg_us':>9s}"]
gem = {"f": 0.0, "w": 0.0, "n": 0}
for k, d in sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) * 2 + kv[1].get("WRITE_SIZE", 0)))[:40]:
    n = d.get("n_FETCH_SIZE", 1)
    f = d.get("FETCH_SIZE", 0) * 2 / n / 1024; w = d.get("WRITE_SIZE", 0) / max(1, d.get("n_WRITE_SIZE", 1)) / 1024
    lines.append(f"{k[:70]:70s} {n:8d} {f:10.2f} {w:10.2f} {d.get('dur_FETCH_SIZE', 0) / n / 1e3:9.1f}")
    if k.startswith("gemm_"):
        gem["f"] += d.get("FETCH_SIZE", 0) * 2 / 1024; gem["w"] += d.get("WRITE_SIZE", 0) / 1024; gem["n"] += n
txt = "\n".join(lines) + "\n"
print(txt)
if len(sys.argv) > 4:
    open(sys.argv[4], "w").write(txt)
if len(sys.argv) > 5:
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_hash   # the profile is valid for exactly these kernel sources (bench.py refuses a stale one)
    json.dump({"step_hbm_gb": round(tf + tw, 2), "gemm_mb_per_launch": round((gem["f"] + gem["w"]) / max(1, gem["n"]), 2), "csrc_sha256": csrc_hash(),
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 (gfx950), " + sys.argv[4]}, open(sys.argv[5], "w"))
