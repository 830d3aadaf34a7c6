#!/usr/bin/env python3
"""Per-kernel roofline numbers of the bench step (north_star: "evidenced by rocprof HBM GB/s and MFMA utilisation against gfx950 peak"),
as NUMBERS, from four rocprofv3 runs of the same bench.py command:
    trace.db   --kernel-trace only                      -> launches per step, average duration (un-perturbed by counters)
    fetch.db   --kernel-trace --pmc FETCH_SIZE          -> bytes fetched from HBM (x2 on gfx950: 64 B tallied per 128-B request)
    write.db   --kernel-trace --pmc WRITE_SIZE          -> bytes written
    sq.db      --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES  -> matrix-pipe busy cycles (summed over the 1024 SIMDs)
usage: kernel_roofline.py trace.db fetch.db write.db sq.db steps_in_run [out.txt]
HBM GB/s = (2 FETCH + WRITE) / duration against 8000 GB/s; MFMA utilisation = MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs), both
with the duration of the dispatch the counter was read on (a counter pass slows a dispatch; rates within one pass are consistent)."""
import os
import sqlite3
import sys

HBM_PEAK_GBS, CLK_HZ, SIMDS = 8000.0, 2.4e9, 1024


def short(kn):
    return kn.split("(")[0].replace("void ", "")


def pmc(db, names):
    c = sqlite3.connect(db)
    out = {}
    q = "select kernel_name, counter_name, sum(value), count(*), sum(duration) from counters_collection group by kernel_name, counter_name"
    for kn, cn, v, n, dur in c.execute(q):
        if cn in names:
            d = out.setdefault(short(kn), {})
            d[cn] = d.get(cn, 0.0) + v
            d["n_" + cn] = d.get("n_" + cn, 0) + n
            d["dur_" + cn] = d.get("dur_" + cn, 0.0) + dur
    return out


def main():
    trace, fdb, wdb, sdb, steps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
    c = sqlite3.connect(trace)
    base = {}
    for kn, s, e in c.execute("select name, start, end from kernels"):
        d = base.setdefault(short(kn), [0, 0.0])
        d[0] += 1
        d[1] += e - s
    F, W, S = pmc(fdb, {"FETCH_SIZE"}), pmc(wdb, {"WRITE_SIZE"}), pmc(sdb, {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"})
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from csrc_hash import csrc_hash
    lines = [f"# per-kernel roofline numbers of `python bench.py` (ViT-B/16 MsLdCeCd, 224^2, N = 128, bf16), {steps} steps per run; kernel sources sha256 {csrc_hash()[:16]}",
             "# HBM: (2 x FETCH_SIZE + WRITE_SIZE) / duration of the counted dispatches, peak 8000 GB/s (MI355X_MICROARCH.md; ~6300 achievable)",
             "# MFMA: SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs); durations in the overlapped step (two to three streams share the chip)",
             f"{'kernel':58s} {'n/step':>6s} {'avg_us':>8s} {'ms/step':>8s} {'MB/launch':>10s} {'HBM GB/s':>9s} {'of peak':>8s} {'MFMA util':>9s}"]
    tot_ms = tot_gb = 0.0
    for k, (n, dur) in sorted(base.items(), key=lambda kv: -kv[1][1]):
        if dur / steps < 20e3:      # (< 20 us per step)
            continue
        f, w, s = F.get(k, {}), W.get(k, {}), S.get(k, {})
        nb = max(1, f.get("n_FETCH_SIZE", 0))
        mb = (2.0 * f.get("FETCH_SIZE", 0) / nb + w.get("WRITE_SIZE", 0) / max(1, w.get("n_WRITE_SIZE", 0))) / 1024.0   # counters are in KiB
        durf = f.get("dur_FETCH_SIZE", 0) / nb
        gbs = mb * 2 ** 20 / max(durf, 1.0)   # bytes / ns = GB/s
        ns = max(1, s.get("n_SQ_VALU_MFMA_BUSY_CYCLES", 0))
        durs = s.get("dur_SQ_VALU_MFMA_BUSY_CYCLES", 0) / ns
        util = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / ns / max(durs * 1e-9 * CLK_HZ * SIMDS, 1.0)
        lines.append(f"{k[:58]:58s} {n / steps:6.1f} {dur / n / 1e3:8.1f} {dur / steps / 1e6:8.3f} {mb:10.1f} {gbs:9.0f} {gbs / HBM_PEAK_GBS:8.1%} {util:9.1%}")
        tot_ms += dur / steps / 1e6
        tot_gb += mb * n / steps / 1024.0
    lines.append(f"# listed kernels: {tot_ms:.2f} ms of kernel time and {tot_gb:.1f} GB of HBM traffic per step")
    txt = "\n".join(lines) + "\n"
    print(txt)
    if len(sys.argv) > 6:
        open(sys.argv[6], "w").write(txt)


if __name__ == "__main__":
    main()
