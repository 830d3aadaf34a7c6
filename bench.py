#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): Cross-Scale MAE pre-training images/s, ViT-B/16 MAE_ViT_MsLdCeCd, 224^2 two-scale crops,
mask 0.75, bf16 MFMA, batch 128 per GPU, synthetic data, full optimizer steps (crop -> 2-view fwd -> 4 loss terms -> bwd ->
grad all-reduce -> AdamW).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` prices the whole step against the dense bf16 MFMA peak (118.80 GFLOP per
image, fwd+bwd) and reports the dominant kernel (the MFMA GEMM family) timed with HIP events on its launch stream.
`cpu_baseline` times the CPU oracle (oracle/, a validated restatement of the reference step) on the host cores — a reported
baseline, never the thing being measured."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))

PEAK_BF16_TFLOPS = 2516.6          # 256 CU x 4096 FLOP/clk/CU x 2.4 GHz (MI355X_MICROARCH.md)
PEAK_HBM_TBS = 8.0                 # HBM3E peak, TB/s (MI355X_MICROARCH.md; ~6.3 achievable)
PEAK_FP8_TFLOPS = 5033.2           # dense fp8 MFMA peak (2 x bf16; MI355X_MICROARCH.md) — the roof BASELINE.json configs[4] is priced against
GFLOP_PER_IMAGE = 118.80           # SURVEY.md §8(d), config 2, fwd + bwd (= 3 x 39.599)
BATCH_PER_GPU, INPUT, PATCH = 128, 224, 16


# the other BASELINE.json configs, as single-GPU slices (`--preset`; never the headline line): name -> (factory, input, patch, channels,
# per-GPU batch, algorithmic GFLOP per image fwd+bwd from SURVEY.md §8d)
PRESETS = {
    "base": ("mae_vit_base_MsLdCeCd", 224, 16, 3, 128, 118.80),
    "large": ("mae_vit_large_MsLdCeCd", 224, 16, 3, 128, 250.15),       # configs[2]: ViT-L/16 224^2, 128 per GPU
    "large4": ("mae_vit_large_MsLdCeCd", 256, 16, 4, 128, 328.20),      # configs[3]: ViT-L/16 256^2 4-band
    "huge14": ("mae_vit_huge_MsLdCeCd", 224, 14, 3, 256, 584.23),       # configs[4] geometry (ViT-H/14, 256 per GPU) with bf16 GEMMs
}


def build(device, batch, world, loss="mse", preset="base"):
    import models_mae
    from csmae_hip.optim import FusedAdamW, add_weight_decay
    from csmae_hip.parallel import DataParallel
    torch.manual_seed(0)
    factory, size, patch, chans = PRESETS[preset][:4]
    model = getattr(models_mae, factory)(input_size=size, patch_size=str(patch), input_channels=chans, loss=loss, loss_cd="mse", mask_ratio=0.75,
                                         device=str(device))
    model.to(device).train()
    model.compute_dtype = torch.bfloat16
    lr = 5e-5 * batch * world / 256  # main_pretrain.py:406-412 (blr 5e-5)
    opt = FusedAdamW(add_weight_decay(model, 0.05), lr=lr, betas=(0.9, 0.95), overlap=True)   # (the step runs under the next forward pass's first layers, as main_pretrain.py runs it)
    wrapped = DataParallel(model) if world > 1 else model
    return model, wrapped, opt


# Algorithmic HBM bytes of one step: every activation / gradient tensor of a transformer block counted once per kernel that
# must read or write it, in units of one [tokens, width] bf16 matrix — forward 28 units per block (LayerNorm 2 + 2, qkv 1 + 3, attention
# 3 + 1, proj 3, fc1 1 + 4 + 2 (h and the 8-bit gelu'), fc2 4 + 2), backward 50 (fc2-dX 7, its dW 5, fc1-dX 5, its dW 5, LayerNorm 4 + 4,
# proj-dX 2, its dW 2, attention 8, qkv-dX 4, its dW 4) — plus AdamW (28 B / parameter), the gradient clear (4 B / parameter) and the
# images (two views read by patch-gather and by both reconstruction kernels).  Operand re-reads across tiles, split-K slabs and the heads /
# stem (< 3 %) are NOT in it: the gap between this figure and the PMC traffic is what the kernels waste.
BLOCK_UNITS = 28 + 50
GEOM = {"base": (768, 12, 512, 8), "large": (1024, 24, 512, 8), "large4": (1024, 24, 512, 8), "huge14": (1280, 32, 512, 8)}   # D, Ne, Dd, Nd


def algorithmic_hbm_gb(preset, batch, n_params):
    _, size, patch, chans = PRESETS[preset][:4]
    D, Ne, Dd, Nd = GEOM[preset]
    L = (size // patch) ** 2
    keep = int(L * 0.25)
    Me, Md = 2 * batch * (keep + 1), 2 * batch * (L + 1)
    acts = BLOCK_UNITS * 2 * (Ne * Me * D + Nd * Md * Dd)
    return (acts + 32 * n_params + 3 * 2 * batch * chans * size * size * 4) / 1e9


def csrc_hash():
    """Content hash of the kernel sources: what a committed PMC profile is valid for."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from csrc_hash import csrc_hash as h
    return h()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(seconds_budget=30.0):
    """The oracle's fwd+bwd+AdamW on the host cores at N=16.  Bounded: one warm-up step, then as many timed
    steps (<= 3) as fit the budget; if the warm-up alone exceeds the budget it is the sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import csmae_oracle as O
    import models_mae
    n = 16
    threads = min(os.cpu_count() or 1, 64)  # beyond ~64 threads torch's CPU GEMMs at these sizes stop scaling
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    m = models_mae.mae_vit_base_MsLdCeCd(input_size=INPUT, patch_size=str(PATCH))
    sd = O.trainable_copy(m.state_dict())
    cfg = O.make_cfg(input_size=INPUT, patch_size=PATCH, variant="MsLdCeCd", **O.PRESETS["base"])
    opt = torch.optim.AdamW(O.adamw_groups(sd.items(), 0.05), lr=1e-4, betas=(0.9, 0.95))
    g = torch.Generator().manual_seed(0)
    imgs = torch.randn(n, 3, INPUT, INPUT, generator=g)
    bn = dict(running_mean=sd["predictor.1.running_mean"], running_var=sd["predictor.1.running_var"], num_batches_tracked=sd["predictor.1.num_batches_tracked"])

    def step():
        box = O.crop_box(INPUT, (0.25, 0.75))
        O.train_step(sd, cfg, imgs, torch.rand(n, 196, generator=g), torch.rand(n, 196, generator=g), box, opt, bn=bn)
    t0 = time.time()
    step()
    warm = time.time() - t0
    log(f"cpu_baseline warm-up step {warm:.1f} s on {threads} threads")
    if warm > seconds_budget:
        k, dt = 1, warm
    else:
        t0, k = time.time(), 0
        while k < 3 and time.time() - t0 + warm < seconds_budget:
            step()
            k += 1
        k, dt = (k, time.time() - t0) if k else (1, warm)
    return dict(value=round(n * k / dt, 3), unit="images/s", cores=threads, kind="port",
                sample=f"oracle/csmae_oracle.py fwd+bwd+AdamW, ViT-B/16 MsLdCeCd 224^2 fp32, N={n}, {k} timed step(s) after warm-up, "
                       f"{threads} torch threads of {os.cpu_count()} host cores")


class ChipWatch:
    """Shader clock and board power over the timed region, sampled from sysfs (hwmon freq1_input / power1_average of this rank's GPU) on a host
    thread: the chip clocks to its power budget (MI355X_MICROARCH.md, DVFS), boxes of the pool differ by a few %, and a line that carries its
    own clock can be told apart from a slower kernel.  Best effort: fields are null where sysfs does not offer them."""

    def __init__(self, device_index):
        import glob
        self.freq = self.power = None
        self.samples = []
        try:
            props = torch.cuda.get_device_properties(device_index)
            want = f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}".lower() if hasattr(props, "pci_bus_id") else None
        except Exception:
            want = None
        cards = []
        for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
            real = os.path.realpath(dev)
            hw = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
            if not hw or not os.path.exists(os.path.join(hw[0], "freq1_input")):
                continue
            cards.append((real, hw[0]))
        pick = next((c for c in cards if want and want in c[0].lower()), cards[device_index] if device_index < len(cards) else (cards[0] if cards else None))
        if pick:
            self.freq = os.path.join(pick[1], "freq1_input")
            self.power = next((q for q in (os.path.join(pick[1], n) for n in ("power1_average", "power1_input")) if os.path.exists(q)), None)
        self._stop = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError, TypeError):
            return None

    def __enter__(self):
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                self.samples.append((self._read(self.freq), self._read(self.power)))
                self._stop.wait(0.02)
        self._th = threading.Thread(target=loop, daemon=True)
        if self.freq:
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.freq:
            self._th.join()

    def summary(self):
        f = [a / 1e6 for a, _ in self.samples if a]
        w = [b / 1e6 for _, b in self.samples if b]
        return {"clock_mhz_mean": round(sum(f) / len(f), 1) if f else None, "clock_mhz_min": round(min(f), 1) if f else None,
                "power_w_mean": round(sum(w) / len(w), 1) if w else None, "samples": len(self.samples)}


def chip_info(device_index):
    p = torch.cuda.get_device_properties(device_index)
    return {"gpu_name": p.name, "gcn_arch": getattr(p, "gcnArchName", None), "compute_units": p.multi_processor_count, "hbm_gib": round(p.total_memory / 2 ** 30, 1)}


SETTLE_STEPS = 10


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # BASELINE.md §6 protocol: >= 50 timed steps (the driver passes its own --steps)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (the headline metric is defined at 128)")
    ap.add_argument("--preset", type=str, default="base", choices=list(PRESETS), help="single-GPU slice of another BASELINE.json config (not the headline)")
    ap.add_argument("--loss", type=str, default="mse", help="reconstruction loss (the headline metric is defined with mse; e.g. mse_ssim, ms_ssim for SURVEY §8 f-4)")
    ap.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the "
                    "N > 1 code path with several ranks on one GPU: set CSMAE_BENCH_ONE_GPU=1)")
    ap.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "fp8"], help="fp8: the blocks' forward / dX GEMMs on the fp8 MFMA path "
                    "(BASELINE.json configs[4]; use with --preset huge14).  The headline metric is defined in bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if os.environ.get("CSMAE_BENCH_ONE_GPU"):  # plumbing test only: every rank on device 0 (RCCL refuses that, gloo does not)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL's kernels share the chip with GEMM workgroups that each own a whole CU (160 KiB LDS, 512 threads): a high-priority queue lets
        # a collective's workgroups take the next CU that frees up instead of waiting behind the rest of the GEMM's grid
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(a.backend)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    import csmae_hip
    csmae_hip.load()
    if csmae_hip.source_hash() != csrc_hash():
        log(f"WARNING: libcsmae_hip.so was built from csrc {csmae_hip.source_hash()[:16]}, the tree holds {csrc_hash()[:16]} (stale build?)")

    from csmae_hip import debug_opt as csmae_dbg   # CSMAE_DEBUG="key=value,...": experiment knobs
    factory, size, patch, chans, pbatch, gflop = PRESETS[a.preset]
    if a.batch is None:
        a.batch = pbatch
    model, wrapped, opt = build(device, a.batch, world, a.loss, a.preset)
    if a.dtype == "fp8":
        model.compute_dtype = "fp8"
    torch.manual_seed(0 + rank)  # main_pretrain.py:368
    samples = torch.randn(a.batch, chans, size, size, device=device)

    if csmae_dbg("bench_stream"):   # experiment aid (CU partition): the whole step on a non-blocking stream of its own, nothing on
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1 if csmae_dbg("bench_stream") == "hi" else 0))   # the legacy null stream — a CU-masked stream is a BLOCKING stream (hipExtStreamCreateWithCUMask takes no flags)
    main_cus = csmae_dbg("main_cus")   # ... "lo:hi" = mask bits of the main stream
    if main_cus:
        from csmae_hip import ops as _ops
        lo, hi = (int(v) for v in main_cus.split(":"))
        torch.cuda.set_stream(_ops.cu_masked_stream(lo, hi))

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _, _ = wrapped(samples, mask_ratio=0.75)
        loss.backward()
        opt.step()
        return loss

    # Initialisation, not measurement: the first ~8 steps of a process carry one-off costs (allocator growth, HIP signal / kernarg
    # pools, clock ramp) that showed up as a 70-100 ms stall somewhere in steps 5..10 when the queue runs unsynchronised.
    for _ in range(SETTLE_STEPS):
        step()
    torch.cuda.synchronize()
    log(f"model built and settled ({SETTLE_STEPS} steps); warm-up {a.warmup} steps")
    for _ in range(a.warmup):
        loss = step()
    torch.cuda.synchronize()
    log("warm-up done; timing")

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    watch = ChipWatch(local)
    with watch:
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = step()
        fence()
        mine = time.perf_counter() - t0
    elapsed = torch.tensor([mine], device=device, dtype=torch.float64)
    dp_check = None
    if world > 1:
        # self-verification of the N > 1 line (nobody can re-run it by hand): every rank really took part (all-reduce of ones), the
        # per-rank wall times (a straggler shows as max >> min), the loss each rank saw, and that the data-parallel replicas still hold
        # identical weights after the timed steps (the fp32 masters' checksum: max - min over ranks must be exactly 0)
        ones = torch.ones(1, device=device, dtype=torch.float64)
        dist.all_reduce(ones)
        tmin, tmax = elapsed.clone(), elapsed.clone()
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        chk = model._flat.p.double().sum().reshape(1)
        cmin, cmax = chk.clone(), chk.clone()
        dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
        lmin, lmax = loss.detach().double().reshape(1).clone(), loss.detach().double().reshape(1).clone()
        dist.all_reduce(lmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(lmax, op=dist.ReduceOp.MAX)
        sync = wrapped._sync
        # What the exchange costs the step (the first multi-GPU run must explain itself; nobody can re-run it by hand): (a) per-bucket
        # times from HIP events on the exchange stream over one more step, (b) the same K steps once more WITHOUT the collectives
        # (`no_sync`: every rank steps on its local gradients — the replicas diverge, which is why the weight checksum above was taken
        # first): exposed communication per step = ms_with - ms_without.
        buckets = None
        if sync is not None:
            sync.timing = True
            step()
            per = sync.bucket_times()
            sync.timing = False
            buckets = [{"elements": int(n), "ms": round(ms, 3)} for n, ms in per]
        fence()
        t1 = time.perf_counter()
        with wrapped.no_sync():
            for _ in range(a.steps):
                step()
        fence()
        nosync = torch.tensor([time.perf_counter() - t1], device=device, dtype=torch.float64)
        dist.all_reduce(nosync, op=dist.ReduceOp.MAX)
        ms_with, ms_without = 1e3 * float(tmax) / a.steps, 1e3 * float(nosync) / a.steps
        dp_check = {"ranks_seen": int(ones.item()), "ms_per_step_min": round(1e3 * float(tmin) / a.steps, 3), "ms_per_step_max": round(1e3 * float(tmax) / a.steps, 3),
                    "weights_checksum_spread": float(cmax - cmin), "loss_min": round(float(lmin), 5), "loss_max": round(float(lmax), 5),
                    "grad_payload": "bf16" if (sync is not None and sync._staging is not None) else "fp32",
                    "buckets_per_step": len(buckets) if buckets is not None else 0, "backend": a.backend,
                    "ms_per_step_without_collectives": round(ms_without, 3), "exposed_comm_ms": round(ms_with - ms_without, 3),
                    "bucket_allreduce": buckets, "bucket_allreduce_ms_sum": round(sum(b["ms"] for b in buckets), 3) if buckets else 0.0,
                    # knobs a real multi-GPU run should re-measure: RCCL's queue priority, and its channel count = the CUs its
                    # kernels take from the GEMMs (a CU mask cannot confine them: torch's ProcessGroupNCCL launches on a stream of its own,
                    # and static CU partitions measured slower than time-slicing on one GPU — DESIGN §5)
                    "rccl_env": {k: os.environ.get(k) for k in ("TORCH_NCCL_HIGH_PRIORITY", "NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS")}}
        elapsed = tmax
    elapsed = float(elapsed)
    final_loss = float(loss.detach())
    log(f"timed {a.steps} steps in {elapsed:.3f} s, loss {final_loss:.4f}")
    if not (final_loss == final_loss and abs(final_loss) < 1e30):
        raise SystemExit(f"non-finite loss {final_loss}")

    kernel = None
    if not a.no_kernel_timing:
        from csmae_hip.ops import KernelTimer
        with KernelTimer() as kt:
            step()
        summ = kt.summary()
        log("kernel timing pass done")
        tot_ms = sum(v["ms"] for v in summ.values())
        tot_fl = sum(v["work"] for v in summ.values())
        kernel = dict(name="GEMM family: gemm_bf16_k2_kernel (128x256 tiles, two workgroups per CU) + gemm_bf16_k8_kernel (128x512 full-row tiles, LayerNorm in the epilogue) + gemm_bf16_k64_kernel + gemm_dw_group_kernel (256x256 tiles) (MFMA 16x16x32 bf16)" +
                      (" + gemm_fp8_pipe_kernel / gemm_fp8_dw_group_kernel (MFMA 16x16x128 f8f6f4)" if a.dtype == "fp8" else "") +
                      "; serialised on one stream for the HIP-event timing (every launch alone on the chip; in the overlapped step the weight-gradient launches are held to 160 workgroups)", launches_per_step=sum(v["launches"] for v in summ.values()),
                      ms_per_step=round(tot_ms, 3), tflops=round(tot_fl / tot_ms / 1e9, 1),
                      by_layout={k: dict(ms=round(v["ms"], 3), launches=v["launches"], tflops=round(v["work"] / v["ms"] / 1e9, 1)) for k, v in summ.items()})
    # HBM bytes per GEMM launch from the committed PMC passes (counters cannot be read from inside this process).  The profile names the
    # kernel sources it was taken on; any later edit of csrc/ makes it stale and it is NOT reported (null) until the passes are re-run.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        if t.get("csrc_sha256") == csrc_hash() == csmae_hip.source_hash():
            traffic = {"bytes_per_launch": int(t["gemm_mb_per_launch"] * 2 ** 20), "step_hbm_gb": t["step_hbm_gb"], "source": t["source"], "csrc_sha256": t["csrc_sha256"][:16]}
        else:
            log("profiles/pmc_traffic.json was measured on different kernel sources: roofline.traffic = null (re-run tools/pmc_traffic.py)")
    except (OSError, KeyError, ValueError):
        pass
    if rank == 0:
        ips = a.batch * world * a.steps / elapsed
        ach = ips / world * gflop / 1e3  # TFLOP/s per GPU
        scale = a.batch == BATCH_PER_GPU and a.loss == "mse" and a.preset == "base" and a.dtype == "bf16"
        peak = PEAK_FP8_TFLOPS if a.dtype == "fp8" else PEAK_BF16_TFLOPS
        # which roof binds: the MFMA floor (algorithmic FLOP at the dense peak) against the HBM floor (measured traffic — or, without a valid
        # PMC profile, the algorithmic bytes — at the 8 TB/s peak); `frac` stays the fraction of the MFMA peak BASELINE.json's north_star asks for
        n_params = sum(p.numel() for p in model.parameters())
        alg_gb = algorithmic_hbm_gb(a.preset, a.batch, n_params)
        step_gb = traffic["step_hbm_gb"] if (traffic and scale) else None
        ms = 1e3 * elapsed / a.steps
        mfma_floor_ms = a.batch * gflop / peak
        hbm_floor_ms = (step_gb if step_gb is not None else alg_gb) / PEAK_HBM_TBS
        hbm = {"step_gb": step_gb, "algorithmic_gb": round(alg_gb, 1), "tb_per_s": round(step_gb / ms, 3) if step_gb is not None else None,
               "frac_of_8": round(step_gb / ms / PEAK_HBM_TBS, 4) if step_gb is not None else None,
               "algorithmic_tb_per_s": round(alg_gb / ms, 3), "floor_ms": round(hbm_floor_ms, 2), "floor_ms_at_6p3": round(hbm_floor_ms * PEAK_HBM_TBS / 6.3, 2),
               "mfma_floor_ms": round(mfma_floor_ms, 2), "peak_tb_per_s": PEAK_HBM_TBS}
        out = {
            "metric": "pretrain images/sec ViT-B/16 224^2 two-scale", "value": round(ips, 2), "unit": "images/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{factory} /{patch}, {size}^2 two-scale crops, mask 0.75, AdamW, full optimizer step" if a.preset != "base" else
                       "MAE_ViT_MsLdCeCd ViT-B/16, 224^2 two-scale crops, mask 0.75, AdamW, full optimizer step",
                       "loss": a.loss, "batch_per_gpu": a.batch, "global_batch": a.batch * world, "input": [chans, size, size], "parallelism": f"dp{world}",
                       "headline_config": bool(scale)},
            "loss": round(final_loss, 5), "library_source_sha256": csmae_hip.source_hash()[:16],
            "chip": dict(chip_info(local), **watch.summary()),
            "roofline": {"bound": "hbm" if hbm_floor_ms > mfma_floor_ms else "mfma", "hbm": hbm, "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         "traffic": traffic if (a.preset == "base" and a.dtype == "bf16" and scale) else None, "algorithmic_gflop_per_image": gflop, "dominant_kernel": kernel},
        }
        if dp_check is not None:
            out["data_parallel"] = dp_check
        if world == 1 and not a.no_cpu_baseline and a.preset == "base":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
