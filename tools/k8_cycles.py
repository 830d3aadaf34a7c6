#!/usr/bin/env python3
"""Shader clocks of one workgroup (block 300, thread 0) of the full-row GEMM + LayerNorm kernel (csrc/gemm_ln.hip) by phase.  Builds its own
-DGEMM_TIMING variant of the library on the box (hipcc there, ~1 min) into /tmp.  usage: tools/k8_cycles.py [extra hipcc flags]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = "/tmp/libcsmae_k8t.so"
if not os.environ.get("K8_CHILD"):
    F = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DGEMM_TIMING".split() + sys.argv[1:]
    src = os.path.join(ROOT, "cross-scale-mae_amd/csrc")
    procs = [subprocess.Popen(["hipcc", *F, "-c", f"{src}/{n}.hip", "-o", f"/tmp/k8t_{n}.o"]) for n in ("gemm", "gemm_k2", "gemm_ln")]
    assert all(p.wait() == 0 for p in procs)
    objs = [os.path.join(ROOT, "build/obj", f"{n}.o") for n in ("api", "attention", "fp8", "loss", "norm", "optim", "tokens")]
    if not all(os.path.exists(o) for o in objs):   # (build/obj does not travel with gpurun)
        for o in objs:
            n = os.path.basename(o)[:-2]
            subprocess.check_call(["hipcc", *F, "-c", f"{src}/{n}.hip", "-o", f"/tmp/k8t_{n}.o"])
        objs = [f"/tmp/k8t_{os.path.basename(o)}" for o in objs]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "/tmp/k8t_gemm.o", "/tmp/k8t_gemm_k2.o", "/tmp/k8t_gemm_ln.o", "-o", lib])
    sys.exit(subprocess.run([sys.executable, __file__], env=dict(os.environ, K8_CHILD="1", CSMAE_LIB_PATH=lib)).returncode)
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
import torch
import csmae_hip
from csmae_hip import ops
L = csmae_hip.load()


def ts(fn_name):
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    getattr(L, fn_name)(buf)
    return list(buf)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def show(label, us, t, K):
    print(f"{label:34s} {us:7.1f} us  pro {t[1] - t[0]:5d} loop {t[2] - t[1]:6d} = {(t[2] - t[1]) / (K // 64):5.0f}/step epi {t[3] - t[2]:6d}  tile {t[3] - t[0]:6d}")


M, N, bf, dev = 50432, 512, torch.bfloat16, "cuda"
g, b, bias = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1, torch.randn(N, device=dev) * 0.1
mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
x, y, resid = (torch.randn(M, N, device=dev).to(bf) for _ in range(3))
part = torch.empty(1024 * 2 * N, device=dev)
for K in (512, 2048):
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
    Wk = W.view(N, K // 32, 32).permute(1, 0, 2).contiguous().reshape(-1)
    us = timed(lambda: ops.gemm_ln_fwd(A, Wk, bias, resid, x, g, b, y, mean, rstd))
    show(f"k8 fwd K={K} (+resid +LN)", us, ts("csmae_debug_k8_ts"), K)
    L.csmae_gemm_k2_mode(3, 3)
    us = timed(lambda: ops.gemm_ks(A, Wk, W, x, bias=bias, epilogue=2, resid=resid))
    show(f"k2 NT K={K} (+resid)", us, ts("csmae_debug_k2_ts"), K)
    L.csmae_gemm_k2_mode(0, 0)
    us = timed(lambda: ops.gemm(A, W, x, bias=bias, epilogue=2, resid=resid))
    show(f"k64 NT K={K} (+resid)", us, ts("csmae_debug_gemm_ts"), K)
ops.layernorm_fwd(x, g, b, y, mean, rstd)
for K in (1536, 2048):
    dY = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(K, N, device=dev) * K ** -0.5).to(bf)
    us = timed(lambda: ops.gemm_ln_bwd(dY, W, x, mean, rstd, g, resid, y, partial_ws=part))
    show(f"k8 bwd K={K} (+LN' +dres)", us, ts("csmae_debug_k8_ts"), K)
    L.csmae_gemm_k2_mode(3, 3)
    us = timed(lambda: ops.gemm(dY, W, y, trans_b=True))
    show(f"k2 NN K={K} (plain)", us, ts("csmae_debug_k2_ts"), K)
    L.csmae_gemm_k2_mode(0, 0)
    us = timed(lambda: ops.gemm(dY, W, y, trans_b=True))
    show(f"k64 NN K={K} (plain)", us, ts("csmae_debug_gemm_ts"), K)
