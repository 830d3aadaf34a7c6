"""Kernel-level parity: every HIP entry point (called through the C ABI) against the CPU oracle / plain torch
fp32 on the same seeded inputs, plus the committed golden vectors where the op has one.  Needs an MI355X."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from csmae_hip import ops as o
    import csmae_hip
    csmae_hip.load()
    return o


def dev(t, dtype=None):
    t = t.cuda()
    return t.to(dtype) if dtype is not None else t


def assert_close(actual, expected, rtol, atol, what=""):
    a, e = actual.detach().float().cpu(), expected.detach().float().cpu()
    assert a.shape == e.shape, (what, a.shape, e.shape)
    err = (a - e).abs()
    tol = atol + rtol * e.abs()
    if not bool((err <= tol).all()):
        idx = int((err - tol).argmax())
        raise AssertionError(f"{what}: max|err|={err.max():.3e} at flat {idx}: got {a.reshape(-1)[idx]:.6g} want {e.reshape(-1)[idx]:.6g}; "
                             f"bad={int((err > tol).sum())}/{err.numel()} ref_absmax={e.abs().max():.3e}")


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


# ------------------------------------------------------------------------------------------------ GEMM
SHAPES = [(128, 128, 64), (256, 384, 192), (104, 72, 40), (8, 16, 8), (1024, 768, 768), (640, 2304, 768), (200, 512, 2048),
          (512, 512, 200), (304, 264, 328), (776, 512, 584)]   # (K not a multiple of 64 at M, N >= 256: routed past the predicate-free pipelined kernel)


@pytest.mark.parametrize("mnk", SHAPES)
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_layouts(ops, mnk, layout, dtype):
    M, N, K = mnk
    A = rnd(M, K, seed=1).to(dtype)
    B = rnd(N, K, seed=2).to(dtype)  # logical B(k,n) = B[n,k]
    bias = rnd(N, seed=3)
    ref = A.float() @ B.float().t() + bias
    ta, tb = layout[0] == "t", layout[1] == "n"
    a_st = A.t().contiguous() if ta else A
    b_st = B.t().contiguous() if tb else B
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm(dev(a_st), dev(b_st), out, trans_a=ta, trans_b=tb, bias=dev(bias))
    tol = 2e-5 if dtype == torch.float32 else 1e-4
    assert_close(out, ref, tol, tol * K ** 0.5, f"gemm {layout} {mnk} {dtype}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_epilogues(ops, dtype):
    M, N, K = 200, 256, 136
    A, B, bias = rnd(M, K, seed=4).to(dtype), rnd(N, K, seed=5, scale=0.1).to(dtype), rnd(N, seed=6)
    pre = A.float() @ B.float().t() + bias
    # GELU: writes pre-activation + activation in the operand dtype
    out, aux = torch.empty(M, N, device="cuda", dtype=dtype), torch.empty(M, N, device="cuda", dtype=dtype)
    ops.gemm(dev(A), dev(B), out, bias=dev(bias), epilogue=1, aux=aux)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    pg = pre.clone().requires_grad_(True)
    act = torch.nn.functional.gelu(pg)
    act.sum().backward()
    assert_close(aux, pg.grad, tol, tol, "gelu aux = gelu'(pre)")
    assert_close(out, act, tol, tol, "gelu out")
    # residual: fp32 in/out
    R = rnd(M, N, seed=7)
    o32 = torch.empty(M, N, device="cuda")
    ops.gemm(dev(A), dev(B), o32, bias=dev(bias), epilogue=2, resid=dev(R))
    assert_close(o32, pre + R, 1e-5 if dtype == torch.float32 else 1e-4, 1e-4, "resid")
    # in-place residual (resid aliases out) is what the block uses
    x = dev(R.clone())
    ops.gemm(dev(A), dev(B), x, bias=dev(bias), epilogue=2, resid=x)
    assert_close(x, pre + R, 1e-5 if dtype == torch.float32 else 1e-4, 1e-4, "resid in place")
    # dGELU: out = acc * aux
    P = rnd(M, N, seed=8).to(dtype)
    od = torch.empty(M, N, device="cuda", dtype=dtype)
    ops.gemm(dev(A), dev(B), od, epilogue=3, aux=dev(P))
    assert_close(od, (A.float() @ B.float().t()) * P.float(), tol, tol, "dgelu")
    # atomic split-K accumulate on top of existing content (weight-gradient form: both operands K-strided)
    Kt = 1000
    dY, X = rnd(Kt, 64, seed=9).to(dtype), rnd(Kt, 48, seed=10).to(dtype)
    acc0 = rnd(64, 48, seed=11)
    acc = dev(acc0.clone())
    ops.gemm(dev(dY), dev(X), acc, trans_a=True, trans_b=True, epilogue=4, splitk=5)
    assert_close(acc, acc0 + dY.float().t() @ X.float(), 1e-4, 1e-3, "atomic splitk")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("mnk", [(768, 512, 4000), (64, 48, 1000), (2304, 768, 2560), (256, 256, 300), (520, 264, 77 * 8)])
def test_gemm_dw_split_slabs(ops, dtype, mnk):
    M, N, K = mnk
    dY, X = rnd(K, M, seed=12).to(dtype), rnd(K, N, seed=13).to(dtype)
    acc0 = rnd(M, N, seed=14)
    acc = dev(acc0.clone())
    ws = torch.empty(max(M * N * 3, 1 << 20), device="cuda")
    db0 = rnd(M, seed=15)
    db = dev(db0.clone())
    ops.gemm_dw(dev(dY), dev(X), acc, ws, db=db)
    assert_close(acc, acc0 + dY.float().t() @ X.float(), 1e-4, 1e-3 * K ** 0.5 / 30, f"gemm_dw {mnk}")
    assert_close(db, db0 + dY.float().sum(0), 1e-4, 1e-3 * K ** 0.5 / 30, f"gemm_dw bias grad {mnk}")
    # padded operands (ld > width), as the patch-embed / decoder_pred gradients use
    dYp = torch.zeros(K, M + 8, dtype=dtype); dYp[:, :M] = dY
    acc = dev(acc0.clone())
    ops.gemm_dw(dev(dYp)[:, :M], dev(X), acc, ws)
    assert_close(acc, acc0 + dY.float().t() @ X.float(), 1e-4, 1e-3 * K ** 0.5 / 30, "gemm_dw padded")


@pytest.mark.parametrize("case", [(512, [(256, 256)], 4, 0), (1000, [(256, 512), (520, 264)], 16, 8), (4096, [(768, 768), (2304, 768)], 160, 0), (640, [(1024, 1024), (1024, 4096)], 256, 0)])
def test_gemm_dw_group_two_workgroups_per_cu_kernel(ops, case):
    """csmae_gemm_dw_group through k2_tile_tn (csmae_gemm_dw_mode(1): 128 x 256 tiles, hand-counted transposing reads): dW += dY^T X and db += colsum(dY)
    against fp32 torch on the same bf16 operands, one and several K slices, ragged sizes and padded rows, deterministic (ordered fold)."""
    import csmae_hip
    lib = csmae_hip.load()
    K, prods, slots, pad = case
    ws = torch.empty(64 << 20, device="cuda")
    items = []
    for i, (M, N) in enumerate(prods):
        dy = dev(rnd(K, M + pad, seed=300 + i).to(torch.bfloat16))[:, :M]
        x = dev(rnd(K, N + pad, seed=310 + i).to(torch.bfloat16))[:, :N]
        items.append((dy, x))
    if lib.csmae_gemm_dw_mode(1) != 0:
        pytest.skip("the two-workgroups-per-CU weight-gradient kernel is an ablation-build option (-DCSMAE_K2_DW): not in the product library")
    try:
        runs = []
        for _ in range(2):
            outs = [(torch.full((dy.shape[1], x.shape[1]), 0.25, device="cuda"), torch.full((dy.shape[1],), 0.25, device="cuda")) for dy, x in items]
            ops.DwGroup([(dy, x, dw, db) for (dy, x), (dw, db) in zip(items, outs)], ws).launch(slots)
            runs.append(outs)
        for (dy, x), (dw, db), (dw2, db2) in zip(items, runs[0], runs[1]):
            assert torch.equal(dw, dw2) and torch.equal(db, db2), "weight-gradient fold is not deterministic"
            assert_close(dw, 0.25 + dy.float().t() @ x.float(), 1e-4, 1e-3 * K ** 0.5, f"k2 dW {tuple(dw.shape)} K={K}")
            assert_close(db, 0.25 + dy.float().sum(0), 1e-4, 1e-3 * K ** 0.5, f"k2 db {tuple(dw.shape)} K={K}")
    finally:
        lib.csmae_gemm_dw_mode(0)


@pytest.mark.parametrize("case", [(512, [(256, 256)], 4, 0), (1000, [(256, 512), (520, 264)], 16, 16), (4096, [(768, 768), (2304, 768)], 160, 0), (640, [(1280, 1280), (1280, 5120)], 256, 0),
                                  (131, [(256, 256)], 1, 0)])
def test_gemm_dw_group_fp8(ops, case):
    """csmae_gemm_dw_group_fp8: dW += dq_y dq_x dY8^T X8 and db += dq_y colsum(dY8) with both operands K-strided fp8 bytes (dY e5m2, X e4m3), against fp32
    torch on the SAME fp8 values (the kernel's only error is fp32 accumulation order), one and several K slices, ragged sizes, padded rows, a K that is
    not a multiple of the 128-token step; deterministic."""
    K, prods, slots, pad = case
    ws = torch.empty(64 << 20, device="cuda")
    items = []
    for i, (M, N) in enumerate(prods):
        pm, pn = (pad + (-(M + pad)) % 16, pad + (-(N + pad)) % 16) if pad else ((-M) % 16, (-N) % 16)   # rows are 16-byte multiples
        dy = rnd(K, M + pm, seed=320 + i) * 3.0
        x = rnd(K, N + pn, seed=330 + i)
        dy8 = dev(dy).to(torch.float8_e5m2)
        x8 = dev(x).to(torch.float8_e4m3fn)
        dqy, dqx = torch.tensor([0.37 + 0.1 * i], device="cuda"), torch.tensor([1.7], device="cuda")
        items.append((dy8, dqy, x8, dqx, M, N))
    runs = []
    for _ in range(2):
        outs = [(torch.full((M, N), 0.25, device="cuda"), torch.full((M,), 0.25, device="cuda")) for *_, M, N in items]
        ops.DwGroup8([(dy8.view(torch.uint8)[:, :M], dqy, x8.view(torch.uint8)[:, :N], dqx, dw, db) for (dy8, dqy, x8, dqx, M, N), (dw, db) in zip(items, outs)], ws).launch(slots)
        runs.append(outs)
    for (dy8, dqy, x8, dqx, M, N), (dw, db), (dw2, db2) in zip(items, runs[0], runs[1]):
        assert torch.equal(dw, dw2) and torch.equal(db, db2), "fp8 weight-gradient fold is not deterministic"
        a, b = dy8.float()[:, :M], x8.float()[:, :N]
        want = 0.25 + float(dqy) * float(dqx) * (a.double().t() @ b.double()).float()
        assert_close(dw, want, 1e-4, 1e-4 * float(want.abs().max()), f"fp8 dW {M}x{N} K={K}")
        wantb = 0.25 + float(dqy) * a.double().sum(0).float()
        assert_close(db, wantb, 1e-4, 1e-4 * float(wantb.abs().max()), f"fp8 db {M} K={K}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("slots", [4, 40, 160])
def test_gemm_dw_group(ops, dtype, slots):
    """csmae_gemm_dw_group: the weight gradients of several Linear layers over the same tokens in one launch — tiles of all products
    share the chip; one K slice accumulates into dW / db directly, several (`slots`) go through slabs and one ordered fold.
    Shapes: a block's qkv / proj / fc1 / fc2 at a small width incl. ragged edges and a padded operand; fp32 = the per-product fallback."""
    K = 2000 + 24
    shapes = [(768, 256), (256, 256), (1024, 264), (300, 1024)]    # (out M, in N)
    prods, refs = [], []
    for k, (M, N) in enumerate(shapes):
        dY, X = rnd(K, M, seed=20 + k).to(dtype), rnd(K, N, seed=30 + k).to(dtype)
        if k == 3:   # padded dY rows (ld > width), as decoder_pred's gradient has them
            dYp = torch.zeros(K, M + 4, dtype=dtype); dYp[:, :M] = dY
            dy_dev = dev(dYp)[:, :M]
        else:
            dy_dev = dev(dY)
        acc0, db0 = rnd(M, N, seed=40 + k), rnd(M, seed=50 + k)
        prods.append((dy_dev, dev(X), dev(acc0.clone()), dev(db0.clone()) if k != 1 else None))
        refs.append((acc0 + dY.float().t() @ X.float(), db0 + dY.float().sum(0)))
    ws = torch.empty(8 << 20, device="cuda")
    grp = ops.DwGroup(prods, ws)
    grp.launch(slots)
    tol = 1e-3 * K ** 0.5 / 30
    for k, ((_, _, dw, db), (rw, rb)) in enumerate(zip(prods, refs)):
        assert_close(dw, rw, 1e-4, tol, f"group dW {k} slots {slots}")
        if db is not None:
            assert_close(db, rb, 1e-4, tol, f"group db {k} slots {slots}")
    # a second launch accumulates on top, and the result does not depend on which slice finishes last: two fresh runs are bit-identical
    outs = []
    for _ in range(2):
        fresh = [(dy, x, torch.zeros_like(dw), None if db is None else torch.zeros_like(db)) for dy, x, dw, db in prods]
        g2 = ops.DwGroup(fresh, ws)
        g2.launch(slots)
        g2.launch(slots)
        outs.append(fresh)
    for a, b in zip(*outs):
        assert torch.equal(a[2], b[2])
        if a[3] is not None:   # (the fp32 parity path sums its bias gradient with atomics: equal up to the order of the additions)
            assert torch.equal(a[3], b[3]) if dtype == torch.bfloat16 else torch.allclose(a[3], b[3], rtol=1e-5, atol=1e-4)
    k = 0
    assert_close(outs[0][k][2], 2 * (refs[k][0] - rnd(*shapes[k], seed=40 + k)), 1e-4, 2 * tol, "accumulating launches")


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("src_dtype", [torch.bfloat16, torch.float32])
def test_fp8_quantize_matches_torch_float8(ops, fmt, src_dtype):
    """Per-tensor current scaling into OCP fp8 (e4m3fn / e5m2): scale = FMAX / max|x| on the device, bytes equal to torch's own float8
    conversion of the scaled tensor (round to nearest even), de-quantisation factor = max|x| / FMAX; transposed mirror = the same bytes
    transposed."""
    tdt, fmax = (torch.float8_e4m3fn, 448.0) if fmt == 0 else (torch.float8_e5m2, 57344.0)
    for rows, cols in ((64, 128), (200, 520), (1284, 68)):
        x = (rnd(rows, cols, seed=70) * 3.0).to(src_dtype)
        x[3, 5] = 17.5   # the maximum
        q = torch.empty(rows, cols, device="cuda", dtype=torch.uint8)
        amax, dq = torch.zeros(64, device="cuda"), torch.empty(1, device="cuda")   # an amax = 64 partial maxima
        ops.fp8_quantize(dev(x), q, amax, dq, fmt=fmt)
        am = x.float().abs().max()
        assert float(amax.max()) == float(am) and abs(float(dq) - float(am) / fmax) <= 1e-6 * float(am) / fmax
        scale = torch.tensor(fmax, dtype=torch.float32) / am
        want = (x.float() * scale).clamp(-fmax, fmax).to(tdt).view(torch.uint8)
        assert torch.equal(q.cpu(), want), (fmt, rows, cols, int((q.cpu() != want).sum()))
        qt = torch.empty(cols, rows, device="cuda", dtype=torch.uint8)
        amax.zero_()
        ops.fp8_quantize(dev(x), qt, amax, dq, fmt=fmt, transpose=True)
        assert torch.equal(qt.cpu(), want.t().contiguous())


@pytest.mark.parametrize("a_fmt", [0, 1])
def test_gemm_fp8_vs_fp32_on_the_quantised_operands(ops, a_fmt):
    """csmae_gemm_fp8 (v_mfma_scale_f32_16x16x128_f8f6f4, fp32 accumulation) against an fp32 matmul of the DE-QUANTISED operands: the
    products are exact in fp32, so the two agree up to the order of the additions.  Asymmetric operands (a transposed or permuted
    fragment layout would not pass), ragged M / N, K with a partial last K step, every epilogue the step uses."""
    from csmae_hip import EPI_DGELU, EPI_GELU, EPI_NONE, EPI_RESID
    adt = torch.float8_e4m3fn if a_fmt == 0 else torch.float8_e5m2
    for M, N, K in ((256, 256, 128), (512, 768, 1280), (300, 520, 400), (1000, 264, 2048)):
        a = rnd(M, K, seed=80) * (1.0 + torch.arange(K) / K)[None, :] + torch.arange(M)[:, None] / M
        w = rnd(N, K, seed=81) * 0.05 + torch.arange(N)[:, None] / (8.0 * N)
        a8 = torch.empty(M, K, device="cuda", dtype=torch.uint8)
        w8 = torch.empty(N, K, device="cuda", dtype=torch.uint8)
        am, wm = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
        dqa, dqw = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
        ops.fp8_quantize(dev(a.to(torch.bfloat16)), a8, am, dqa, fmt=a_fmt)
        ops.fp8_quantize(dev(w), w8, wm, dqw, fmt=0)
        ad = a8.view(adt).float().cpu() * float(dqa)
        wd = w8.view(torch.float8_e4m3fn).float().cpu() * float(dqw)
        ref = ad.double() @ wd.double().t()
        bias = rnd(N, seed=82)
        out = torch.empty(M, N, device="cuda", dtype=torch.float32)
        ops.gemm_fp8(a8, w8, out, dqa, dqw, a_fmt=a_fmt, bias=dev(bias))
        scale = float(ref.abs().max())
        assert_close(out, (ref + bias).float(), 1e-5, 2e-5 * scale, f"fp8 gemm fp32 out {M}x{N}x{K}")
        outb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        resid = rnd(M, N, seed=83).to(torch.bfloat16)
        ops.gemm_fp8(a8, w8, outb, dqa, dqw, a_fmt=a_fmt, bias=dev(bias), epilogue=EPI_RESID, resid=dev(resid))
        assert_close(outb, (ref + bias + resid.double()).float(), 1e-2, 1e-2 * scale, "fp8 gemm + residual (bf16 stream)")
        aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm_fp8(a8, w8, outb, dqa, dqw, a_fmt=a_fmt, bias=dev(bias), epilogue=EPI_GELU, aux=aux)
        pre = (ref + bias).float()
        assert_close(outb, torch.nn.functional.gelu(pre), 1e-2, 1e-2 * scale, "fp8 gemm + gelu")
        ops.gemm_fp8(a8, w8, outb, dqa, dqw, a_fmt=a_fmt, epilogue=EPI_DGELU, aux=aux)
        assert_close(outb, ref.float() * aux.float().cpu(), 1e-2, 1e-2 * scale, "fp8 gemm x gelu'")
        # delayed scaling fused into the epilogue: the output also leaves as fp8 bytes, scaled with a GIVEN (previous-step) amax, and
        # this step's amax is recorded — what the separate quantisation pass of the consuming GEMM would have produced
        prev = torch.zeros(64, device="cuda"); prev[17] = 0.8 * float(outb.float().abs().max())   # (smaller than today's: saturation path)
        nxt, dqo = torch.zeros(64, device="cuda"), torch.zeros(1, device="cuda")
        q = torch.empty(M, N, device="cuda", dtype=torch.uint8)
        out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        efmt, edt, fmax = (0, torch.float8_e4m3fn, 448.0) if a_fmt == 0 else (1, torch.float8_e5m2, 57344.0)
        ops.gemm_fp8(a8, w8, out2, dqa, dqw, a_fmt=a_fmt, epilogue=EPI_DGELU, aux=aux, emit=(q, efmt, prev, nxt, dqo))
        assert torch.equal(out2, outb)
        assert abs(float(dqo) - float(prev.max()) / fmax) <= 1e-6 * float(dqo)
        assert abs(float(nxt.max()) - float(out2.float().abs().max())) <= 8e-3 * float(nxt.max())      # (recorded before the bf16 rounding)
        deq = q.view(edt).float().cpu() * float(dqo)
        want = out2.float().cpu().clamp(-float(prev.max()), float(prev.max()))
        tol = (0.0625 if a_fmt == 0 else 0.125) * want.abs() + float(prev.max()) * (2.0 ** -9 if a_fmt == 0 else 2.0 ** -16) + 8e-3 * want.abs()
        assert bool(((deq - want).abs() <= tol).all()), float(((deq - want).abs() - tol).max())
        # ... and with the bf16 output skipped (fp8 mode's steady state: every reader takes the fp8 bytes): the same copy, `out` untouched
        for epi, ax in ((EPI_DGELU, aux), (EPI_NONE, None)):
            q_ref, q3, keep = torch.zeros_like(q), torch.zeros_like(q), torch.full_like(out2, 7.0)
            ops.gemm_fp8(a8, w8, out2, dqa, dqw, a_fmt=a_fmt, epilogue=epi, aux=ax, emit=(q_ref, efmt, prev, nxt, dqo))
            nxt.zero_()
            ops.gemm_fp8(a8, w8, keep, dqa, dqw, a_fmt=a_fmt, epilogue=epi, aux=ax, emit=(q3, efmt, prev, nxt, dqo), skip_out=True)
            assert torch.equal(q3, q_ref) and bool((keep == 7.0).all()) and abs(float(nxt.max()) - float(out2.float().abs().max())) <= 8e-3 * float(nxt.max())
        # non-finite values are not hidden by the copy: a NaN stays a NaN (it is not clamped to a finite code), an Inf saturates, both record an amax of +Inf
        bad = dev(bias).clone(); bad[3] = float("nan"); bad[9] = float("inf"); bad[12] = float("-inf")
        q_ok, q_bad = torch.zeros_like(q), torch.zeros_like(q)
        ops.gemm_fp8(a8, w8, out2, dqa, dqw, a_fmt=a_fmt, bias=dev(bias), emit=(q_ok, efmt, prev, nxt, dqo))
        nxt.zero_()
        ops.gemm_fp8(a8, w8, out2, dqa, dqw, a_fmt=a_fmt, bias=bad, emit=(q_bad, efmt, prev, nxt, dqo))
        fb = q_bad.view(edt).float()
        assert bool(torch.isnan(fb[:, 3]).all()) and bool((fb[:, 9] == fmax).all()) and bool((fb[:, 12] == -fmax).all()) and float(nxt.max()) == float("inf")
        keep_cols = [c for c in range(N) if c not in (3, 9, 12)]
        assert torch.equal(q_bad[:, keep_cols], q_ok[:, keep_cols])
        nxt.zero_()
        with pytest.raises(RuntimeError):   # (a residual epilogue's bf16 output is the residual stream: it cannot be skipped)
            ops.gemm_fp8(a8, w8, keep, dqa, dqw, a_fmt=a_fmt, epilogue=EPI_RESID, resid=dev(resid), emit=(q3, efmt, prev, nxt, dqo), skip_out=True)


def test_fp8_weights_batched_matches_per_weight_quantisation(ops):
    """csmae_fp8_weights (all weight mirrors of a model in three launches) writes the same bytes and de-quantisation factors as
    csmae_fp8_amax + csmae_fp8_quantize per weight (plain and transposed mirror)."""
    g = torch.Generator().manual_seed(3)
    shapes = [(256, 128), (1536, 512), (64, 320), (132, 68)]
    parts, desc, off = [], [], 0
    for i, (N, K) in enumerate(shapes):
        w = torch.randn(N, K, generator=g) * (0.02 * (i + 1))
        parts.append(w.reshape(-1))
        desc.append([off, N, K])
        off += N * K
    p = dev(torch.cat(parts))
    w8, w8t = torch.zeros(off, device="cuda", dtype=torch.uint8), torch.zeros(off, device="cuda", dtype=torch.uint8)
    amax, dq = torch.zeros(len(shapes), ops.FP8_SLOTS, device="cuda"), torch.zeros(len(shapes), device="cuda")
    ops.fp8_weights(dev(torch.tensor(desc, dtype=torch.long)), p, w8, w8t, amax, dq)
    for k, ((o, N, K), _) in enumerate(zip(desc, shapes)):
        w = p[o:o + N * K].view(N, K)
        a1, d1 = torch.zeros(ops.FP8_SLOTS, device="cuda"), torch.zeros(1, device="cuda")
        q, qt = torch.empty(N, K, device="cuda", dtype=torch.uint8), torch.empty(K, N, device="cuda", dtype=torch.uint8)
        ops.fp8_quantize(w, q, a1, d1)
        ops.fp8_quantize(w, qt, a1, d1, transpose=True)
        assert torch.equal(w8[o:o + N * K].view(N, K), q) and torch.equal(w8t[o:o + N * K].view(K, N), qt), f"weight {k}"
        assert float(dq[k]) == float(d1) and float(amax[k].max()) == float(w.abs().max())


def test_gemm_rejects_bad_args(ops):
    import csmae_hip
    a = torch.zeros(8, 12, device="cuda", dtype=torch.bfloat16)  # K = 12 not a multiple of 8
    with pytest.raises(csmae_hip.CsmaeError):
        ops.gemm(a, a, torch.zeros(8, 8, device="cuda"))


# ------------------------------------------------------------------------------------------------ attention
ATT = [(3, 50, 2, 64), (2, 197, 2, 32), (2, 17, 3, 32), (2, 5, 2, 64), (1, 65, 2, 80), (1, 257, 1, 32), (2, 33, 2, 16), (1, 224, 1, 64),
       (1, 129, 2, 32), (1, 97, 1, 64), (2, 96, 2, 32), (1, 160, 1, 32),  # single-pass backward: 5 / 4 / 3 / 5 key pairs over 4 waves
       # beyond the LDS-resident kernels (the any-length fallback; --input_size 320 / 384 / 512 with 16-pixel patches, 518 / 14: T = 401, 577, 1025, 1370)
       (1, 401, 2, 64), (1, 577, 1, 80), (1, 300, 2, 32), (1, 1025, 1, 32), (1, 225, 1, 64), (1, 100, 1, 96), (1, 40, 2, 128), (1, 1370, 1, 64)]


@pytest.mark.parametrize("geom", ATT)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_attention_fwd_bwd(ops, geom, dtype):
    B, T, H, hd = geom
    D = H * hd
    qkv = rnd(B * T, 3 * D, seed=20).to(dtype)
    dout = rnd(B * T, D, seed=21).to(dtype)
    q32 = qkv.float().requires_grad_(True)
    q, k, v = q32.reshape(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
    att = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)
    ref = (att @ v).transpose(1, 2).reshape(B * T, D)
    ref.backward(dout.float())
    lse_ref = torch.logsumexp((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)
    out = torch.full((B * T, D), float("nan"), device="cuda", dtype=dtype)
    lse = torch.empty(B, H, T, device="cuda")
    dqkv = torch.full((B * T, 3 * D), float("nan"), device="cuda", dtype=dtype)
    gq = dev(qkv)
    ops.attn_fwd(gq, out, lse, B, T, H, hd)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert_close(out, ref, tol, tol, f"attn out {geom}")
    assert_close(lse, lse_ref, 1e-4 if dtype == torch.float32 else 2e-2, 1e-4 if dtype == torch.float32 else 2e-2, "lse")
    ops.attn_bwd(gq, out, dev(dout), lse, dqkv, B, T, H, hd)
    gscale = q32.grad.abs().max().item()
    assert_close(dqkv, q32.grad, tol, tol * max(gscale, 1.0), f"attn dqkv {geom}")


@pytest.mark.parametrize("case", ["k64 gemm", "small gemm", "attention backward"])
def test_launch_carried_event_orders_a_second_stream(ops, case):
    """csmae_next_launch_event: the kernel carries the event as its dispatch packet's completion signal (pipelined GEMM, bf16 attention
    backward) or — launch sites that do not support it — gets the plain record from csmae_flush_launch_event.  Either way a second stream
    that waits for the event must see the kernel's output (what torch's `event.record()` behind the kernel guarantees: the hand-off the
    weight-gradient stream of csmae_hip/engine.py relies on; autograd's stream hand-offs in the reference)."""
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    ev = torch.cuda.Event()
    ev.record()                                   # (torch creates the HIP event at its first record)
    spin = torch.empty(64 << 20, device="cuda")   # a long kernel in front, so that the launch is still queued when the second stream starts waiting
    if case == "attention backward":
        B, T, H, hd = 8, 197, 16, 32
        D = H * hd
        qkv = dev(rnd(B * T, 3 * D, seed=30).to(torch.bfloat16))
        dout = dev(rnd(B * T, D, seed=31).to(torch.bfloat16))
        out = torch.empty(B * T, D, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(B, H, T, device="cuda")
        ops.attn_fwd(qkv, out, lse, B, T, H, hd)
        want = torch.empty(B * T, 3 * D, device="cuda", dtype=torch.bfloat16)
        ops.attn_bwd(qkv, out, dout, lse, want, B, T, H, hd)
        got = torch.zeros_like(want)
        run = lambda: ops.attn_bwd(qkv, out, dout, lse, got, B, T, H, hd)
    else:
        M, N, K = (2048, 1024, 512) if case == "k64 gemm" else (96, 72, 40)
        A, Bm = dev(rnd(M, K, seed=32).to(torch.bfloat16)), dev(rnd(N, K, seed=33).to(torch.bfloat16))
        want = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(A, Bm, want)
        got = torch.zeros_like(want)
        run = lambda: ops.gemm(A, Bm, got)
    torch.cuda.synchronize()
    for _ in range(3):
        got.zero_()
        spin.fill_(1.0)
        with ops.launch_done(ev, main.cuda_stream):
            run()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            seen = got.clone()
        side.synchronize()
        assert torch.equal(seen, want), case
    torch.cuda.synchronize()


@pytest.mark.parametrize("geom", [(4, 197, 16, 32), (3, 65, 16, 80), (2, 257, 16, 32), (5, 50, 12, 64)])
def test_attention_emits_the_fp8_copies_of_a_separate_quantisation_pass(ops, geom):
    """csmae_attn_fwd_q / csmae_attn_bwd_q (fp8 mode, BASELINE configs[4]): the kernels' fp8 copies of `out` (e4m3, for attn.proj) and dqkv (e5m2,
    for attn.qkv's backward) are the bytes csmae_fp8_quantize makes from the bf16 tensors with the same previous-step amax; the recorded amax
    is the tensor's; the bf16 outputs are the plain kernels'."""
    B, T, H, hd = geom
    D = H * hd
    assert ops.attn_resident(ops.BF16, T, hd)
    qkv = dev(rnd(B * T, 3 * D, seed=50).to(torch.bfloat16))
    dout = dev(rnd(B * T, D, seed=51).to(torch.bfloat16))
    out0 = torch.empty(B * T, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, T, device="cuda")
    dq0 = torch.empty(B * T, 3 * D, device="cuda", dtype=torch.bfloat16)
    ops.attn_fwd(qkv, out0, lse, B, T, H, hd)
    ops.attn_bwd(qkv, out0, dout, lse, dq0, B, T, H, hd)
    for which, ref, fmt in (("fwd", out0, ops.FP8_E4M3), ("bwd", dq0, ops.FP8_E5M2)):
        prev = torch.zeros(ops.FP8_SLOTS, device="cuda")
        prev[7] = float(ref.float().abs().max()) * 0.8      # a stale amax: some values clamp, as one step after a jump
        nxt, dq = torch.zeros(ops.FP8_SLOTS, device="cuda"), torch.zeros(1, device="cuda")
        q = torch.zeros(ref.shape, device="cuda", dtype=torch.uint8)
        got = torch.zeros_like(ref)
        if which == "fwd":
            ops.attn_fwd(qkv, got, lse, B, T, H, hd, emit=(q, fmt, prev, nxt, dq))
        else:
            ops.attn_bwd(qkv, out0, dout, lse, got, B, T, H, hd, emit=(q, fmt, prev, nxt, dq))
        assert torch.equal(got, ref), which
        want_q, want_next, want_dq = torch.zeros_like(q), torch.zeros(ops.FP8_SLOTS, device="cuda"), torch.zeros(1, device="cuda")
        ops.fp8_quantize(ref, want_q, prev, want_dq, fmt=fmt, amax_next=want_next)
        assert torch.equal(q, want_q), which
        assert float(nxt.max()) == float(ref.float().abs().max()) == float(want_next.max()), which
        assert float(dq) == float(want_dq), which
        if which == "bwd":   # fp8 mode's steady state: no reader of the bf16 gradient — only the fp8 copy is written, the same bytes
            q2, keep = torch.zeros_like(q), torch.full_like(ref, 7.0)
            nxt.zero_()
            ops.attn_bwd(qkv, out0, dout, lse, keep, B, T, H, hd, emit=(q2, fmt, prev, nxt, dq), skip_out=True)
            assert torch.equal(q2, want_q) and bool((keep == 7.0).all()) and float(nxt.max()) == float(want_next.max())


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("MD", [(37, 768), (200, 512), (9, 64), (5, 128), (16, 1024), (7, 1280)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_layernorm(ops, MD, dtype):
    M, D = MD
    x = rnd(M, D, seed=30, scale=2.0) + 0.5
    g, b = rnd(D, seed=31) * 0.2 + 1.0, rnd(D, seed=32) * 0.1
    dy = rnd(M, D, seed=33).to(dtype)
    dres = rnd(M, D, seed=34)
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    ref.backward(dy.float())
    y = torch.empty(M, D, device="cuda", dtype=dtype)
    y32 = torch.empty(M, D, device="cuda")
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.layernorm_fwd(dev(x), dev(g), dev(b), y, mean, rstd, y32=y32)
    assert_close(y32, ref, 1e-5, 1e-5, "ln y32")
    assert_close(y, ref, 1e-5 if dtype == torch.float32 else 1e-2, 1e-5 if dtype == torch.float32 else 1e-2, "ln y")
    dx, dxlp = torch.empty(M, D, device="cuda"), torch.empty(M, D, device="cuda", dtype=dtype)
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    ops.layernorm_bwd(dev(dy), dev(x), mean, rstd, dev(g), dx, dg, db, dres_in=dev(dres), dx_lp=dxlp)
    assert_close(dx, xr.grad + dres, 1e-4, 1e-4, "ln dx")
    assert_close(dxlp, xr.grad + dres, 1e-4 if dtype == torch.float32 else 1e-2, 1e-4 if dtype == torch.float32 else 2e-2, "ln dx lp")
    assert_close(dg, gr.grad, 1e-4, 1e-3, "ln dgamma")
    assert_close(db, br.grad, 1e-4, 1e-3, "ln dbeta")
    # per-block partial rows instead of atomics (what the engine uses): accumulates on top of existing content
    pw = torch.empty(64 * 2 * D, device="cuda")
    ops.layernorm_bwd(dev(dy), dev(x), mean, rstd, dev(g), dx, dg, db, dres_in=dev(dres), dx_lp=dxlp, partial_ws=pw)
    assert_close(dg, 2 * gr.grad, 1e-4, 2e-3, "ln dgamma (partials)")
    assert_close(db, 2 * br.grad, 1e-4, 2e-3, "ln dbeta (partials)")


@pytest.mark.parametrize("MD", [(37, 768), (200, 512), (7, 1280)])
def test_layernorm_bf16_residual_stream_and_deferred_param_grads(ops, MD):
    """Throughput mode: the residual stream x and the residual-gradient stream (dres_in -> dx_out) are bf16; statistics and arithmetic
    stay fp32.  Reference: fp32 torch LayerNorm on the bf16-rounded inputs (exact up to the output rounding).  Parameter gradients of a
    batch of launches are left as partial rows and folded by ONE csmae_ln_param_reduce launch into a flat gradient buffer at given
    offsets — deterministically."""
    M, D = MD
    x = (rnd(M, D, seed=30, scale=2.0) + 0.5).to(torch.bfloat16)
    g, b = rnd(D, seed=31) * 0.2 + 1.0, rnd(D, seed=32) * 0.1
    dy = rnd(M, D, seed=33).to(torch.bfloat16)
    dres = rnd(M, D, seed=34).to(torch.bfloat16)
    xr, gr, br = x.float().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    ref.backward(dy.float())
    y = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.layernorm_fwd(dev(x), dev(g), dev(b), y, mean, rstd)
    assert_close(y, ref, 1e-2, 1e-2, "ln y (bf16 stream)")
    assert_close(mean, x.float().mean(1), 1e-5, 1e-5, "ln mean")
    # three "LayerNorms" of a stack: same launch geometry, own partial slices; the reduce adds into a flat buffer at scattered offsets
    K = 3
    part = torch.full((K, 1024 * 2 * D), float("nan"), device="cuda")   # (only the rows a launch writes may be read)
    outs = []
    for k in range(K):
        dx = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
        ops.layernorm_bwd(dev(dy) if k != 1 else dev(dy * 2), dev(x), mean, rstd, dev(g), dx, None, None, dres_in=dev(dres), partial_ws=part[k])
        outs.append(dx)
    assert_close(outs[0], xr.grad + dres.float(), 1e-2, 2e-2, "ln dx (bf16 stream)")
    flatg = torch.ones(8 * D + 64, device="cuda")
    goff = torch.tensor([[0, D], [5 * D + 8, 3 * D], [2 * D, 7 * D + 16]], dtype=torch.long, device="cuda")
    ops.ln_param_reduce(K, M, D, part, goff, flatg)
    for k, scale in enumerate((1.0, 2.0, 1.0)):
        assert_close(flatg[goff[k, 0]: goff[k, 0] + D] - 1.0, scale * gr.grad, 1e-4, 2e-3, f"deferred dgamma {k}")
        assert_close(flatg[goff[k, 1]: goff[k, 1] + D] - 1.0, scale * br.grad, 1e-4, 2e-3, f"deferred dbeta {k}")
    again = torch.ones_like(flatg)
    ops.ln_param_reduce(K, M, D, part, goff, again)
    assert torch.equal(again, flatg)                                          # fixed summation order
    touched = torch.zeros_like(flatg, dtype=torch.bool)
    for k in range(K):
        touched[goff[k, 0]: goff[k, 0] + D] = True
        touched[goff[k, 1]: goff[k, 1] + D] = True
    assert bool((flatg[~touched] == 1.0).all())


@pytest.mark.parametrize("fmt", [0, 1])
def test_layernorm_emits_fp8_copy(ops, fmt):
    """fp8 mode, delayed scaling: LayerNorm forward / backward leave the tensor the next GEMM reads as fp8 bytes too, scaled with a given
    (previous-step) amax; the new amax is recorded.  The bf16 outputs are unchanged by the emission."""
    M, D = 200, 512
    edt, fmax = (torch.float8_e4m3fn, 448.0) if fmt == 0 else (torch.float8_e5m2, 57344.0)
    x = (rnd(M, D, seed=30, scale=2.0) + 0.5).to(torch.bfloat16)
    g, b = rnd(D, seed=31) * 0.2 + 1.0, rnd(D, seed=32) * 0.1
    y, y2 = torch.empty(M, D, device="cuda", dtype=torch.bfloat16), torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.layernorm_fwd(dev(x), dev(g), dev(b), y, mean, rstd)
    prev = torch.zeros(64, device="cuda"); prev[5] = 0.9 * float(y.float().abs().max())
    nxt, dq = torch.zeros(64, device="cuda"), torch.zeros(1, device="cuda")
    q = torch.empty(M, D, device="cuda", dtype=torch.uint8)
    ops.layernorm_fwd(dev(x), dev(g), dev(b), y2, mean, rstd, emit=(q, fmt, prev, nxt, dq))
    assert torch.equal(y, y2)
    am = float(prev.max())
    assert abs(float(dq) - am / fmax) <= 1e-6 * am / fmax and abs(float(nxt.max()) - float(y.float().abs().max())) <= 8e-3 * float(nxt.max())
    deq = q.view(edt).float().cpu() * float(dq)
    want = y.float().cpu().clamp(-am, am)
    tol = (0.0625 if fmt == 0 else 0.125) * want.abs() + am * (2.0 ** -9 if fmt == 0 else 2.0 ** -16) + 8e-3 * want.abs()
    assert bool(((deq - want).abs() <= tol).all())
    xb = x.clone(); xb[5, 7] = float("nan"); xb[9, 1] = float("inf")   # a non-finite row: its copy is NaN (not a clamped finite code), the recorded amax +Inf
    qb, nb = torch.zeros_like(q), torch.zeros(64, device="cuda")
    ops.layernorm_fwd(dev(xb), dev(g), dev(b), y2, torch.empty_like(mean), torch.empty_like(rstd), emit=(qb, fmt, prev, nb, dq))
    fbq = qb.view(edt).float()
    assert bool(torch.isnan(fbq[5]).all()) and bool(torch.isnan(fbq[9]).all()) and float(nb.max()) == float("inf")
    rows = [r for r in range(M) if r not in (5, 9)]
    assert torch.equal(qb[rows], q[rows])
    q2, keep, mean2, rstd2 = torch.zeros_like(q), torch.full_like(y, 7.0), torch.empty_like(mean), torch.empty_like(rstd)   # only the fp8 copy (fp8 mode's steady state)
    ops.layernorm_fwd(dev(x), dev(g), dev(b), keep, mean2, rstd2, emit=(q2, fmt, prev, nxt, dq), skip_out=True)
    assert torch.equal(q2, q) and bool((keep == 7.0).all()) and torch.equal(mean2, mean) and torch.equal(rstd2, rstd)
    dy, dres = rnd(M, D, seed=33).to(torch.bfloat16), rnd(M, D, seed=34).to(torch.bfloat16)
    dx, dx2 = torch.empty(M, D, device="cuda", dtype=torch.bfloat16), torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    part = torch.empty(1024 * 2 * D, device="cuda")
    ops.layernorm_bwd(dev(dy), dev(x), mean, rstd, dev(g), dx, None, None, dres_in=dev(dres), partial_ws=part)
    prev.zero_(); prev[40] = 1.1 * float(dx.float().abs().max()); nxt.zero_()
    ops.layernorm_bwd(dev(dy), dev(x), mean, rstd, dev(g), dx2, None, None, dres_in=dev(dres), partial_ws=part, emit=(q, fmt, prev, nxt, dq))
    assert torch.equal(dx, dx2)
    am = float(prev.max())
    deq = q.view(edt).float().cpu() * float(dq)
    want = dx.float().cpu()
    tol = (0.0625 if fmt == 0 else 0.125) * want.abs() + am * (2.0 ** -9 if fmt == 0 else 2.0 ** -16) + 8e-3 * want.abs()
    assert bool(((deq - want).abs() <= tol).all()) and abs(float(nxt.max()) - float(want.abs().max())) <= 8e-3 * float(nxt.max())


def _kslab_torch(w):
    """K-slab mirror of w [N, K] (bf16): Wk[K/32][N][32], flat (csmae.h csmae_gemm_ks)."""
    N, K = w.shape
    return w.view(N, K // 32, 32).permute(1, 0, 2).contiguous().reshape(-1)


LN_GEMM_SHAPES = [(128, 512, 512), (394, 512, 2048), (1000, 512, 1536), (77, 256, 64), (300, 384, 128), (25216, 512, 512)]


@pytest.mark.parametrize("mnk", LN_GEMM_SHAPES)
def test_gemm_ln_fwd_full_row_tile(ops, mnk):
    """csmae_gemm_ln_fwd (128 x 512 tile, eight waves along N, LayerNorm in the epilogue): X = A W^T + bias + resid, Y = LN(X), mean / rstd of X
    — against fp32 torch on the same bf16 operands AND against the two-kernel path it replaces (csmae_gemm_ks + csmae_layernorm_fwd: X bit-identical,
    same MFMA shape and K order; Y / statistics to fp32 rounding of the different summation trees), ragged M, N < 512 (idle waves), K = 64 (one K step)."""
    from csmae_hip import EPI_RESID
    M, N, K = mnk
    a = dev(rnd(M, K, seed=60).to(torch.bfloat16))
    w = dev(rnd(N, K, seed=61, scale=K ** -0.5).to(torch.bfloat16))
    bias, g, b = dev(rnd(N, seed=62) * 0.1), dev(rnd(N, seed=63) * 0.2 + 1.0), dev(rnd(N, seed=64) * 0.1)
    resid = dev((rnd(M, N, seed=65, scale=2.0) + 0.5).to(torch.bfloat16))
    wk = _kslab_torch(w)
    x = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    y = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    mean, rstd = torch.full((M,), float("nan"), device="cuda"), torch.full((M,), float("nan"), device="cuda")
    assert ops.gemm_ln_supported(M, N, K)
    ops.gemm_ln_fwd(a, wk, bias, resid, x, g, b, y, mean, rstd)
    ref_x = a.float() @ w.float().t() + bias + resid.float()
    assert_close(x, ref_x, 1e-2, 1e-2, f"gemm_ln_fwd x {mnk}")
    xr = x.float()
    assert_close(mean, xr.mean(1), 1e-5, 1e-5, "gemm_ln_fwd mean")
    assert_close(rstd, (xr.var(1, unbiased=False) + 1e-6).rsqrt(), 1e-5, 1e-6, "gemm_ln_fwd rstd")
    assert_close(y, torch.nn.functional.layer_norm(xr, (N,), g, b, 1e-6), 1e-2, 1e-2, "gemm_ln_fwd y")
    # the two kernels it replaces
    x2, y2 = torch.empty_like(x), torch.empty_like(y)
    m2, r2 = torch.empty_like(mean), torch.empty_like(rstd)
    ops.gemm(a, w, x2, bias=bias, epilogue=EPI_RESID, resid=resid)
    ops.layernorm_fwd(x2, g, b, y2, m2, r2)
    assert torch.equal(x, x2), "the fused product differs from csmae_gemm's"
    assert_close(mean, m2, 1e-5, 1e-5, "mean vs ln_fwd"); assert_close(rstd, r2, 1e-5, 1e-6, "rstd vs ln_fwd")
    assert float((y.float() - y2.float()).abs().max()) <= 2.0 ** -6 * float(y2.float().abs().max()), "y vs ln_fwd: more than one bf16 step apart"
    # deterministic
    x3, y3 = torch.empty_like(x), torch.empty_like(y)
    ops.gemm_ln_fwd(a, wk, bias, resid, x3, g, b, y3, m2, r2)
    assert torch.equal(x, x3) and torch.equal(y, y3) and torch.equal(mean, m2) and torch.equal(rstd, r2)


@pytest.mark.parametrize("mnk", LN_GEMM_SHAPES)
@pytest.mark.parametrize("with_dres", [True, False])
def test_gemm_ln_bwd_full_row_tile(ops, mnk, with_dres):
    """csmae_gemm_ln_bwd: dx = LayerNorm'(dY W) + dres_in and the tiles' dgamma / dbeta partial rows (folded by csmae_ln_param_reduce_rows), against
    fp32 torch autograd on the same bf16 operands and against the two-kernel path it replaces (csmae_gemm + csmae_layernorm_bwd on the rounded
    product: same arithmetic, different summation trees)."""
    M, N, K = mnk
    dy = dev(rnd(M, K, seed=70).to(torch.bfloat16))
    w = dev(rnd(K, N, seed=71, scale=K ** -0.5).to(torch.bfloat16))     # [out = K][in = N], torch's layout of the Linear behind the LayerNorm
    x = dev((rnd(M, N, seed=72, scale=2.0) + 0.5).to(torch.bfloat16))
    g, b = dev(rnd(N, seed=73) * 0.2 + 1.0), dev(rnd(N, seed=74) * 0.1)
    dres = dev(rnd(M, N, seed=75).to(torch.bfloat16)) if with_dres else None
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.layernorm_fwd(x, g, b, y, mean, rstd)
    rows = -(-M // 128)
    part = torch.full((2, rows * 2 * N + 64), float("nan"), device="cuda")
    dx = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.gemm_ln_bwd(dy, w, x, mean, rstd, g, dres, dx, partial_ws=part[0])
    # reference: autograd through LayerNorm with the (bf16-rounded, as the two-kernel path rounds it) product as upstream gradient
    t1 = (dy.float() @ w.float()).to(torch.bfloat16).float()
    xr, gr, br = x.float().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (N,), gr, br, 1e-6).backward(t1)
    want = xr.grad + (dres.float() if with_dres else 0.0)
    assert_close(dx, want, 1e-2, 2e-2, f"gemm_ln_bwd dx {mnk}")
    flatg = torch.zeros(2 * N, device="cuda")
    goff = torch.tensor([[0, N]], dtype=torch.long, device="cuda")
    ops.ln_param_reduce_rows(1, rows, N, part, goff, flatg)
    scale = max(1.0, M ** 0.5)
    assert_close(flatg[:N], gr.grad, 2e-2, 2e-2 * scale, "gemm_ln_bwd dgamma")
    assert_close(flatg[N:], br.grad, 2e-2, 2e-2 * scale, "gemm_ln_bwd dbeta")
    # the two kernels it replaces
    t1k = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(dy, w, t1k, trans_b=True)
    dx2 = torch.empty_like(dx)
    pw = torch.empty(1024 * 2 * N, device="cuda")
    dg2, db2 = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    ops.layernorm_bwd(t1k, x, mean, rstd, g, dx2, dg2, db2, dres_in=dres, partial_ws=pw)
    d = (dx.float() - dx2.float()).abs()
    assert float(d.max()) <= 2.0 ** -6 * float(dx2.float().abs().max()) + 1e-6, f"dx vs gemm + ln_bwd: {float(d.max())}"
    assert_close(flatg[:N], dg2, 1e-3, 1e-3 * scale, "dgamma vs ln_bwd"); assert_close(flatg[N:], db2, 1e-3, 1e-3 * scale, "dbeta vs ln_bwd")
    # deterministic (fixed summation order across the eight waves and the tiles)
    dx3 = torch.empty_like(dx)
    ops.gemm_ln_bwd(dy, w, x, mean, rstd, g, dres, dx3, partial_ws=part[1])
    assert torch.equal(dx, dx3) and torch.equal(part[0][: rows * 2 * N], part[1][: rows * 2 * N])


def test_gemm_ln_rejects_shapes_it_cannot_hold(ops):
    import csmae_hip
    assert not ops.gemm_ln_supported(128, 768, 768) and not ops.gemm_ln_supported(128, 512, 96) and not ops.gemm_ln_supported(128, 520, 64)
    a = torch.zeros(128, 768, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(csmae_hip.CsmaeError, match="N <= 512"):
        ops.gemm_ln_bwd(a, torch.zeros(768, 768, device="cuda", dtype=torch.bfloat16), a, torch.zeros(128, device="cuda"), torch.zeros(128, device="cuda"),
                        torch.ones(768, device="cuda"), None, torch.empty_like(a))


def test_gemm_residual_epilogue_bf16_stream(ops):
    """C = A W^T + bias + resid with the residual stream in bf16 (C and resid share a dtype), on the pipelined-kernel shapes
    (256-row / 192-row tiles) and a small one; reference in fp32 on the same bf16 operands."""
    from csmae_hip import EPI_RESID
    for M, N, K in ((512, 512, 512), (400, 768, 256), (64, 128, 64), (1000, 520, 320)):
        a = rnd(M, K, seed=50).to(torch.bfloat16)
        w = (rnd(N, K, seed=51) * 0.05).to(torch.bfloat16)
        bias = rnd(N, seed=52)
        resid = rnd(M, N, seed=53).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(dev(a), dev(w), out, bias=dev(bias), epilogue=EPI_RESID, resid=dev(resid))
        ref = a.float() @ w.float().t() + bias + resid.float()
        assert_close(out, ref, 1e-2, 2e-2, f"gemm resid bf16 {M}x{N}x{K}")
        with pytest.raises(AssertionError):
            ops.gemm(dev(a), dev(w), out, bias=dev(bias), epilogue=EPI_RESID, resid=dev(resid.float()))


def test_gemm_plain_epilogue_bf16_is_the_rounded_fp32_result(ops):
    """The bias-only bf16 epilogue sends the tile through LDS already rounded (csrc/gemm.hip epilogue_rows_bf16_plain); the fp32-output
    epilogue of the same kernel keeps acc + bias in fp32.  Same accumulators, one rounding: the bf16 output must equal the fp32
    output rounded to nearest even, bit for bit — on 256- and 192-row tiles, ragged M, an N that ends inside a lane's 8 columns
    (row pitch a multiple of 8, so the 16-byte-row path is taken) and both K-contiguous-A layouts."""
    for M, N, K, pitch in ((512, 512, 512, 512), (400, 768, 256, 768), (1000, 516, 320, 520), (777, 260, 1000, 264), (2048, 2304, 768, 2304)):
        for tb in (False, True):
            a = rnd(M, K, seed=60).to(torch.bfloat16)
            w = (rnd(K, pitch, seed=61) * 0.05).to(torch.bfloat16)[:, :N] if tb else (rnd(N, K, seed=61) * 0.05).to(torch.bfloat16)
            bias = rnd(N, seed=62)
            o16 = torch.zeros(M, pitch, device="cuda", dtype=torch.bfloat16)[:, :N]
            o32 = torch.zeros(M, pitch, device="cuda", dtype=torch.float32)[:, :N]
            wd = dev(torch.zeros(K, pitch, dtype=torch.bfloat16))[:, :N] if tb else None
            if tb:
                wd.copy_(w)
            ops.gemm(dev(a), wd if tb else dev(w), o16, trans_b=tb, bias=dev(bias))
            ops.gemm(dev(a), wd if tb else dev(w), o32, trans_b=tb, bias=dev(bias))
            assert torch.equal(o16.contiguous().view(torch.int16), o32.to(torch.bfloat16).contiguous().view(torch.int16)), \
                f"plain bf16 epilogue {M}x{N}x{K} trans_b={tb}"
            ref = a.float() @ (w.float() if tb else w.float().t()) + bias
            assert_close(o16, ref, 1e-2, 2e-2, f"gemm plain bf16 {M}x{N}x{K} trans_b={tb}")


def _kslab(ops, w):
    """K-slab mirror of w [N, K] through csmae_weights_kslab, checked against the layout csmae.h states: Wk[k / 32][n][k % 32]."""
    N, K = w.shape
    wd = dev(w).contiguous()
    desc = torch.tensor([[0, N, K]], dtype=torch.long, device="cuda")
    dst = torch.zeros(N * K, device="cuda", dtype=torch.bfloat16)
    ops.weights_kslab(desc, wd.reshape(-1), dst)
    assert torch.equal(dst, wd.view(N, K // 32, 32).permute(1, 0, 2).contiguous().view(-1)), "csmae_weights_kslab layout"
    return dst


@pytest.mark.parametrize("mnk", [(128, 256, 64), (130, 256, 128), (300, 264, 192), (1000, 768, 768), (257, 512, 512), (515, 2304, 768), (2100, 520, 1024)])
def test_gemm_two_workgroups_per_cu_kernel_is_bit_identical(ops, mnk):
    """csrc/gemm_k2.hip (128 x 256 tiles, two 4-wave workgroups per CU; forward products through csmae_gemm_ks with the K-slab weight mirror, dX
    products through csmae_gemm) against the one-workgroup kernels: same MFMA shape, same K order -> the same bits, for every epilogue, ragged
    M / N, bf16 and fp32 outputs; and against fp32 torch on the same bf16 operands."""
    import csmae_hip
    from csmae_hip import EPI_DGELU, EPI_GELU, EPI_NONE, EPI_RESID
    lib = csmae_hip.load()
    M, N, K = mnk
    a = dev(rnd(M, K, seed=70).to(torch.bfloat16))
    w = (rnd(N, K, seed=71) * 0.05).to(torch.bfloat16)
    wk, wd, wt = _kslab(ops, w), dev(w), dev(w.t().contiguous())
    bias = dev(rnd(N, seed=72))
    resid = dev(rnd(M, N, seed=73).to(torch.bfloat16))
    codes = torch.randint(0, 255, (M, N), dtype=torch.uint8, generator=torch.Generator().manual_seed(74)).cuda()
    ref = a.float() @ wd.float().t()
    try:
        for epi, odt in ((EPI_NONE, torch.bfloat16), (EPI_NONE, torch.float32), (EPI_RESID, torch.bfloat16), (EPI_GELU, torch.bfloat16), (EPI_DGELU, torch.bfloat16)):
            for layout in ("nt", "nn"):
                outs = []
                for mode in (3, 0):
                    lib.csmae_gemm_k2_mode(mode, mode)
                    out = torch.full((M, N), float("nan"), device="cuda", dtype=odt)
                    aux = codes.clone() if epi == EPI_DGELU else (torch.zeros(M, N, device="cuda", dtype=torch.uint8) if epi == EPI_GELU else None)
                    kw = dict(bias=None if epi == EPI_DGELU else bias, epilogue=epi, aux=aux, resid=resid if epi == EPI_RESID else None)
                    if layout == "nt":
                        ops.gemm_ks(a, wk, wd, out, **kw)
                    else:
                        ops.gemm(a, wt, out, trans_b=True, **kw)
                    outs.append((out, aux))
                assert torch.equal(outs[0][0].view(torch.int16 if odt == torch.bfloat16 else torch.int32), outs[1][0].view(torch.int16 if odt == torch.bfloat16 else torch.int32)), \
                    f"k2 vs one-workgroup kernel {layout} epi {epi} {odt} {mnk}"
                if epi == EPI_GELU:
                    assert torch.equal(outs[0][1], outs[1][1]), f"gelu' codes {layout} {mnk}"
                if epi == EPI_NONE:
                    assert_close(outs[0][0], ref + bias, 1e-2, 2e-2, f"k2 {layout} {mnk}")
                elif epi == EPI_RESID:
                    assert_close(outs[0][0], ref + bias + resid.float(), 1e-2, 2e-2, f"k2 resid {layout} {mnk}")
    finally:
        lib.csmae_gemm_k2_mode(1, 2)   # the defaults (csrc/gemm.hip)


def test_gemm_ks_falls_back_to_the_plain_weight(ops):
    """Shapes the two-workgroups kernel does not take (K % 64 != 0, M < 128, N < 256, fp32 operands) go through csmae_gemm with the plain weight."""
    for M, N, K, dt_ in ((64, 256, 128, torch.bfloat16), (256, 128, 128, torch.bfloat16), (256, 256, 96, torch.bfloat16), (256, 256, 128, torch.float32)):
        a = dev(rnd(M, K, seed=75).to(dt_))
        w = dev((rnd(N, K, seed=76) * 0.05).to(dt_))
        bias = dev(rnd(N, seed=77))
        out = torch.empty(M, N, device="cuda", dtype=dt_)
        garbage = torch.full((N * K,), float("nan"), device="cuda", dtype=torch.bfloat16)   # must not be read
        ops.gemm_ks(a, garbage, w, out, bias=bias)
        assert_close(out, a.float() @ w.float().t() + bias, 1e-2 if dt_ == torch.bfloat16 else 1e-5, 2e-2 if dt_ == torch.bfloat16 else 1e-4, f"gemm_ks fallback {M}x{N}x{K}")


def test_gemm_gelu_epilogues_with_8bit_derivative(ops):
    """fc1 epilogue: C = gelu(x), aux = gelu'(x) stored as one byte (q = round(200 g + 26)); fc2-backward epilogue: C = acc * aux.
    The code's resolution is 5e-3, i.e. |error| <= 2.5e-3, and saturated units (gelu' = 0 or 1) are codes 26 / 226 exactly — they decode
    without bias; on the pipelined-kernel shapes and on a small one."""
    from csmae_hip import EPI_DGELU, EPI_GELU
    for M, N, K in ((512, 1024, 256), (400, 520, 192), (64, 128, 64)):
        a = rnd(M, K, seed=90).to(torch.bfloat16)
        w = (rnd(N, K, seed=91) * 0.12).to(torch.bfloat16)
        bias = rnd(N, seed=92) * 0.5
        h = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        gq = torch.empty(M, N, device="cuda", dtype=torch.uint8)
        ops.gemm(dev(a), dev(w), h, bias=dev(bias), epilogue=EPI_GELU, aux=gq)
        x = (a.float() @ w.float().t() + bias).requires_grad_(True)
        ref_h = torch.nn.functional.gelu(x)
        ref_h.sum().backward()
        assert_close(h, ref_h, 1e-2, 1e-2, f"gelu {M}x{N}")
        dec = (gq.float().cpu() - 26.0) / 200.0
        assert float((dec - x.grad).abs().max()) <= 2.5e-3 + 2e-3, float((dec - x.grad).abs().max())   # code step / 2 + the bf16-GELU approximation
        sat0, sat1 = x.detach() < -6.0, x.detach() > 6.0     # saturated units decode to exactly 0 / 1 (no systematic bias in dpre)
        assert bool((gq.cpu()[sat0] == 26).all()) and bool((gq.cpu()[sat1] == 226).all()), "saturated gelu' codes"
        dy = rnd(M, N, seed=93).to(torch.bfloat16)          # backward: dpre = (dy W2) * gelu' with dy [M, N2] -> here: a generic product times aux
        w2 = (rnd(N, N, seed=94) * 0.05).to(torch.bfloat16) if N <= 1024 else None
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(dev(dy), dev(w2), out, trans_b=True, epilogue=EPI_DGELU, aux=gq)
        assert_close(out, (dy.float() @ w2.float()) * dec, 1e-2, 2e-2, f"dgelu q8 {M}x{N}")
    # The codes leave in 16-byte pieces (two lanes trade the codes of two passes) when the rows of aux are 16-byte aligned, in 8-byte
    # pieces otherwise: both forms must write the same bytes — full tiles, ragged M, and an N that ends inside a lane pair's 16 columns.
    for M, N, K, pitch in ((512, 1024, 256, 1024), (777, 520, 192, 528), (300, 264, 128, 272)):
        a = dev(rnd(M, K, seed=95).to(torch.bfloat16))
        w = dev((rnd(N, K, seed=96) * 0.12).to(torch.bfloat16))
        bias = dev(rnd(N, seed=97) * 0.5)
        h1, h2 = (torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(2))
        g1 = torch.zeros(M, pitch, device="cuda", dtype=torch.uint8)[:, :N]
        raw = torch.zeros(M * pitch + 16, device="cuda", dtype=torch.uint8)
        off = 8 if raw.data_ptr() % 16 == 0 else 0            # rows 8 (mod 16) bytes off: the 8-byte form
        g2 = raw[off:off + M * pitch].view(M, pitch)[:, :N]
        assert g1.data_ptr() % 16 == 0 and g2.data_ptr() % 16 == 8
        ops.gemm(a, w, h1, bias=bias, epilogue=EPI_GELU, aux=g1)
        ops.gemm(a, w, h2, bias=bias, epilogue=EPI_GELU, aux=g2)
        assert torch.equal(g1, g2) and torch.equal(h1, h2), f"8-bit gelu' store forms differ {M}x{N}"
        assert int(g1.max()) > 200 and int(g1.min()) < 30      # (codes actually written)


def test_stack_boundaries_bf16_stream(ops):
    """embed_assemble / unshuffle_fwd write, embed_assemble_bwd / unshuffle_bwd read the bf16 residual stream; bf16 -> fp32 cast."""
    B2, keep, D, L = 4, 5, 64, 16
    tok, pos, cls = rnd(B2 * keep, D, seed=60), rnd(L + 1, D, seed=61), rnd(D, seed=62)
    ids = torch.stack([torch.randperm(L, generator=torch.Generator().manual_seed(70 + i))[:keep] for i in range(B2)]).int()
    x32, x16 = torch.empty(B2 * (keep + 1), D, device="cuda"), torch.empty(B2 * (keep + 1), D, device="cuda", dtype=torch.bfloat16)
    ops.embed_assemble(dev(tok), dev(pos), dev(cls), dev(ids), x32, B2, keep)
    ops.embed_assemble(dev(tok), dev(pos), dev(cls), dev(ids), x16, B2, keep)
    assert torch.equal(x16, x32.to(torch.bfloat16))
    back = torch.empty_like(x32)
    ops.cast_f32(x16, back)
    assert torch.equal(back, x16.float())
    dx = rnd(B2 * (keep + 1), D, seed=63).to(torch.bfloat16)
    dtok_a, dtok_b = torch.empty(B2 * keep, D, device="cuda", dtype=torch.bfloat16), torch.empty(B2 * keep, D, device="cuda", dtype=torch.bfloat16)
    dcls_a, dcls_b = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    ops.embed_assemble_bwd(dev(dx), dtok_a, dcls_a, B2, keep)
    ops.embed_assemble_bwd(dev(dx.float()), dtok_b, dcls_b, B2, keep)
    assert torch.equal(dtok_a, dtok_b) and torch.equal(dcls_a, dcls_b)
    z, mt, dpos = rnd(B2 * (keep + 1), D, seed=64), rnd(D, seed=65), rnd(L + 1, D, seed=66)
    ids_restore = torch.stack([torch.randperm(L, generator=torch.Generator().manual_seed(80 + i)) for i in range(B2)])
    xd32, xd16 = torch.empty(B2 * (L + 1), D, device="cuda"), torch.empty(B2 * (L + 1), D, device="cuda", dtype=torch.bfloat16)
    ops.unshuffle_fwd(dev(z), dev(mt), dev(dpos), dev(ids_restore), xd32, B2, L, keep)
    ops.unshuffle_fwd(dev(z), dev(mt), dev(dpos), dev(ids_restore), xd16, B2, L, keep)
    assert torch.equal(xd16, xd32.to(torch.bfloat16))
    dxd = rnd(B2 * (L + 1), D, seed=67).to(torch.bfloat16)
    dz_a, dz_b = torch.zeros(B2 * (keep + 1), D, device="cuda", dtype=torch.bfloat16), torch.zeros(B2 * (keep + 1), D, device="cuda", dtype=torch.bfloat16)
    dm_a, dm_b = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    ops.unshuffle_bwd(dev(dxd), dev(ids_restore), dz_a, dm_a, B2, L, keep)
    ops.unshuffle_bwd(dev(dxd.float()), dev(ids_restore), dz_b, dm_b, B2, L, keep)
    assert torch.equal(dz_a, dz_b)
    assert_close(dm_a, dm_b, 1e-5, 1e-5, "mask-token gradient")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("geom", [(6, 9, 64), (37, 5, 2048), (3, 4, 100)])
def test_bnrelu_token_axis(ops, dtype, geom):
    N, L, Hp = geom   # (37, 5, 2048): the unrolled bf16 fast path at the predictor's width; (3, 4, 100): the generic kernels
    u = (rnd(N * L, Hp, seed=40) * 1.5 + 0.3).to(dtype)
    gamma, beta = rnd(L, seed=41) * 0.3 + 1.0, rnd(L, seed=42) * 0.2
    dr = rnd(N * L, Hp, seed=43).to(dtype)
    bn = torch.nn.BatchNorm1d(L)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    bn.train()
    ur = u.float().reshape(N, L, Hp).requires_grad_(True)
    ref = torch.relu(bn(ur))
    ref.backward(dr.float().reshape(N, L, Hp))
    r = torch.empty(N * L, Hp, device="cuda", dtype=dtype)
    mean, rstd = torch.empty(L, device="cuda"), torch.empty(L, device="cuda")
    rm, rv, nbt = torch.zeros(L, device="cuda"), torch.ones(L, device="cuda"), torch.zeros((), device="cuda", dtype=torch.long)
    ops.bnrelu_fwd(dev(u), dev(gamma), dev(beta), r, mean, rstd, N, L, rm, rv, nbt)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert_close(r, ref.reshape(N * L, Hp), tol, tol, "bn relu")
    assert_close(rm, bn.running_mean, 1e-5, 1e-6, "running_mean")
    assert_close(rv, bn.running_var, 1e-5, 1e-6, "running_var")
    assert int(nbt) == 1
    du = torch.empty(N * L, Hp, device="cuda", dtype=dtype)
    dg, db = torch.zeros(L, device="cuda"), torch.zeros(L, device="cuda")
    ops.bnrelu_bwd(dev(u), dev(dr), dev(gamma), dev(beta), mean, rstd, du, dg, db, N, L)
    assert_close(du, ur.grad.reshape(N * L, Hp), tol, tol, "bn du")
    assert_close(dg, bn.weight.grad, 1e-4, 1e-3, "bn dgamma")
    assert_close(db, bn.bias.grad, 1e-4, 1e-3, "bn dbeta")


# ------------------------------------------------------------------------------------------------ token plumbing
def test_cu_masked_stream_runs_kernels_and_rejects_an_empty_mask(ops):
    """csmae_stream_create_cu_mask / csmae_stream_destroy (the CU-partition experiment's tool): a stream confined to a
    few CUs of every XCD runs the library's kernels with the same results; an empty mask is an error, not a queue that never finishes."""
    import ctypes
    import csmae_hip
    st = ops.cu_masked_stream(0, 64)                      # 8 CUs of every XCD
    assert ops.cu_masked_stream(0, 64) is st              # cached per range
    src = rnd(100003, seed=5).cuda()
    a, b = torch.empty(100003, device="cuda", dtype=torch.bfloat16), torch.empty(100003, device="cuda", dtype=torch.bfloat16)
    ops.cast_bf16(src, a)
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        ops.cast_bf16(src, b)
    torch.cuda.current_stream().wait_stream(st)
    assert torch.equal(a, b)
    lib = csmae_hip.load()
    h = ctypes.c_void_p()
    zero = (ctypes.c_uint32 * 8)(*([0] * 8))
    rc = lib.csmae_stream_create_cu_mask(8, ctypes.cast(zero, ctypes.c_void_p), ctypes.cast(ctypes.byref(h), ctypes.c_void_p))
    assert rc != 0 and b"empty mask" in lib.csmae_last_error()
    one = (ctypes.c_uint32 * 8)(*([0xFFFFFFFF] * 8))
    assert lib.csmae_stream_create_cu_mask(8, ctypes.cast(one, ctypes.c_void_p), ctypes.cast(ctypes.byref(h), ctypes.c_void_p)) == 0 and h.value
    assert lib.csmae_stream_destroy(h) == 0
    with pytest.raises(ValueError):
        ops.cu_masked_stream(10, 10)


@pytest.mark.parametrize("D", [7, 64, 768])
def test_random_masking_helper_gathers_through_hip(ops, D):
    """MAE_ViT_Shared.py:57-84 as a stand-alone helper: kept rows == torch.gather over the kernel's own ids (the noise is drawn inside)."""
    import models_mae
    m = models_mae.MAE_ViT_Baseline(dim_model=64, encoder_num_layers=1, encoder_num_heads=2, decoder_embed_dim=32, decoder_num_layers=1,
                                    decoder_num_heads=2, input_size=64, patch_size="16")
    x = rnd(5, 49, D, seed=71).cuda()
    for mr in (0.75, 0.1, 0.99):
        keep = int(49 * (1 - mr))
        xm, mask, ids_restore = m.random_masking(x, mr)
        assert xm.shape == (5, keep, D) and mask.shape == (5, 49) and ids_restore.dtype == torch.long
        ids_keep = torch.argsort(ids_restore, dim=1)[:, :keep]          # inverse permutation = the shuffle order
        assert torch.equal(xm, torch.gather(x, 1, ids_keep.unsqueeze(-1).expand(-1, -1, D)))
        assert torch.equal(mask.sum(1), torch.full((5,), 49.0 - keep, device="cuda"))
    ids = torch.randint(0, 49, (5, 12), dtype=torch.int32, device="cuda")
    out = ops.rows_gather_idx(x, ids, 9, torch.empty(5, 9, D, device="cuda"))
    assert torch.equal(out, torch.gather(x, 1, ids[:, :9].long().unsqueeze(-1).expand(-1, -1, D)))


@pytest.mark.parametrize("L", [16, 196, 256])
@pytest.mark.parametrize("mr", [0.75, 0.5])
def test_mask_sort_bit_exact_vs_reference(ops, L, mr):
    d = np.load(os.path.join(G, "masking.npz"))
    noise = torch.from_numpy(d[f"noise_L{L}"])
    ties = d[f"tie_rows_L{L}"].astype(bool)
    N = noise.shape[0]
    keep = int(L * (1 - mr))
    ids_restore = torch.empty(N, L, device="cuda", dtype=torch.long)
    mask = torch.empty(N, L, device="cuda")
    ids_keep = torch.empty(N, keep, device="cuda", dtype=torch.int32)
    ids_shuffle = torch.empty(N, L, device="cuda", dtype=torch.int32)
    ops.mask_sort(dev(noise), keep, ids_restore, mask, ids_keep, ids_shuffle)
    tag = f"L{L}_mr{int(mr * 100)}"
    free = ~ties
    assert np.array_equal(ids_restore.cpu().numpy()[free], d[f"ids_restore_{tag}"][free])  # bit-exact vs the reference
    assert np.array_equal(mask.cpu().numpy()[free], d[f"mask_{tag}"][free])
    stable = torch.argsort(noise, dim=1, stable=True)
    assert torch.equal(ids_shuffle.cpu().long(), stable)  # tie rows: the stable order
    assert torch.equal(ids_keep.cpu().long(), stable[:, :keep])
    assert torch.equal(ids_restore.cpu(), torch.argsort(stable, dim=1))


def test_crop_resize_vs_reference_golden(ops):
    d = np.load(os.path.join(G, "crop.npz"))
    imgs = torch.from_numpy(d["imgs64"])
    for n, box in enumerate(d["boxes64"]):
        out = torch.empty_like(imgs, device="cuda")
        ops.crop_resize(dev(imgs), out, torch.tensor(box, dtype=torch.int32, device="cuda"))
        assert_close(out, torch.from_numpy(d[f"out64_{n}"]), 0, 5e-6, f"crop64 {box}")
    big = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(int(d["seed224"][0])))
    for n, box in enumerate(d["boxes224"]):
        out = torch.empty_like(big, device="cuda")
        ops.crop_resize(dev(big), out, torch.tensor(box, dtype=torch.int32, device="cuda"))
        assert_close(out.reshape(-1)[torch.from_numpy(d["idx224"]).cuda()], torch.from_numpy(d[f"val224_{n}"]), 0, 5e-6, f"crop224 {box}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("geom", [(3, 3, 64, 16), (2, 4, 128, 16), (2, 3, 56, 14)])
def test_patch_embed_kept_tokens(ops, dtype, geom):
    N, C, S, p = geom
    G_, D = S // p, 64
    L = G_ * G_
    keep = L // 4
    img0, img1 = rnd(N, C, S, S, seed=50), rnd(N, C, S, S, seed=51)
    W, b = rnd(D, C, p, p, seed=52, scale=0.05), rnd(D, seed=53)
    pos, cls = rnd(L + 1, D, seed=54), rnd(D, seed=55)
    noise = torch.rand(2 * N, L, generator=torch.Generator().manual_seed(56))
    ids = torch.argsort(noise, dim=1, stable=True)[:, :keep]
    full = torch.nn.functional.conv2d(torch.cat([img0, img1]).to(dtype).float(), W.to(dtype).float(), b, stride=p).flatten(2).transpose(1, 2) + pos[1:]
    ref = torch.cat([(cls + pos[0]).expand(2 * N, 1, D), torch.gather(full, 1, ids.unsqueeze(-1).expand(-1, -1, D))], dim=1)
    P = C * p * p
    ld = (P + 7) // 8 * 8
    A = torch.empty(2 * N * keep, ld, device="cuda", dtype=dtype)
    ops.patch_gather(dev(img0), dev(img1), dev(ids.int().contiguous()), A, N, C, S, p, keep)
    Wm = torch.zeros(D, ld, dtype=dtype)
    Wm[:, :P] = W.reshape(D, P).to(dtype)
    tok = torch.empty(2 * N * keep, D, device="cuda")
    ops.gemm(A, dev(Wm), tok, bias=dev(b))
    x = torch.empty(2 * N, keep + 1, D, device="cuda")
    ops.embed_assemble(tok, dev(pos), dev(cls), dev(ids.int().contiguous()), x, 2 * N, keep)
    assert_close(x, ref, 1e-4, 1e-4, "patch embed")
    # backward plumbing
    dx = rnd(2 * N, keep + 1, D, seed=57)
    dtok = torch.empty(2 * N * keep, D, device="cuda", dtype=dtype)
    dcls = torch.zeros(D, device="cuda")
    ops.embed_assemble_bwd(dev(dx), dtok, dcls, 2 * N, keep)
    assert_close(dtok, dx[:, 1:].reshape(-1, D), 0 if dtype == torch.float32 else 1e-2, 0 if dtype == torch.float32 else 1e-2, "dtok")
    assert_close(dcls, dx[:, 0].sum(0), 1e-5, 1e-5, "dcls")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_unshuffle_and_row_views(ops, dtype):
    B2, L, keep, Dd = 5, 16, 4, 64
    z = rnd(B2 * (keep + 1), Dd, seed=60)
    mt, dpos = rnd(Dd, seed=61), rnd(L + 1, Dd, seed=62)
    ids_restore = torch.argsort(torch.argsort(torch.rand(B2, L, generator=torch.Generator().manual_seed(63)), dim=1), dim=1)
    zr, mtr = z.clone().requires_grad_(True), mt.clone().requires_grad_(True)
    z3 = zr.reshape(B2, keep + 1, Dd)
    seq = torch.cat([z3[:, 1:], mtr.expand(B2, L - keep, Dd)], dim=1)
    seq = torch.gather(seq, 1, ids_restore.unsqueeze(-1).expand(-1, -1, Dd))
    ref = torch.cat([z3[:, :1], seq], dim=1) + dpos
    dxd = rnd(B2, L + 1, Dd, seed=64)
    ref.backward(dxd)
    xd = torch.empty(B2, L + 1, Dd, device="cuda")
    ops.unshuffle_fwd(dev(z), dev(mt), dev(dpos), dev(ids_restore), xd, B2, L, keep)
    assert_close(xd, ref, 0, 1e-6, "unshuffle")
    dz = torch.full((B2 * (keep + 1), Dd), float("nan"), device="cuda", dtype=dtype)
    dmt = torch.zeros(Dd, device="cuda")
    ops.unshuffle_bwd(dev(dxd), dev(ids_restore), dz, dmt, B2, L, keep)
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert_close(dz, zr.grad, tol, tol, "dz")
    assert_close(dmt, mtr.grad, 1e-5, 1e-5, "dmask_token")
    # row views: rows (n*Td + 1 + l) of the second half
    N, Td = 2, L + 1
    src = rnd(2 * N * Td, Dd, seed=65)
    dst = torch.empty(N * L, Dd, device="cuda", dtype=dtype)
    ops.rows_gather(dev(src), dst, L, Td, N * Td + 1)
    want = src.reshape(2 * N, Td, Dd)[N:, 1:].reshape(N * L, Dd)
    assert_close(dst, want, tol, tol, "rows_gather")
    acc0 = rnd(2 * N * Td, Dd, seed=66)
    acc = dev(acc0.clone())
    ops.rows_scatter_add(dst, acc, L, Td, N * Td + 1, scale=0.5)
    exp = acc0.clone().reshape(2 * N, Td, Dd)
    exp[N:, 1:] += 0.5 * dst.float().cpu().reshape(N, L, Dd)
    assert_close(acc, exp.reshape(-1, Dd), 1e-6, 1e-6, "rows_scatter_add")


# ------------------------------------------------------------------------------------------------ losses
@pytest.mark.parametrize("kind", ["mse", "l2", "mae", "l1", "bce"])
@pytest.mark.parametrize("norm_pix", [False, True])
def test_recon_loss_fwd_bwd(ops, kind, norm_pix):
    import csmae_oracle as O
    N, C, S, p = 3, 3, 32, 16
    L, P = 4, 768
    B2 = 2 * N
    img0, img1 = rnd(N, C, S, S, seed=70), rnd(N, C, S, S, seed=71)
    pred_full = rnd(B2 * (L + 1), P, seed=72)
    mask = (torch.rand(B2, L, generator=torch.Generator().manual_seed(73)) > 0.3).float()
    mask[0, 0] = 1.0
    pr = pred_full.clone().requires_grad_(True)
    pv = pr.reshape(B2, L + 1, P)[:, 1:]
    lo = O.loss_fn(kind, O.recon_target(img0, p, C, norm_pix), pv[:N], mask[:N])
    lc = O.loss_fn(kind, O.recon_target(img1, p, C, norm_pix), pv[N:], mask[N:])
    g = 0.7
    ((lo + lc) * g).backward()
    mm = None
    if kind == "bce":
        mm = torch.empty(4, device="cuda")
        ops.target_minmax(dev(img0), dev(img1), torch.empty(B2 * L * 2, device="cuda"), mm, B2, N, C, S, p, norm_pix)
    rowloss = torch.empty(B2 * L, device="cuda")
    gp = dev(pred_full)
    ops.recon_loss_fwd(kind, norm_pix, dev(img0), dev(img1), gp, mm, rowloss, B2, N, C, S, p)
    losses = torch.zeros(8, device="cuda")
    ops.loss_finalize(N * L, 2, rowloss, dev(mask), 1.0, losses)
    assert_close(losses[1:3], torch.stack([lo, lc]), 2e-5, 1e-6, f"recon {kind}")
    assert_close(losses[0], lo + lc, 2e-5, 1e-6, "total")
    assert_close(losses[6:8], torch.stack([mask[:N].sum(), mask[N:].sum()]), 0, 0, "masksum")
    gout = torch.tensor([g], device="cuda")
    dpred = torch.full((B2 * (L + 1), P), float("nan"), device="cuda")
    ops.recon_loss_bwd(kind, norm_pix, dev(img0), dev(img1), gp, mm, dev(mask), losses, gout, 1.0, dpred, B2, N, C, S, p)
    assert_close(dpred, pr.grad, 1e-4, 1e-7, f"dpred {kind}")
    # patches the loss does not weigh may be skipped: same terms, their rowloss is 0
    rl2 = torch.full((B2 * L,), float("nan"), device="cuda")
    ops.recon_loss_fwd(kind, norm_pix, dev(img0), dev(img1), gp, mm, rl2, B2, N, C, S, p, mask=dev(mask))
    assert torch.equal(rl2 * dev(mask).reshape(-1), rowloss * dev(mask).reshape(-1)) and float((rl2 * (1 - dev(mask).reshape(-1))).abs().max()) == 0.0


@pytest.mark.parametrize("kind", ["mse", "l2", "mae", "l1", "bce"])
@pytest.mark.parametrize("norm_pix", [False, True])
@pytest.mark.parametrize("C", [3, 4, 5])
def test_recon_loss_bf16_predictions(ops, kind, norm_pix, C):
    """Throughput mode: decoder_pred writes bf16, dpred is bf16.  C = 3 / 4 with 16-pixel patches take the pair-per-lane kernels (mse / l2 /
    mae / l1), everything else the generic ones on bf16 rows — against the oracle on the SAME (bf16-rounded) predictions, and against the
    generic fp32-input kernels."""
    import csmae_oracle as O
    N, S, p = 3, 48, 16
    L, P = 9, 256 * C
    B2 = 2 * N
    img0, img1 = rnd(N, C, S, S, seed=80), rnd(N, C, S, S, seed=81)
    pred16 = rnd(B2 * (L + 1), P, seed=82).to(torch.bfloat16)
    pred_full = pred16.float()
    mask = (torch.rand(B2, L, generator=torch.Generator().manual_seed(83)) > 0.3).float()
    mask[0, 0] = 1.0
    pr = pred_full.clone().requires_grad_(True)
    pv = pr.reshape(B2, L + 1, P)[:, 1:]
    lo = O.loss_fn(kind, O.recon_target(img0, p, C, norm_pix), pv[:N], mask[:N])
    lc = O.loss_fn(kind, O.recon_target(img1, p, C, norm_pix), pv[N:], mask[N:])
    g = 1.3
    ((lo + lc) * g).backward()
    mm = None
    if kind == "bce":
        mm = torch.empty(4, device="cuda")
        ops.target_minmax(dev(img0), dev(img1), torch.empty(B2 * L * 2, device="cuda"), mm, B2, N, C, S, p, norm_pix)
    gm = dev(mask)
    for use_mask in (True, False):
        rowloss = torch.full((B2 * L,), float("nan"), device="cuda")
        ops.recon_loss_fwd(kind, norm_pix, dev(img0), dev(img1), dev(pred16), mm, rowloss, B2, N, C, S, p, mask=gm if use_mask else None)
        losses = torch.zeros(8, device="cuda")
        ops.loss_finalize(N * L, 2, rowloss, gm, 1.0, losses)
        assert_close(losses[1:3], torch.stack([lo, lc]), 2e-5, 1e-6, f"recon {kind} bf16 pred")
    ref_rows = torch.empty(B2 * L, device="cuda")
    ops.recon_loss_fwd(kind, norm_pix, dev(img0), dev(img1), dev(pred_full), mm, ref_rows, B2, N, C, S, p)
    assert_close(rowloss, ref_rows, 2e-5, 1e-6, "rowloss vs the fp32-input kernel")
    gout = torch.tensor([g], device="cuda")
    dpred = torch.full((B2 * (L + 1), P), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.recon_loss_bwd(kind, norm_pix, dev(img0), dev(img1), dev(pred16), mm, gm, losses, gout, 1.0, dpred, B2, N, C, S, p)
    want = pr.grad
    assert torch.isfinite(dpred.float()).all()
    assert_close(dpred.float(), want, 2 ** -8, 1e-7, f"dpred {kind} bf16")     # (one bf16 rounding of the fp32 value)
    if kind != "bce":   # the per-view call of the step engine (a chunk is a view: B2 = N = the view's samples, its own image tensor)
        rl_v = torch.empty(N * L, device="cuda")
        ops.recon_loss_fwd(kind, norm_pix, dev(img1), None, dev(pred16)[N * (L + 1):], None, rl_v, N, N, C, S, p, mask=gm[N:])
        ops.recon_loss_fwd(kind, norm_pix, dev(img0), dev(img1), dev(pred16), mm, rowloss, B2, N, C, S, p, mask=gm)
        assert torch.equal(rl_v, rowloss[N * L:])


def _ssim_run(ops, kind, norm_pix, img0, img1, pred_rows, mask, p, g=0.7, vscale=1.0):
    """Two views through csmae_ssim_fwd / loss_finalize / ssim_apply / ssim_bwd / recon_loss_bwd -> (losses[8], dpred rows)."""
    from csmae_hip import SSIM_KINDS
    base, levels, weight = SSIM_KINDS[kind]
    N, C, S = img0.shape[0], img0.shape[1], img0.shape[2]
    L, P = (S // p) ** 2, p * p * C
    B2 = 2 * N
    gp, gm = dev(pred_rows), dev(mask)
    ws = torch.empty(ops.ssim_workspace_floats(B2, C, S, p, levels), device="cuda")
    terms = torch.empty(2, device="cuda")
    ops.ssim_fwd(levels, norm_pix, dev(img0), dev(img1), gp, gm, ws, terms, B2, N, C, S, p)
    rowloss = torch.zeros(B2 * L, device="cuda")
    if base != "none":
        ops.recon_loss_fwd(base, norm_pix, dev(img0), dev(img1), gp, None, rowloss, B2, N, C, S, p)
    losses = torch.zeros(8, device="cuda")
    ops.loss_finalize(N * L, 2, rowloss, gm, vscale, losses)
    ops.ssim_apply(base == "none", 2, weight, vscale, terms, losses)
    gout = torch.tensor([g], device="cuda")
    extra = torch.full((B2 * L, P), float("nan"), device="cuda")
    ops.ssim_bwd(levels, gp, gm, gout, vscale * weight, ws, extra, B2, N, C, S, p)
    dpred = torch.full((B2 * (L + 1), P), float("nan"), device="cuda")
    ops.recon_loss_bwd(base, norm_pix, dev(img0), dev(img1), gp, None, gm, losses, gout, vscale, dpred, B2, N, C, S, p, extra=extra)
    return losses, dpred


@pytest.mark.parametrize("kind,S,p,norm_pix", [("ssim", 64, 16, False), ("mse_ssim", 64, 16, True), ("ssim", 48, 8, False),
                                               ("ms_ssim", 176, 16, False), ("mse_ms_ssim", 168, 8, False), ("ms_ssim", 200, 8, True)])
def test_ssim_family_fwd_bwd(ops, kind, S, p, norm_pix):
    """MAE_ViT_Shared.forward_loss_{ssim,ms_ssim,mse_ssim,mse_ms_ssim} (:165-267), two views, against the oracle's
    autograd.  200 / 8 -> 25 patches per side: levels 200, 100, 50, 25 (odd: padded pooling), 13.  Tolerance: fp32, 1e-4 on the
    loss; the gradient within 2e-3 of its largest element (E[x^2] - mu^2 cancellations, different summation order)."""
    import csmae_oracle as O
    N, C = 2, 3
    L, P = (S // p) ** 2, p * p * C
    B2 = 2 * N
    img0, img1 = rnd(N, C, S, S, seed=80), rnd(N, C, S, S, seed=81)
    imgs = torch.cat([img0, img1])
    pred_full = rnd(B2 * (L + 1), P, seed=82, scale=0.5)
    pred_full.reshape(B2, L + 1, P)[:, 1:] += 0.7 * O.patchify(imgs, p, C)   # image-like: scores away from the relu clamps
    mask = (torch.rand(B2, L, generator=torch.Generator().manual_seed(83)) > 0.3).float()
    pr = pred_full.clone().requires_grad_(True)
    pv = pr.reshape(B2, L + 1, P)[:, 1:]
    lo = O.loss_fn(kind, O.recon_target(img0, p, C, norm_pix), pv[:N], mask[:N], p, C)
    lc = O.loss_fn(kind, O.recon_target(img1, p, C, norm_pix), pv[N:], mask[N:], p, C)
    g = 0.7
    ((lo + lc) * 0.5 * g).backward()
    losses, dpred = _ssim_run(ops, kind, norm_pix, img0, img1, pred_full, mask, p, g=g, vscale=0.5)
    assert_close(losses[1:3], torch.stack([lo, lc]), 1e-4, 1e-6, f"recon {kind}")
    assert_close(losses[0], 0.5 * (lo + lc), 1e-4, 1e-6, "total")
    ref = pr.grad
    assert torch.isfinite(dpred).all()
    err = (dpred.cpu() - ref).abs().max().item()
    assert err <= 2e-3 * ref.abs().max().item(), (err, ref.abs().max().item())
    assert dpred.reshape(B2, L + 1, P)[:, 0].abs().max().item() == 0.0   # cls rows


@pytest.mark.parametrize("tag,kinds", [("s64", ("ssim", "mse_ssim")), ("s168", ("ms_ssim", "mse_ms_ssim"))])
def test_ssim_family_golden(ops, tag, kinds):
    """The committed fixture: the reference's own loss wiring around a float64 formulation of pytorch-msssim (oracle/gen_golden.py)."""
    d = np.load(os.path.join(G, "ssim_loss.npz"))
    imgs, pred, mask, p = torch.from_numpy(d[f"{tag}_imgs"]), torch.from_numpy(d[f"{tag}_pred"]), torch.from_numpy(d[f"{tag}_mask"]), int(d[f"{tag}_p"])
    N, L, P = pred.shape
    rows = torch.zeros(2 * N, L + 1, P)
    rows[:N, 1:] = pred
    rows[N:, 1:] = pred   # both "views" carry the fixture: each must reproduce it
    for kind in kinds:
        losses, dpred = _ssim_run(ops, kind, False, imgs, imgs, rows.reshape(-1, P), torch.cat([mask, mask]), p, g=1.0, vscale=1.0)
        want = float(d[f"{tag}_{kind}_masked"])
        assert_close(losses[1:3], torch.tensor([want, want]), 1e-4, 1e-6, f"golden {kind}")
        ref = torch.from_numpy(d[f"{tag}_{kind}_masked_grad"])
        for v in range(2):
            got = dpred.reshape(2 * N, L + 1, P)[v * N:(v + 1) * N, 1:].cpu()
            assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_ssim_rejects_what_the_reference_rejects(ops):
    with pytest.raises(Exception, match="larger than 160"):
        ws = torch.empty(768 * 1024, device="cuda")
        ops.ssim_fwd(5, False, ws, ws, ws.reshape(-1, 768), None, ws, ws, 2, 1, 3, 64, 16)


@pytest.mark.parametrize("kind", ["mse", "l2", "mae", "l1"])
def test_pair_loss(ops, kind):
    import csmae_oracle as O
    N, L, Td, Dd = 3, 4, 5, 32
    v = rnd(N * L, Dd, seed=80)
    emb = rnd(2 * N * Td, Dd, seed=81)
    vr, er = v.clone().requires_grad_(True), emb.clone().requires_grad_(True)
    tgt = er.reshape(2 * N, Td, Dd)[:N, 1:]
    ref = O.loss_fn(kind, tgt, vr.reshape(N, L, Dd))
    g = 1.3
    (ref * g).backward()
    partial = torch.empty(512, device="cuda")
    aview, tview = (N * L, 0, 0), (L, Td, 1)
    ops.pair_loss_fwd(kind, N * L, Dd, dev(v), aview, dev(emb), tview, partial)
    scale = 1.0 / (N * L * Dd) if kind in ("mse", "mae") else 1.0 / (N * L)
    losses = torch.zeros(8, device="cuda")
    dummy_rl, dummy_m = torch.zeros(N * L, device="cuda"), torch.ones(N * L, device="cuda")
    ops.loss_finalize(N * L, 1, dummy_rl, dummy_m, 1.0, losses, cd_partial=partial, cd_scale=scale)
    assert_close(losses[3], ref, 2e-5, 1e-7, f"pair {kind}")
    gout = torch.tensor([g], device="cuda")
    da = torch.empty(N * L, Dd, device="cuda")
    dt_acc = torch.zeros(2 * N * Td, Dd, device="cuda")
    ops.pair_loss_bwd(kind, N * L, Dd, dev(v), aview, dev(emb), tview, gout, scale, da_lp=da, dt_acc=dt_acc)
    assert_close(da, vr.grad, 1e-5, 1e-8, "pair da")
    assert_close(dt_acc, er.grad, 1e-5, 1e-8, "pair dt")


@pytest.mark.parametrize("bs", [2, 4, 128, 16])
def test_ntxent_vs_reference_golden(ops, bs):
    d = np.load(os.path.join(G, "ntxent.npz"))
    f1, f2 = torch.from_numpy(d[f"f1_{bs}"]), torch.from_numpy(d[f"f2_{bs}"])
    D = f1.shape[1]
    # latent [2N, Te=3, D] whose kept-token mean equals f (tokens f-d and f+d), cls row = junk
    delta = rnd(2 * bs, D, seed=90)
    f = torch.cat([f1, f2])
    latent = torch.stack([rnd(2 * bs, D, seed=91), f - delta, f + delta], dim=1).contiguous()
    z, inv = torch.empty(2 * bs, D, device="cuda"), torch.empty(2 * bs, device="cuda")
    E, neg, rl = torch.empty(2 * bs, 2 * bs, device="cuda"), torch.empty(2 * bs, device="cuda"), torch.empty(2 * bs, device="cuda")
    ops.ntxent_fwd(dev(latent), z, inv, E, neg, rl, bs, 3, 2)
    losses = torch.zeros(8, device="cuda")
    ops.loss_finalize(4, 1, torch.zeros(4, device="cuda"), torch.ones(4, device="cuda"), 1.0, losses, ce_rowloss=rl, ce_rows=2 * bs)
    assert_close(losses[4], torch.from_numpy(d[f"loss_{bs}"]), 1e-5, 1e-6, f"ntxent {bs}")
    gout = torch.tensor([1.0], device="cuda")
    dpool = torch.empty(2 * bs, D, device="cuda")
    ops.ntxent_bwd(z, inv, E, neg, gout, dpool, bs)
    ref = torch.cat([torch.from_numpy(d[f"g1_{bs}"]), torch.from_numpy(d[f"g2_{bs}"])])
    assert_close(dpool, ref, 2e-3, 1e-6 + 1e-3 * ref.abs().max().item(), "ntxent grad")
    dlat = torch.zeros(2 * bs * 3, D, device="cuda")
    lp = torch.empty(2 * bs * 3, D, device="cuda", dtype=torch.bfloat16)
    ops.latent_grad_finish(dlat, dpool, 0.5, lp, 2 * bs, 3)
    want = torch.stack([torch.zeros_like(ref), ref * 0.5, ref * 0.5], dim=1).reshape(-1, D)
    assert_close(dlat, want, 2e-3, 1e-6 + 1e-3 * ref.abs().max().item(), "latent grad finish")
    assert_close(lp, want, 2e-2, 1e-6 + 1e-2 * ref.abs().max().item(), "latent grad lp")


# ------------------------------------------------------------------------------------------------ optimizer side
def test_adamw_matches_torch(ops):
    sizes, wds = [1000, 37, 4096 * 3 + 5, 64], [0.05, 0.0, 0.05, 0.0]
    ps = [rnd(s, seed=100 + i) for i, s in enumerate(sizes)]
    tp = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.AdamW([dict(params=[tp[0], tp[2]], weight_decay=0.05), dict(params=[tp[1], tp[3]], weight_decay=0.0)], lr=1e-3, betas=(0.9, 0.95))
    offs = np.cumsum([0] + [(s + 3) // 4 * 4 for s in sizes])
    total = int(offs[-1])
    flat_p, flat_g = torch.zeros(total), torch.zeros(total)
    m, v = torch.zeros(total, device="cuda"), torch.zeros(total, device="cuda")
    for i, p in enumerate(ps):
        flat_p[offs[i]:offs[i] + sizes[i]] = p
    toff, tcnt, twd = [], [], []
    for i, s in enumerate(sizes):
        for o in range(0, s, 4096):
            toff.append(offs[i] + o); tcnt.append(min(4096, s - o)); twd.append(wds[i])
    toff, tcnt, twd = torch.tensor(toff, device="cuda"), torch.tensor(tcnt, dtype=torch.int32, device="cuda"), torch.tensor(twd, device="cuda")
    gp = flat_p.cuda()
    lp = torch.zeros(total, device="cuda", dtype=torch.bfloat16)
    for step in range(1, 4):
        for i, p in enumerate(tp):
            p.grad = rnd(sizes[i], seed=200 + 10 * step + i)
            flat_g[offs[i]:offs[i] + sizes[i]] = p.grad
        opt.step()
        ops.adamw(toff, tcnt, twd, gp, flat_g.cuda(), m, v, 1e-3, 0.9, 0.95, 1e-8, step, p_lp=lp)
        for i, p in enumerate(tp):
            assert_close(gp[offs[i]:offs[i] + sizes[i]], p, 1e-6, 1e-7, f"adamw step {step} tensor {i}")
    assert_close(lp[:1000], gp[:1000], 1e-2, 1e-3, "bf16 mirror")


def test_adamw_writes_the_kslab_mirrors(ops):
    """csmae_adamw with tile_ks / p_ks: the launch that steps the fp32 masters and the bf16 mirror also writes the K-slab mirror Wk[K/32][N][32] of the
    weights that have one (csmae.h csmae_gemm_ks) — byte for byte what csmae_weights_kslab makes of the bf16 mirror afterwards; tiles of tensors
    without a mirror (K = 0 rows) leave theirs untouched."""
    shapes = [(256, 128), (37,), (520, 192), (64, 64)]          # [N, K] weights with K % 32 == 0, a bias, a weight WITHOUT a mirror
    has_ks = [True, False, True, False]
    sizes = [int(np.prod(sh)) for sh in shapes]
    offs = np.cumsum([0] + [(n + 7) // 8 * 8 for n in sizes])
    total = int(offs[-1])
    p = torch.zeros(total)
    for i, n in enumerate(sizes):
        p[offs[i]:offs[i] + n] = rnd(n, seed=400 + i)
    g = rnd(total, seed=410)
    toff, tcnt, tks = [], [], []
    for i, n in enumerate(sizes):
        for o in range(0, n, 4096):
            toff.append(offs[i] + o); tcnt.append(min(4096, n - o))
            tks.append((offs[i], shapes[i][0], shapes[i][1]) if has_ks[i] else (0, 0, 0))
    toff, tcnt = torch.tensor(toff, device="cuda"), torch.tensor(tcnt, dtype=torch.int32, device="cuda")
    twd = torch.full((len(tks),), 0.05, device="cuda")
    tks = torch.tensor(tks, dtype=torch.long, device="cuda")
    gp, gg = p.cuda(), g.cuda()
    m, v = torch.zeros(total, device="cuda"), torch.zeros(total, device="cuda")
    lp = torch.zeros(total, device="cuda", dtype=torch.bfloat16)
    ks = torch.full((total,), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.adamw(toff, tcnt, twd, gp, gg, m, v, 1e-2, 0.9, 0.95, 1e-8, 1, p_lp=lp, tile_ks=tks, p_ks=ks)
    desc = torch.tensor([[offs[i], shapes[i][0], shapes[i][1]] for i in range(len(shapes)) if has_ks[i]], dtype=torch.long, device="cuda")
    ref = torch.full((total,), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.weights_kslab(desc, lp, ref)
    assert torch.equal(ks.view(torch.int16), ref.view(torch.int16))
    assert float((lp[offs[0]:offs[0] + sizes[0]].float() - gp[offs[0]:offs[0] + sizes[0]]).abs().max()) < 1e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_colsum_and_cast(ops, dtype):
    x = rnd(1237, 200, seed=110).to(dtype)
    out0 = rnd(200, seed=111)
    out = dev(out0.clone())
    ops.colsum(dev(x), out)
    assert_close(out, out0 + x.float().sum(0), 1e-5, 1e-3, "colsum")
    src = rnd(10007, seed=112)
    dst = torch.empty(10008, device="cuda", dtype=torch.bfloat16)
    ops.cast_bf16(dev(src), dst)
    assert torch.equal(dst[:10007].cpu(), src.to(torch.bfloat16))


@pytest.mark.gpu
def test_augment_u8_matches_reference_transform_chain():
    """ToTensor/Normalize/flips/RandomResizedCrop(bicubic, antialias) of util/datasets.py:120-136 in one kernel,
    against the oracle's torch restatement, for up- and down-sampling boxes, both flips, ragged image sizes and C = 3 / 4."""
    import csmae_oracle as O
    from util.gpu_input import FMOW_RGB_MEAN, FMOW_RGB_STD, GpuAugment
    g = torch.Generator().manual_seed(5)
    for C, S in ((3, 64), (4, 32)):
        mean, std = list(FMOW_RGB_MEAN)[:3] + [0.4] * (C - 3), list(FMOW_RGB_STD)[:3] + [0.2] * (C - 3)
        sizes = [(97, 120), (300, 260), (64, 64), (512, 400), (33, 47)]
        imgs = [torch.randint(0, 256, (h, w, C), generator=g, dtype=torch.uint8) for h, w in sizes]
        params = [(97, 120, 5, 9, 60, 80, 0, 0), (300, 260, 0, 0, 300, 260, 1, 0), (64, 64, 10, 12, 20, 24, 0, 1), (512, 400, 30, 20, 470, 370, 1, 1),
                  (33, 47, 0, 0, 33, 47, 1, 1)]
        aug = GpuAugment(S, mean, std)
        out = aug(imgs, params).cpu()
        assert out.shape == (len(imgs), C, S, S)
        for n, (im, pr) in enumerate(zip(imgs, params)):
            want = O.train_transform(im, pr, mean, std, S)
            err = (out[n] - want).abs().max().item()
            assert err < 2e-4, (C, S, n, err)
    # ... and against the REFERENCE's own chain (util/datasets.py:107-136 behind Dataset_fmow_rgb), tests/golden/input_transform.npz
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "input_transform.npz"))
    S, idx = int(d["S"]), [int(v) for v in d["index"]]
    aug = GpuAugment(S, [float(v) for v in d["mean"]], [float(v) for v in d["std"]])
    out = aug([torch.from_numpy(d[f"img{i}"]) for i in idx], [tuple(int(v) for v in p) for p in d["params"]]).cpu().numpy()
    assert out.shape == d["out"].shape
    for n in range(out.shape[0]):
        err = np.abs(out[n] - d["out"][n]).max()
        assert err < 2e-4, (n, d["params"][n], err)
    # the parameter draw consumes the CPU RNG in the reference's order and stays inside the image
    torch.manual_seed(0)
    for H, W in ((224, 224), (1000, 700), (50, 400)):
        for _ in range(20):
            h_, w_, i, j, h, w, hf, vf = __import__("util.gpu_input", fromlist=["x"]).sample_transform_params(H, W)
            assert (h_, w_) == (H, W) and 0 <= i and i + h <= H and 0 <= j and j + w <= W and hf in (0, 1) and vf in (0, 1)
            # a proposal of torchvision's get_params (area in [0.25, 1] of the image, aspect in [3/4, 4/3], up to the rounding of h and
            # w) or, after ten rejected proposals, its centred fallback crop
            proposal = 0.25 * 0.95 * H * W <= h * w <= H * W and 0.75 * 0.9 <= w / h <= 4.0 / 3.0 * 1.1
            centred = i == (H - h) // 2 and j == (W - w) // 2
            assert proposal or centred, (H, W, i, j, h, w)


@pytest.mark.gpu
def test_prefetch_loader_double_buffering_is_consistent():
    from util.gpu_input import GpuAugment, PrefetchLoader
    g = torch.Generator().manual_seed(9)
    batches = [([torch.randint(0, 256, (40 + 3 * k, 50 + k, 3), generator=g, dtype=torch.uint8) for _ in range(4)], torch.arange(4) + k) for k in range(5)]
    aug = GpuAugment(32)
    torch.manual_seed(3)
    got = [(x.cpu(), y) for x, y in PrefetchLoader(batches, aug)]
    torch.manual_seed(3)
    aug2 = GpuAugment(32)
    want = [(aug2(imgs).cpu(), y) for imgs, y in batches]
    assert len(got) == len(want)
    for (a, ya), (b, yb) in zip(got, want):
        assert torch.equal(ya, yb) and torch.allclose(a, b, atol=0, rtol=0)
