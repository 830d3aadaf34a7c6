"""Tensor-level launchers for the C ABI (include/csmae.h).  Every function only enqueues HIP kernels on
torch's current stream; all tensors must live on the GPU, be contiguous in their last dim, and are owned
by PyTorch.  No function here computes anything with torch ops."""
from __future__ import annotations

import ctypes

import torch

from . import BF16, EPI_ATOMIC, EPI_DGELU, EPI_GELU, EPI_NONE, EPI_RESID, F32, LOSS_KINDS, check, load

_DT = {torch.float32: F32, torch.bfloat16: BF16}
_timer = None


class KernelTimer:
    """HIP-event timing of individual launches on the stream they are enqueued on (bench.py roofline leg)."""

    def __init__(self):
        self.records, self._e0 = [], None

    def __enter__(self):
        global _timer
        _timer = self
        return self

    def __exit__(self, *a):
        global _timer
        _timer = None

    def begin(self):
        self._e0 = torch.cuda.Event(enable_timing=True)
        self._e0.record()

    def end(self, kind, work):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((kind, work, self._e0, e1))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, work, e0, e1 in self.records:
            d = out.setdefault(kind, dict(ms=0.0, work=0.0, launches=0))
            d["ms"] += e0.elapsed_time(e1)
            d["work"] += work
            d["launches"] += 1
        return out


def dt(t: torch.Tensor) -> int:
    return _DT[t.dtype]


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("csmae_hip ops need GPU tensors: the MI355X path has no CPU fallback")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_masked = {}


def cu_masked_stream(lo: int, hi: int, total_cus: int = None):
    """A torch stream whose kernels run only on the compute units of mask bits [lo, hi) (csmae_stream_create_cu_mask: bit i is CU i / 8 of
    XCD i % 8, so a range whose ends are multiples of 8 is the same CUs in every XCD).  Cached per range: queues are a finite resource."""
    if total_cus is None:
        total_cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    key = (lo, hi, total_cus, torch.cuda.current_device())
    s = _masked.get(key)
    if s is None:
        if not _masked:
            import atexit
            atexit.register(_destroy_masked)
        if not (0 <= lo < hi <= total_cus):
            raise ValueError(f"cu_masked_stream: empty or out-of-range CU range [{lo}, {hi}) of {total_cus}")
        words = (total_cus + 31) // 32
        bits = ((1 << (hi - lo)) - 1) << lo
        mask = (ctypes.c_uint32 * words)(*[(bits >> (32 * w)) & 0xFFFFFFFF for w in range(words)])
        h = ctypes.c_void_p()
        check(load().csmae_stream_create_cu_mask(words, ctypes.cast(mask, ctypes.c_void_p), ctypes.cast(ctypes.byref(h), ctypes.c_void_p)), "csmae_stream_create_cu_mask")
        s = _masked[key] = torch.cuda.ExternalStream(h.value)
    return s


def _destroy_masked():
    for s in _masked.values():
        try:
            load().csmae_stream_destroy(s.cuda_stream)
        except Exception:
            pass
    _masked.clear()


class launch_done:
    """`with launch_done(ev, st): <one op>` — the op's kernel carries the torch event `ev` as its own completion signal (csmae_next_launch_event)
    instead of a marker packet recorded behind it on stream `st`; ops whose launch site does not support that get the plain record.  `ev` must
    have been recorded once before (torch creates the HIP event at its first record).
    Contract (csrc: every CSMAE_LAUNCH site): the event goes to the FIRST CSMAE_LAUNCH kernel issued inside the block, so the wrapped ABI call
    must issue that kernel LAST — a call that launched anything behind it (a split-K fold, a trailing quantisation) would release the waiter
    before its output is complete.  True of the three users (csmae_gemm / csmae_gemm_fp8: pipelined kernel last, quantisation passes in
    front through plain launches; csmae_attn_bwd: one kernel)."""

    def __init__(self, ev, st):
        self.h, self.st = ev.cuda_event, st

    def __enter__(self):
        check(load().csmae_next_launch_event(self.h), "csmae_next_launch_event")

    def __exit__(self, *exc):
        rc = load().csmae_flush_launch_event(self.st)
        if exc[0] is None:   # (an exception already on its way out of the body is the one to report: never replace it with the flush's)
            check(rc, "csmae_flush_launch_event")
        return False


def gemm(a, b, out, *, trans_a=False, trans_b=False, bias=None, epilogue=EPI_NONE, aux=None, resid=None, splitk=1, st=None):
    """out[M,N] (+)= A(m,k) B(k,n).  a: [M,K] ([K,M] if trans_a); b: [N,K] ([K,N] if trans_b)."""
    K, M = (a.shape[0], a.shape[1]) if trans_a else (a.shape[1], a.shape[0])
    Kb, N = (b.shape[0], b.shape[1]) if trans_b else (b.shape[1], b.shape[0])
    assert K == Kb and out.shape[0] == M and out.shape[1] == N, (a.shape, b.shape, out.shape, trans_a, trans_b)
    assert a.dtype == b.dtype and a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
    assert resid is None or resid.dtype == out.dtype, "the residual epilogue reads its addend in the output's dtype"
    if aux is not None and aux.dtype == torch.uint8:   # gelu' as one byte per element (CSMAE_EPI_GELU_Q8 / CSMAE_EPI_DGELU_Q8)
        epilogue = {EPI_GELU: 6, EPI_DGELU: 7}[epilogue]
    if _timer is not None:
        _timer.begin()
    check(load().csmae_gemm(dt(a), int(trans_a), int(trans_b), M, N, K, _p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0),
                            dt(out), _p(bias), epilogue, _p(aux), aux.stride(0) if aux is not None else 0, _p(resid),
                            resid.stride(0) if resid is not None else 0, splitk, st if st is not None else stream()), "csmae_gemm")
    if _timer is not None:
        _timer.end(("gemm_bf16" if a.dtype == torch.bfloat16 else "gemm_f32") + ("_T" if trans_a else "_N") + ("N" if trans_b else "T"), 2.0 * M * N * K)
    return out


def gemm_ks(a, bk, b_plain, out, *, bias=None, epilogue=EPI_NONE, aux=None, resid=None, st=None):
    """out[M,N] = a[M,K] W^T (+ epilogue) with W [N,K] given twice: `bk` its K-slab mirror (flat bf16, Wk[K/32][N][32], weights_kslab) for the
    two-workgroups-per-CU kernel, `b_plain` in torch's layout for the shapes that kernel does not take."""
    M, K = a.shape
    N = b_plain.shape[0]
    assert b_plain.shape[1] == K and out.shape == (M, N) and a.stride(1) == 1 and out.stride(1) == 1 and b_plain.stride(1) == 1
    assert resid is None or resid.dtype == out.dtype
    assert bk.dtype == torch.bfloat16 and bk.is_contiguous() and bk.numel() >= N * K, "gemm_ks: the K-slab mirror is a flat bf16 tensor of N * K elements"
    if aux is not None and aux.dtype == torch.uint8:
        epilogue = {EPI_GELU: 6, EPI_DGELU: 7}[epilogue]
    if _timer is not None:
        _timer.begin()
    check(load().csmae_gemm_ks(dt(a), M, N, K, _p(a), a.stride(0), _p(bk), N, _p(b_plain), b_plain.stride(0), _p(out), out.stride(0), dt(out), _p(bias),
                               epilogue, _p(aux), aux.stride(0) if aux is not None else 0, _p(resid), resid.stride(0) if resid is not None else 0,
                               st if st is not None else stream()), "csmae_gemm_ks")
    if _timer is not None:
        _timer.end(("gemm_bf16" if a.dtype == torch.bfloat16 else "gemm_f32") + "_NT", 2.0 * M * N * K)
    return out


def gemm_ln_supported(M, N, K):
    """True when csmae_gemm_ln_fwd / _bwd take the shape (whole output rows in one workgroup: N <= 512, N % 64 == 0, K % 64 == 0)."""
    return load().csmae_gemm_ln_supported(M, N, K) == 1


def gemm_ln_fwd(a, bk, bias, resid, x_out, gamma, beta, y, mean, rstd, eps=1e-6, st=None):
    """x_out = a W^T + bias + resid; y = LayerNorm(x_out) * gamma + beta; mean / rstd of x_out — one kernel (csmae_gemm_ln_fwd).  `bk`: the K-slab
    mirror of W [N, K] (flat bf16, N * K elements); a, resid, x_out, y bf16 [M, .]."""
    M, K = a.shape
    N = x_out.shape[1]
    assert a.dtype == resid.dtype == x_out.dtype == y.dtype == bk.dtype == torch.bfloat16 and bk.is_contiguous() and bk.numel() >= N * K
    assert resid.shape == x_out.shape == y.shape == (M, N) and a.stride(1) == resid.stride(1) == x_out.stride(1) == y.stride(1) == 1
    assert mean.dtype == rstd.dtype == torch.float32 and mean.numel() >= M and rstd.numel() >= M and gamma.numel() == beta.numel() == N
    if _timer is not None:
        _timer.begin()
    check(load().csmae_gemm_ln_fwd(M, N, K, _p(a), a.stride(0), _p(bk), N, _p(bias), _p(resid), resid.stride(0), _p(x_out), x_out.stride(0), _p(gamma), _p(beta),
                                   eps, _p(y), y.stride(0), _p(mean), _p(rstd), st if st is not None else stream()), "csmae_gemm_ln_fwd")
    if _timer is not None:
        _timer.end("gemm_bf16_NT", 2.0 * M * N * K)


def gemm_ln_bwd(dy, w, x, mean, rstd, gamma, dres_in, dx_out, partial_ws=None, st=None):
    """dx_out = LayerNorm'(dy w; x, mean, rstd, gamma) + dres_in (w [K, N]: the layer's weight as torch stores it); partial_ws receives ceil(M / 128)
    partial rows [2, N] of dgamma / dbeta for ln_param_reduce_rows — one kernel (csmae_gemm_ln_bwd)."""
    M, K = dy.shape
    N = w.shape[1]
    assert w.shape[0] == K and x.shape == dx_out.shape == (M, N) and (dres_in is None or dres_in.shape == (M, N))
    assert dy.dtype == w.dtype == x.dtype == dx_out.dtype == torch.bfloat16 and (dres_in is None or dres_in.dtype == torch.bfloat16)
    assert dy.stride(1) == w.stride(1) == x.stride(1) == dx_out.stride(1) == 1 and mean.dtype == rstd.dtype == torch.float32
    if _timer is not None:
        _timer.begin()
    check(load().csmae_gemm_ln_bwd(M, N, K, _p(dy), dy.stride(0), _p(w), w.stride(0), _p(x), x.stride(0), _p(mean), _p(rstd), _p(gamma), _p(dres_in),
                                   dres_in.stride(0) if dres_in is not None else 0, _p(dx_out), dx_out.stride(0), _p(partial_ws),
                                   partial_ws.numel() if partial_ws is not None else 0, st if st is not None else stream()), "csmae_gemm_ln_bwd")
    if _timer is not None:
        _timer.end("gemm_bf16_NN", 2.0 * M * N * K)


def weights_kslab(desc, src, dst, max_blocks=64, st=None):
    """K-slab mirrors (csmae.h csmae_gemm_ks) of the weights in desc (int64 [count, 3] on the device: flat offset, out, in) from the bf16 mirror."""
    check(load().csmae_weights_kslab(desc.shape[0], _p(desc), max_blocks, _p(src), _p(dst), st if st is not None else stream()), "csmae_weights_kslab")


def gemm_dw(dy, x, dw, workspace, db=None, st=None):
    """dw[out,in] (fp32, contiguous) += dy[tokens,out]^T x[tokens,in] via split-K slabs in `workspace` (fp32); db[out] += colsum(dy)."""
    K, M = dy.shape
    N = x.shape[1]
    assert x.shape[0] == K and dw.shape == (M, N) and dw.is_contiguous() and dy.dtype == x.dtype
    if _timer is not None:
        _timer.begin()
    check(load().csmae_gemm_dw(dt(dy), M, N, K, _p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), _p(db), _p(workspace), workspace.numel(),
                               st if st is not None else stream()), "csmae_gemm_dw")
    if _timer is not None:
        _timer.end(("gemm_bf16" if dy.dtype == torch.bfloat16 else "gemm_f32") + "_TN", 2.0 * M * N * K)


# ---- fp8 MFMA path (BASELINE.json configs[4])
FP8_E4M3, FP8_E5M2 = 0, 1


def fp8_quantize(src, dst, amax, dq, fmt=FP8_E4M3, transpose=False, amax_next=None, st=None):
    """dst (uint8: OCP fp8 bytes) = fp8(src * FMAX / amax), per-tensor scale on the device, `dq` receives the de-quantisation factor.
    `amax` / `amax_next` are 64-slot partial maxima (FP8_SLOTS floats, zeroed by the caller before they are written).  Current scaling
    (amax_next None): `amax` first receives max|src|.  Delayed scaling: `amax` is the previous step's and is only read; this step's
    max|src| is folded into `amax_next`.
    transpose: dst is [cols, rows] (weight mirror for dX)."""
    rows, cols = src.shape
    assert dst.dtype == torch.uint8 and dst.shape == ((cols, rows) if transpose else (rows, cols)) and src.stride(1) == 1 and dst.stride(1) == 1
    s = st if st is not None else stream()
    if amax_next is None:
        check(load().csmae_fp8_amax(dt(src), rows, cols, _p(src), src.stride(0), _p(amax), s), "csmae_fp8_amax")
    check(load().csmae_fp8_quantize(dt(src), fmt, int(transpose), rows, cols, _p(src), src.stride(0), _p(dst), dst.stride(0), _p(amax), _p(dq),
                                    _p(amax_next), s), "csmae_fp8_quantize")


FP8_SLOTS = 64   # an amax is 64 partial maxima (see csrc/fp8.hip)


def fp8_weights(desc, p, w8, w8t, amax, dq, st=None):
    """All fp8 weight mirrors in three launches: desc int64 [count, 3] = (offset in p / w8 / w8t, out, in); amax [count, 64] zeroed, dq [count]."""
    check(load().csmae_fp8_weights(desc.shape[0], _p(desc), _p(p), _p(w8), _p(w8t), _p(amax), _p(dq), st if st is not None else stream()), "csmae_fp8_weights")


def gemm_fp8(a8, b8, out, dq_a, dq_b, *, a_fmt=FP8_E4M3, bias=None, epilogue=EPI_NONE, aux=None, resid=None, emit=None, skip_out=False, st=None):
    """out[M,N] = dq_a * dq_b * a8[M,K] b8[N,K]^T (+ epilogue): both operands K-contiguous fp8 bytes (uint8 tensors).
    emit = (q_out uint8 [M,N], fmt, amax_prev [64], amax_next [64], dq [1]): the epilogue also writes `out` as fp8 bytes for the next GEMM.
    skip_out (with emit): `out` only names shape / dtype / row stride — the kernel writes the fp8 copy (and the gelu' codes) and not the bf16 tensor."""
    assert not skip_out or emit is not None
    M, K = a8.shape
    N = b8.shape[0]
    assert a8.dtype == torch.uint8 and b8.dtype == torch.uint8 and b8.shape[1] == K and out.shape == (M, N)
    assert resid is None or resid.dtype == out.dtype
    if aux is not None and aux.dtype == torch.uint8:
        epilogue = {EPI_GELU: 6, EPI_DGELU: 7}[epilogue]
    if _timer is not None:
        _timer.begin()
    check(load().csmae_gemm_fp8(a_fmt, M, N, K, _p(a8), a8.stride(0), _p(b8), b8.stride(0), None if skip_out else _p(out), out.stride(0), dt(out), _p(bias), epilogue,
                                _p(aux), aux.stride(0) if aux is not None else 0, _p(resid), resid.stride(0) if resid is not None else 0,
                                _p(dq_a), _p(dq_b), _p(emit[0]) if emit else None, emit[0].stride(0) if emit else 0, emit[1] if emit else 0,
                                _p(emit[2]) if emit else None, _p(emit[3]) if emit else None, _p(emit[4]) if emit else None,
                                st if st is not None else stream()), "csmae_gemm_fp8")
    if _timer is not None:
        _timer.end("gemm_fp8_NT", 2.0 * M * N * K)
    return out


class DwGroup:
    """Argument block of one grouped weight-gradient launch (csmae_gemm_dw_group): the host arrays of device pointers are built once
    — the engine's workspace and gradient buffers do not move — and re-used every step."""

    def __init__(self, products, workspace):
        """products: [(dy [K, >=M], x [K, >=N], dw [M, N] fp32 contiguous, db [M] fp32 or None)] with one K."""
        n = len(products)
        self.K = products[0][0].shape[0]
        self.dtype = dt(products[0][0])
        self.keep = (products, workspace)
        for dy, x, dw, db in products:
            assert dy.shape[0] == self.K and x.shape[0] == self.K and dw.is_contiguous() and dy.dtype == x.dtype and dy.stride(1) == 1 and x.stride(1) == 1
        VP, LL = ctypes.c_void_p * n, ctypes.c_longlong * n
        self.n = n
        self.dY, self.X = VP(*[_p(q[0]) for q in products]), VP(*[_p(q[1]) for q in products])
        self.dW, self.dB = VP(*[_p(q[2]) for q in products]), VP(*[_p(q[3]) for q in products])
        self.ldy, self.ldx = LL(*[q[0].stride(0) for q in products]), LL(*[q[1].stride(0) for q in products])
        self.M, self.N = LL(*[q[2].shape[0] for q in products]), LL(*[q[2].shape[1] for q in products])
        self.flops = sum(2.0 * q[2].shape[0] * q[2].shape[1] * self.K for q in products)
        self.ws, self.ws_n = _p(workspace), workspace.numel()

    def launch(self, slots=0, st=None):
        if _timer is not None:
            _timer.begin()
        check(load().csmae_gemm_dw_group(self.dtype, self.n, self.K, self.dY, self.ldy, self.X, self.ldx, self.dW, self.dB, self.M, self.N, slots,
                                         self.ws, self.ws_n, st if st is not None else stream()), "csmae_gemm_dw_group")
        if _timer is not None:
            _timer.end(("gemm_bf16" if self.dtype == BF16 else "gemm_f32") + "_TN", self.flops)


class DwGroup8:
    """Argument block of one grouped fp8 weight-gradient launch (csmae_gemm_dw_group_fp8): products [(dy8 [K, >=M] uint8 e5m2, dq_y [1], x8 [K, >=N] uint8 e4m3,
    dq_x [1], dw [M, N] fp32 contiguous, db [M] fp32 or None)] over one K."""

    def __init__(self, products, workspace):
        n = len(products)
        self.K = products[0][0].shape[0]
        self.keep = (products, workspace)
        for dy, dqy, x, dqx, dw, db in products:
            assert dy.dtype == x.dtype == torch.uint8 and dy.shape[0] == x.shape[0] == self.K and dy.stride(1) == x.stride(1) == 1 and dw.is_contiguous()
            assert dqy.dtype == dqx.dtype == torch.float32 and dqy.numel() >= 1 and dqx.numel() >= 1
        VP, LL = ctypes.c_void_p * n, ctypes.c_longlong * n
        self.n = n
        self.dY, self.X = VP(*[_p(q[0]) for q in products]), VP(*[_p(q[2]) for q in products])
        self.dqy, self.dqx = VP(*[_p(q[1]) for q in products]), VP(*[_p(q[3]) for q in products])
        self.dW, self.dB = VP(*[_p(q[4]) for q in products]), VP(*[_p(q[5]) for q in products])
        self.ldy, self.ldx = LL(*[q[0].stride(0) for q in products]), LL(*[q[2].stride(0) for q in products])
        self.M, self.N = LL(*[q[4].shape[0] for q in products]), LL(*[q[4].shape[1] for q in products])
        self.flops = sum(2.0 * q[4].shape[0] * q[4].shape[1] * self.K for q in products)
        self.ws, self.ws_n = _p(workspace), workspace.numel()

    def launch(self, slots=0, st=None):
        if _timer is not None:
            _timer.begin()
        check(load().csmae_gemm_dw_group_fp8(self.n, self.K, self.dY, self.ldy, self.dqy, self.X, self.ldx, self.dqx, self.dW, self.dB, self.M, self.N, slots,
                                             self.ws, self.ws_n, st if st is not None else stream()), "csmae_gemm_dw_group_fp8")
        if _timer is not None:
            _timer.end("gemm_fp8_TN", self.flops)


def attn_resident(dtype_code, T, hd):
    """True when (dtype, T, head_dim) runs the LDS-resident MFMA attention kernels (the ones that can emit an fp8 copy of their output)."""
    return load().csmae_attn_resident(dtype_code, T, hd) == 1


def attn_fwd(qkv, out, lse, B, T, H, hd, emit=None, st=None):
    """emit = (q_out uint8 [B*T, H*hd], fmt, amax_prev [64], amax_next [64], dq [1]): also write `out` as fp8 bytes for attn.proj's GEMM."""
    if emit is None:
        check(load().csmae_attn_fwd(dt(qkv), B, T, H, hd, _p(qkv), _p(out), _p(lse), st if st is not None else stream()), "csmae_attn_fwd")
    else:
        check(load().csmae_attn_fwd_q(dt(qkv), B, T, H, hd, _p(qkv), _p(out), _p(lse), *_emit_args(emit), st if st is not None else stream()), "csmae_attn_fwd_q")


def attn_bwd(qkv, out, dout, lse, dqkv, B, T, H, hd, emit=None, skip_out=False, st=None):
    """skip_out (with emit): only the fp8 copy of dqkv is written."""
    assert not skip_out or emit is not None
    if emit is None:
        check(load().csmae_attn_bwd(dt(qkv), B, T, H, hd, _p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), st if st is not None else stream()), "csmae_attn_bwd")
    else:
        check(load().csmae_attn_bwd_q(dt(qkv), B, T, H, hd, _p(qkv), _p(out), _p(dout), _p(lse), None if skip_out else _p(dqkv), *_emit_args(emit), st if st is not None else stream()),
              "csmae_attn_bwd_q")


def _emit_args(emit):
    """emit = (q_out uint8, fmt, amax_prev [64], amax_next [64], dq [1]) or None -> the five trailing C arguments"""
    if emit is None:
        return None, 0, None, None, None
    return _p(emit[0]), emit[1], _p(emit[2]), _p(emit[3]), _p(emit[4])


def layernorm_fwd(x, gamma, beta, y, mean, rstd, y32=None, eps=1e-6, emit=None, skip_out=False, st=None):
    """skip_out (with emit): only the fp8 copy of y (and the row statistics) is written."""
    M, D = x.shape
    assert not skip_out or emit is not None
    check(load().csmae_layernorm_fwd(dt(x), dt(y), M, D, _p(x), _p(gamma), _p(beta), eps, None if skip_out else _p(y), _p(y32), _p(mean), _p(rstd), *_emit_args(emit),
                                     st if st is not None else stream()), "csmae_layernorm_fwd")


def layernorm_bwd(dy, x, mean, rstd, gamma, dx_out, dgamma, dbeta, dres_in=None, dx_lp=None, partial_ws=None, emit=None, st=None):
    """x, dres_in and dx_out share one dtype (the residual stream's).  dgamma=None with a workspace: the parameter-gradient partial
    rows stay in `partial_ws` for ln_param_reduce."""
    M, D = x.shape
    assert dx_out.dtype == x.dtype and (dres_in is None or dres_in.dtype == x.dtype)
    lp = dt(dx_lp) if dx_lp is not None else dt(dy)
    check(load().csmae_layernorm_bwd(dt(dy), dt(x), lp, M, D, _p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres_in), _p(dx_out), _p(dx_lp),
                                     _p(dgamma), _p(dbeta), _p(partial_ws), partial_ws.numel() if partial_ws is not None else 0, *_emit_args(emit),
                                     st if st is not None else stream()), "csmae_layernorm_bwd")


def ln_param_reduce(count, M, D, partials, goff, gbase, st=None):
    """partials [>= count, slice] fp32 (row k = LayerNorm k's partial rows), goff [count, 2] int64 offsets of dgamma / dbeta in gbase."""
    check(load().csmae_ln_param_reduce(count, M, D, _p(partials), partials.stride(0), partials.shape[1], _p(gbase), _p(goff),
                                       st if st is not None else stream()), "csmae_ln_param_reduce")


def ln_param_reduce_rows(count, rows, D, partials, goff, gbase, st=None):
    """The same fold for LayerNorms whose partial rows were left by gemm_ln_bwd: `rows` = ceil(M / 128) rows per LayerNorm."""
    check(load().csmae_ln_param_reduce_rows(count, rows, D, _p(partials), partials.stride(0), _p(gbase), _p(goff), st if st is not None else stream()),
          "csmae_ln_param_reduce_rows")


def bnrelu_fwd(u, gamma, beta, r, mean, rstd, N, L, running_mean=None, running_var=None, nbt=None, eps=1e-5, momentum=0.1, training=True, st=None):
    check(load().csmae_bnrelu_fwd(dt(u), N, L, u.shape[1], _p(u), _p(gamma), _p(beta), eps, momentum, _p(r), _p(mean), _p(rstd),
                                  _p(running_mean), _p(running_var), _p(nbt), int(training), st if st is not None else stream()), "csmae_bnrelu_fwd")


def bnrelu_bwd(u, dr, gamma, beta, mean, rstd, du, dgamma, dbeta, N, L, st=None):
    check(load().csmae_bnrelu_bwd(dt(u), N, L, u.shape[1], _p(u), _p(dr), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(du), _p(dgamma),
                                  _p(dbeta), st if st is not None else stream()), "csmae_bnrelu_bwd")


def crop_resize(src, dst, box, st=None):
    S = src.shape[-1]
    check(load().csmae_crop_resize(src.numel() // (S * S), S, _p(src), _p(dst), _p(box), st if st is not None else stream()), "csmae_crop_resize")


def mask_sort(noise, keep, ids_restore, mask, ids_keep, ids_shuffle=None, st=None):
    rows, L = noise.shape
    check(load().csmae_mask_sort(rows, L, keep, _p(noise), _p(ids_restore), _p(mask), _p(ids_keep), _p(ids_shuffle),
                                 st if st is not None else stream()), "csmae_mask_sort")


def patch_gather(img0, img1, ids_keep, out, N, C, S, p, keep, st=None):
    check(load().csmae_patch_gather(dt(out), out.shape[0], keep, N, C, S, p, _p(img0), _p(img1), _p(ids_keep), _p(out), out.stride(0),
                                    st if st is not None else stream()), "csmae_patch_gather")


def embed_assemble(tok, pos, cls, ids_keep, x, B2, keep, st=None):
    check(load().csmae_embed_assemble(dt(x), B2, keep, x.shape[-1], _p(tok), _p(pos), _p(cls), _p(ids_keep), _p(x),
                                      st if st is not None else stream()), "csmae_embed_assemble")


def embed_assemble_bwd(dx, dtok, dcls, B2, keep, st=None):
    check(load().csmae_embed_assemble_bwd(dt(dx), dt(dtok), B2, keep, dx.shape[-1], _p(dx), _p(dtok), _p(dcls), st if st is not None else stream()),
          "csmae_embed_assemble_bwd")


def unshuffle_fwd(z, mask_token, dpos, ids_restore, xd, B2, L, keep, st=None):
    check(load().csmae_unshuffle_fwd(dt(xd), B2, L, keep, xd.shape[-1], _p(z), _p(mask_token), _p(dpos), _p(ids_restore), _p(xd),
                                     st if st is not None else stream()), "csmae_unshuffle_fwd")


def unshuffle_bwd(dxd, ids_restore, dz, dmask_token, B2, L, keep, st=None):
    check(load().csmae_unshuffle_bwd(dt(dxd), dt(dz), B2, L, keep, dxd.shape[-1], _p(dxd), _p(ids_restore), _p(dz), _p(dmask_token),
                                     st if st is not None else stream()), "csmae_unshuffle_bwd")


def rows_gather(src, dst, group, gstride, off, st=None):
    check(load().csmae_rows_gather(dt(dst), dst.shape[0], dst.shape[1], _p(src), group, gstride, off, _p(dst), st if st is not None else stream()),
          "csmae_rows_gather")


def rows_gather_idx(x, ids, keep, out, st=None):
    """out[n, k, :] = x[n, ids[n, k], :] for k < keep (x [N, L, D] fp32, ids [N, >= keep] int32, out [N, keep, D] fp32)."""
    N, L, D = x.shape
    assert x.dtype == out.dtype == torch.float32 and ids.dtype == torch.int32 and x.is_contiguous() and out.is_contiguous() and ids.stride(1) == 1
    assert out.shape == (N, keep, D) and ids.shape[0] == N and ids.shape[1] >= keep
    check(load().csmae_rows_gather_idx(N, L, keep, D, _p(x), _p(ids), ids.stride(0), _p(out), st if st is not None else stream()), "csmae_rows_gather_idx")
    return out


def rows_scatter_add2(a, scale_a, off_a, b, scale_b, off_b, dst, group, gstride, st=None):
    """dst[view(r) + off_a] += scale_a * a[r]; dst[view(r) + off_b] += scale_b * b[r]   (view(r) = (r // group) * gstride + r % group)."""
    assert a.shape == b.shape and a.dtype == b.dtype and dst.dtype == torch.float32
    check(load().csmae_rows_scatter_add2(dt(a), a.shape[0], a.shape[1], _p(a), scale_a, off_a, _p(b), scale_b, off_b, group, gstride, _p(dst),
                                         st if st is not None else stream()), "csmae_rows_scatter_add2")


def spec_fixup(g, bufs, tmp, dgamma, dbeta, st=None):
    """The speculative cross-decoder backward's fix-up (csmae.h csmae_spec_fixup): dgamma / dbeta += g * tmp[0] / tmp[1]; the three bf16 tensors *= g (no-op for g == 1)."""
    b0, b1, b2 = bufs
    assert all(b.dtype == torch.bfloat16 and b.is_contiguous() for b in bufs) and tmp.shape[0] == 2 and tmp.dtype == torch.float32
    check(load().csmae_spec_fixup(_p(g), _p(b0), b0.numel(), _p(b1), b1.numel(), _p(b2), b2.numel(), _p(tmp[0]), _p(tmp[1]), _p(dgamma), _p(dbeta), tmp.shape[1],
                                  st if st is not None else stream()), "csmae_spec_fixup")


def rows_scatter_add(src, dst, group, gstride, off, scale=1.0, st=None):
    check(load().csmae_rows_scatter_add(dt(src), src.shape[0], src.shape[1], _p(src), scale, group, gstride, off, _p(dst),
                                        st if st is not None else stream()), "csmae_rows_scatter_add")


def target_minmax(img0, img1, scratch, out, B2, N, C, S, p, norm_pix, st=None):
    check(load().csmae_target_minmax(int(norm_pix), B2, N, C, S, p, _p(img0), _p(img1), _p(scratch), _p(out), st if st is not None else stream()),
          "csmae_target_minmax")


def recon_loss_fwd(kind, norm_pix, img0, img1, pred, minmax, rowloss, B2, N, C, S, p, mask=None, st=None):
    """pred: fp32 or bf16 [B2 * (L + 1), >= P]; mask (optional, [B2 * L] fp32): patches with mask 0 are skipped (rowloss 0)."""
    check(load().csmae_recon_loss_fwd(LOSS_KINDS[kind], int(norm_pix), dt(pred), B2, N, C, S, p, _p(img0), _p(img1), _p(pred), pred.stride(0), _p(minmax),
                                      _p(mask), _p(rowloss), st if st is not None else stream()), "csmae_recon_loss_fwd")


def recon_loss_bwd(kind, norm_pix, img0, img1, pred, minmax, mask, losses, gout, vscale, dpred, B2, N, C, S, p, extra=None, st=None):
    check(load().csmae_recon_loss_bwd(LOSS_KINDS[kind], int(norm_pix), dt(dpred), dt(pred), B2, N, C, S, p, _p(img0), _p(img1), _p(pred), pred.stride(0),
                                      _p(minmax), _p(mask), _p(losses), _p(gout), vscale, _p(extra), _p(dpred), dpred.stride(0),
                                      st if st is not None else stream()), "csmae_recon_loss_bwd")


# ---- ssim family (MAE_ViT_Shared.py:165-267)
def ssim_workspace_floats(B2, C, S, p, levels):
    n = ctypes.c_longlong(0)
    check(load().csmae_ssim_workspace_floats(B2, C, S, p, levels, ctypes.byref(n)), "csmae_ssim_workspace_floats")
    return n.value


def ssim_fwd(levels, norm_pix, img0, img1, pred, mask, ws, terms, B2, N, C, S, p, flags=0, st=None):
    check(load().csmae_ssim_fwd(levels, flags, int(norm_pix), B2, N, C, S, p, _p(img0), _p(img1), _p(pred), pred.stride(0), _p(mask), _p(ws), _p(terms),
                                st if st is not None else stream()), "csmae_ssim_fwd")


def ssim_apply(pure, views, weight, recon_scale, terms, losses, st=None):
    check(load().csmae_ssim_apply(int(pure), views, weight, recon_scale, _p(terms), _p(losses), st if st is not None else stream()), "csmae_ssim_apply")


def ssim_bwd(levels, pred, mask, gout, scale, ws, extra, B2, N, C, S, p, st=None):
    check(load().csmae_ssim_bwd(levels, B2, N, C, S, p, _p(pred), pred.stride(0), _p(mask), _p(gout), scale, _p(ws), _p(extra),
                                st if st is not None else stream()), "csmae_ssim_bwd")


def pair_loss_fwd(kind, rows, D, a, aview, t, tview, partial, st=None):
    check(load().csmae_pair_loss_fwd(LOSS_KINDS[kind], rows, D, _p(a), *aview, _p(t), *tview, _p(partial), st if st is not None else stream()),
          "csmae_pair_loss_fwd")


def pair_loss_bwd(kind, rows, D, a, aview, t, tview, gout, coef, da_lp=None, da_acc=None, dt_acc=None, lp_dtype=F32, st=None):
    lp = dt(da_lp) if da_lp is not None else lp_dtype
    check(load().csmae_pair_loss_bwd(LOSS_KINDS[kind], lp, rows, D, _p(a), *aview, _p(t), *tview, _p(gout), coef, _p(da_lp), _p(da_acc), _p(dt_acc),
                                     st if st is not None else stream()), "csmae_pair_loss_bwd")


def ntxent_fwd(latent, z, inv_norm, E, neg, rowloss, N, Te, keep, tau=0.5, eps=1e-8, st=None):
    check(load().csmae_ntxent_fwd(N, Te, keep, latent.shape[-1], _p(latent), tau, eps, _p(z), _p(inv_norm), _p(E), _p(neg), _p(rowloss),
                                  st if st is not None else stream()), "csmae_ntxent_fwd")


def ntxent_bwd(z, inv_norm, E, neg, gout, dpool, N, tau=0.5, eps=1e-8, st=None):
    check(load().csmae_ntxent_bwd(N, z.shape[-1], _p(z), _p(inv_norm), _p(E), _p(neg), tau, eps, _p(gout), _p(dpool), st if st is not None else stream()),
          "csmae_ntxent_bwd")


def latent_grad_finish(dlat, dpool, inv_keep, dlat_lp, B2, Te, st=None):
    lp = dt(dlat_lp) if dlat_lp is not None else F32
    check(load().csmae_latent_grad_finish(lp, B2, Te, dlat.shape[-1], _p(dlat), _p(dpool), inv_keep, _p(dlat_lp), st if st is not None else stream()),
          "csmae_latent_grad_finish")


def loss_finalize(per_view, views, rowloss, mask, recon_scale, losses, cd_partial=None, cd_scale=0.0, e_partial=None, e_scale=0.0,
                  ce_rowloss=None, ce_rows=0, st=None):
    check(load().csmae_loss_finalize(per_view, views, _p(rowloss), _p(mask), recon_scale, _p(cd_partial), cd_scale, _p(e_partial), e_scale,
                                     _p(ce_rowloss), ce_rows, _p(losses), st if st is not None else stream()), "csmae_loss_finalize")


def adamw(tile_off, tile_cnt, tile_wd, p, g, m, v, lr, beta1, beta2, eps, step, p_lp=None, gate=None, tile_ks=None, p_ks=None, st=None):
    """One fused AdamW step over the tiles; `step` (1-based) sets the bias corrections 1 - beta^step; a non-finite `gate` (device
    scalar) turns the launch into a no-op.  tile_ks (int64 [ntiles, 3]: weight offset, N, K; K = 0 none) + p_ks: also write the K-slab mirrors."""
    check(load().csmae_adamw(tile_off.numel(), _p(tile_off), _p(tile_cnt), _p(tile_wd), _p(p), _p(g), _p(m), _p(v), float(lr), float(beta1),
                             float(beta2), float(eps), 1.0 - beta1 ** step, 1.0 - beta2 ** step, _p(p_lp), _p(gate), _p(tile_ks), _p(p_ks),
                             st if st is not None else stream()), "csmae_adamw")


def adamw_fp8(tile8, wd, p, g, m, v, lr, beta1, beta2, eps, step, p_lp, gate, w8, w8t, amax_prev, amax_next, dq, st=None):
    """The fused AdamW step over 64 x 64 sub-blocks of fp8-mirrored weights (tile8 int64 [ntiles, 6]), writing W8 / W8^T with delayed scaling (csmae_adamw_fp8)."""
    check(load().csmae_adamw_fp8(tile8.shape[0], _p(tile8), float(wd), _p(p), _p(g), _p(m), _p(v), float(lr), float(beta1), float(beta2), float(eps),
                                 1.0 - beta1 ** step, 1.0 - beta2 ** step, _p(p_lp), _p(gate), _p(w8), _p(w8t), _p(amax_prev), _p(amax_next), _p(dq),
                                 st if st is not None else stream()), "csmae_adamw_fp8")


def gate_accumulate(loss, slot, accumulate, st=None):
    check(load().csmae_gate_accumulate(_p(loss), _p(slot), int(accumulate), st if st is not None else stream()), "csmae_gate_accumulate")


def clip_grad_norm(g, max_norm, scratch, out, st=None):
    """out[0] = ||g||_2, out[1] = min(1, max_norm / (norm + 1e-6)); g *= out[1] in place (max_norm <= 0: norm only)."""
    check(load().csmae_clip_grad_norm(g.numel(), _p(g), float(max_norm), _p(scratch), _p(out), st if st is not None else stream()),
          "csmae_clip_grad_norm")


def augment_u8(src, meta, mean, inv_std, dst, st=None):
    """src [N, Hmax, Wmax, C] uint8, meta [N, 8] int32 {H, W, i, j, h, w, hflip, vflip}, dst [N, C, S, S] fp32 (util/datasets.py:120-136)."""
    N, Hmax, Wmax, C = src.shape
    assert src.dtype == torch.uint8 and src.is_contiguous() and meta.dtype == torch.int32 and meta.shape == (N, 8) and dst.shape[:2] == (N, C)
    check(load().csmae_augment_u8(N, C, Hmax, Wmax, dst.shape[-1], _p(src), _p(meta), _p(mean), _p(inv_std), _p(dst),
                                  st if st is not None else stream()), "csmae_augment_u8")


def cast_bf16(src, dst, st=None):
    check(load().csmae_cast_f32_to_bf16(src.numel(), _p(src), _p(dst), st if st is not None else stream()), "csmae_cast_f32_to_bf16")


def cast_f32(src, dst, st=None):
    check(load().csmae_cast_bf16_to_f32(src.numel(), _p(src), _p(dst), st if st is not None else stream()), "csmae_cast_bf16_to_f32")


def colsum(x, out, st=None):
    check(load().csmae_colsum(dt(x), x.shape[0], x.shape[1], _p(x), x.stride(0), _p(out), st if st is not None else stream()), "csmae_colsum")


__all__ = [n for n in dir() if not n.startswith("_")]
