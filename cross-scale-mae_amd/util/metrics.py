"""Image-comparison metrics of the evaluation tools (reference util/metrics.py): `calc_metric(x, y, name)` with
name in mse | mae | l1 | l2 | ssim | ms_ssim (aliases ssd -> l2, sad -> l1), returning a python float.

The four element-wise metrics are one-line reductions on whatever device the tensors live on, as in the reference (it calls
them on CPU copies of single images).  `ssim` / `ms_ssim` are the gaussian-window structural similarity of pytorch-msssim 0.2.1
with `data_range=1, size_average=True` (util/metrics.py:5-10,41-46): they run on the MI355X through the same HIP kernels as the
ssim loss family (`csmae_ssim_fwd` with flags 1|2: operands as they are, signed score) — there is no CPU implementation here."""
import torch


def _as_nchw(t, num_channels):
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.shape[-1] == num_channels and t.shape[1] != num_channels:  # channel-last -> [N, C, H, W] (util/metrics.py:6-9)
        t = t.permute(0, 3, 1, 2)
    return t


def _ssim_score(x, y, levels, num_channels=3):
    from csmae_hip import ops
    x, y = _as_nchw(torch.as_tensor(x), num_channels), _as_nchw(torch.as_tensor(y), num_channels)
    if x.shape != y.shape or x.shape[2] != x.shape[3]:
        raise ValueError(f"ssim metrics need two batches of square images of one shape, got {tuple(x.shape)} and {tuple(y.shape)}")
    dev = x.device if x.is_cuda else (y.device if y.is_cuda else torch.device("cuda"))
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("ssim / ms_ssim run on the MI355X only (no CPU fallback)")
    x, y = x.to(dev, torch.float32).contiguous(), y.to(dev, torch.float32).contiguous()
    N, C, S, _ = x.shape
    p = next(q for q in (16, 8, 4, 2, 1) if S % q == 0)      # any patch size that tiles the image: the planes are what is compared
    L, P = (S // p) ** 2, p * p * C
    rows = torch.zeros(N, L + 1, P, device=dev)
    rows[:, 1:] = x.reshape(N, C, S // p, p, S // p, p).permute(0, 2, 4, 3, 5, 1).reshape(N, L, P)   # patch rows of x ("nchpwq->nhwpqc")
    ws = torch.empty(ops.ssim_workspace_floats(N, C, S, p, levels), device=dev)
    terms = torch.empty(2, device=dev)
    ops.ssim_fwd(levels, False, y, None, rows.view(N * (L + 1), P), None, ws, terms, N, N, C, S, p, flags=3)
    return 1.0 - float(terms[0])


def calc_ssim(x, y, num_channels=3):
    return _ssim_score(x, y, 1, num_channels)


def calc_ms_ssim(x, y, num_channels=3):
    return _ssim_score(x, y, 5, num_channels)


METRICS_DICT = {
    "mse": {"full_name": "Mean Squared Error", "is_lower_better": True, "lambda": lambda x, y: torch.mean((x - y) ** 2).item()},
    "mae": {"full_name": "Mean Absolute Error", "is_lower_better": True, "lambda": lambda x, y: torch.mean(torch.abs(x - y)).item()},
    "l1": {"full_name": "L1 Norm", "is_lower_better": True, "lambda": lambda x, y: torch.sum(torch.abs(x - y)).item()},
    "l2": {"full_name": "L2 Norm", "is_lower_better": True, "lambda": lambda x, y: torch.sum((x - y) ** 2).item()},
    "ssim": {"full_name": "Structural Similarity Index", "is_lower_better": False, "lambda": calc_ssim},
    # needs images larger than 160 px (four 2x down-samplings of an 11-tap window)
    "ms_ssim": {"full_name": "Multi-Scale Structural Similarity Index", "is_lower_better": False, "lambda": calc_ms_ssim},
}


def calc_metric(x, y, metric_name):
    name = metric_name.lower()
    name = {"ssd": "l2", "sad": "l1"}.get(name, name)
    return METRICS_DICT[name]["lambda"](x, y)
