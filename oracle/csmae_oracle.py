"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the Cross-Scale MAE pre-training hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product (`cross-scale-mae_amd/`) never does and fails loudly without its HIP library.

Plain `torch` CPU ops in fp32, written as *functions over a state_dict* (the reference is a tree of
`nn.Module`s), each citing the reference lines it restates (paths relative to `/root/reference`).
Third-party arithmetic that is absent from the reference tree is restated from its pinned version:
timm==0.4.12 `Block`/`PatchEmbed` (`env.yml:132`), torchvision==0.15.1 `RandomResizedCrop`
(`env.yml:135`).

Parity pinning: the reference ships no tests or golden vectors.  This restatement is pinned against
outputs of the reference itself, produced in the build container by `oracle/gen_golden.py` (which
imports `/root/reference` through `oracle/ref_stubs.py`) and committed under `tests/golden/`;
`tests/test_oracle_golden.py` checks every fixture.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- configuration
def make_cfg(input_size=128, input_channels=3, patch_size=16, dim_model=1024, encoder_num_layers=24,
             encoder_num_heads=16, decoder_embed_dim=512, decoder_num_layers=8, decoder_num_heads=16,
             predictor_hidden_size=2048, loss="mse", norm_pix_loss=False, ms_range=(0.25, 0.75),
             ms_decoder_loss_reduction="sum", variant="MsLdCeCd", loss_cd=None, loss_e=None, **_):
    """Geometry of one model; defaults follow `models_mae/MAE_ViT_Baseline.py:17-49`."""
    p = int(patch_size)
    g = input_size // p
    return dict(S=input_size, C=input_channels, p=p, G=g, L=g * g, D=dim_model, He=encoder_num_heads,
                Ne=encoder_num_layers, Dd=decoder_embed_dim, Hd=decoder_num_heads, Nd=decoder_num_layers,
                Hp=predictor_hidden_size, loss=loss.lower(), norm_pix=bool(norm_pix_loss),
                ms_range=tuple(ms_range), reduction=ms_decoder_loss_reduction.lower(), variant=variant,
                loss_cd=(loss_cd or loss).lower(), loss_e=(loss_e or loss).lower())  # MAE_ViT_MsLdCeCd.py:17-18, MsLdLe.py:14-15


PRESETS = {  # `models_mae/__init__.py:23-67`
    "base": dict(dim_model=768, encoder_num_layers=12, encoder_num_heads=12, decoder_embed_dim=512,
                 decoder_num_layers=8, decoder_num_heads=16),
    "large": dict(dim_model=1024, encoder_num_layers=24, encoder_num_heads=16, decoder_embed_dim=512,
                  decoder_num_layers=8, decoder_num_heads=16),
    "huge": dict(dim_model=1280, encoder_num_layers=32, encoder_num_heads=16, decoder_embed_dim=512,
                 decoder_num_layers=8, decoder_num_heads=16),
}


# ------------------------------------------------------------------- util/pos_embed.py:16-63
def sincos_2d(dim: int, grid: int, cls_token: bool = True) -> np.ndarray:
    """float64 table; token t=h*G+w -> [sin(w*om) | cos(w*om) | sin(h*om) | cos(h*om)] ("w goes first")."""
    quarter = dim // 4
    omega = 1.0 / 10000 ** (np.arange(quarter, dtype=float) / (dim / 4.0))
    hh, ww = np.divmod(np.arange(grid * grid), grid)
    a_w = ww.astype(float)[:, None] * omega[None, :]
    a_h = hh.astype(float)[:, None] * omega[None, :]
    tab = np.concatenate([np.sin(a_w), np.cos(a_w), np.sin(a_h), np.cos(a_h)], axis=1)
    if cls_token:
        tab = np.concatenate([np.zeros((1, dim)), tab], axis=0)
    return tab


# --------------------------------------------------------------- util/lr_sched.py:9-27
def lr_at(epoch: float, lr: float, min_lr: float, warmup_epochs: float, epochs: float) -> float:
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))


# --------------------------------------------------- models_mae/MAE_ViT_Shared.py:24-55
def patchify(imgs: Tensor, p: int, c: int) -> Tensor:
    n, g = imgs.shape[0], imgs.shape[2] // p
    return imgs.reshape(n, c, g, p, g, p).permute(0, 2, 4, 3, 5, 1).reshape(n, g * g, p * p * c)


def unpatchify(x: Tensor, p: int, c: int) -> Tensor:
    n, g = x.shape[0], int(round(x.shape[1] ** 0.5))
    return x.reshape(n, g, g, p, p, c).permute(0, 5, 1, 3, 2, 4).reshape(n, c, g * p, g * p)


# --------------------------------------------------- models_mae/MAE_ViT_Shared.py:57-84
def masking_indices(noise: Tensor, keep: int):
    """argsort twice; tie rule = stable ascending (the reference's order under exact float ties is
    unspecified — SURVEY §8 a-5 — and identical to this on tie-free rows)."""
    ids_shuffle = torch.argsort(noise, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
    mask = torch.ones_like(noise)
    mask[:, :keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    return ids_shuffle, ids_restore, mask


def keep_count(L: int, mask_ratio: float) -> int:
    return int(L * (1 - mask_ratio))


# --------------------------------------------------- models_mae/MAE_ViT_Shared.py:94-163
def _reduce(per_patch: Tensor, mask: Optional[Tensor]) -> Tensor:
    return (per_patch * mask).sum() / mask.sum() if mask is not None else per_patch.mean()


def loss_fn(kind: str, target: Tensor, pred: Tensor, mask: Optional[Tensor] = None, p: Optional[int] = None, c: Optional[int] = None) -> Tensor:
    if kind == "mse":
        return _reduce(((pred - target) ** 2).mean(-1), mask)
    if kind == "l2":
        return _reduce(((pred - target) ** 2).sum(-1), mask)
    if kind == "mae":
        return _reduce((pred - target).abs().mean(-1), mask)
    if kind == "l1":
        return _reduce((pred - target).abs().sum(-1), mask)
    if kind == "bce":
        t01 = (target - target.min()) / (target.max() - target.min() + 1.0e-6)
        return _reduce(F.binary_cross_entropy_with_logits(pred, t01, reduction="none").mean(-1), mask)
    if kind in SSIM_KINDS:  # MAE_ViT_Shared.py:165-267
        if p is None or c is None:
            raise ValueError("the ssim family works on images: patch size and channel count are required")
        return ssim_family_loss(kind, target, pred, mask, p, c)
    raise ValueError(f"unknown loss {kind!r}")


# ------------------------------------------------ pytorch-msssim 0.2.1 (`env.yml:118`; call sites MAE_ViT_Shared.py:204,247)
# PARITY UNPINNED for this block: the package is not in the reference tree and not installed here, so its
# published algorithm is restated (Wang et al. 2004 / Wang et al. 2003 as implemented by pytorch-msssim 0.2.1:
# separable 11-tap gaussian, sigma 1.5, "valid" windows, K = (0.01, 0.03), five scales with 2x2 average pooling
# padded by `size % 2`).  tests/test_oracle_golden.py cross-checks it against an independent float64 scipy
# formulation and known answers; the reference-owned wiring around it (scale_01, unpatchify, mask, 1 - ssim,
# the 0.1 weight) is pinned against the reference itself with this restatement injected (tests/golden/ssim_loss.npz).
SSIM_KINDS = ("ssim", "ms_ssim", "mse_ssim", "mse_ms_ssim")
MS_SSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def gaussian_window(size: int = 11, sigma: float = 1.5) -> Tensor:
    coords = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def gaussian_filter(x: Tensor, win: Tensor) -> Tensor:
    """Depth-wise separable "valid" blur of [N,C,H,W]: along H first, then along W; an axis shorter than the
    window is left unfiltered (the package warns and skips it)."""
    ch = x.shape[1]
    out = x
    if x.shape[2] >= win.numel():
        out = F.conv2d(out, win.view(1, 1, -1, 1).repeat(ch, 1, 1, 1), groups=ch)
    if x.shape[3] >= win.numel():
        out = F.conv2d(out, win.view(1, 1, 1, -1).repeat(ch, 1, 1, 1), groups=ch)
    return out


def ssim_maps(x: Tensor, y: Tensor, data_range: float, win: Tensor, k=(0.01, 0.03)):
    """-> (ssim per (n, c), contrast-structure per (n, c)): spatial means of the two maps."""
    c1, c2 = (k[0] * data_range) ** 2, (k[1] * data_range) ** 2
    mu1, mu2 = gaussian_filter(x, win), gaussian_filter(y, win)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = gaussian_filter(x * x, win) - mu1_sq
    s2 = gaussian_filter(y * y, win) - mu2_sq
    s12 = gaussian_filter(x * y, win) - mu12
    cs_map = (2 * s12 + c2) / (s1 + s2 + c2)
    ssim_map = ((2 * mu12 + c1) / (mu1_sq + mu2_sq + c1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def ssim(x: Tensor, y: Tensor, data_range: float = 255, size_average: bool = True, win_size: int = 11,
         win_sigma: float = 1.5, nonnegative_ssim: bool = False) -> Tensor:
    per_ch, _ = ssim_maps(x, y, data_range, gaussian_window(win_size, win_sigma).to(x.dtype))
    if nonnegative_ssim:
        per_ch = torch.relu(per_ch)
    return per_ch.mean() if size_average else per_ch.mean(1)


def ms_ssim(x: Tensor, y: Tensor, data_range: float = 255, size_average: bool = True, win_size: int = 11,
            win_sigma: float = 1.5, weights=MS_SSIM_WEIGHTS) -> Tensor:
    assert min(x.shape[-2:]) > (win_size - 1) * 2 ** 4, "Image size should be larger than 160 due to the 4 downsamplings in ms-ssim"
    win = gaussian_window(win_size, win_sigma).to(x.dtype)
    w = torch.tensor(weights, dtype=x.dtype)
    terms = []
    for lvl in range(len(weights)):
        per_ch, cs = ssim_maps(x, y, data_range, win)
        if lvl < len(weights) - 1:
            terms.append(torch.relu(cs))
            pad = [s % 2 for s in x.shape[2:]]
            x = F.avg_pool2d(x, kernel_size=2, padding=pad)
            y = F.avg_pool2d(y, kernel_size=2, padding=pad)
    terms.append(torch.relu(per_ch))
    val = torch.prod(torch.stack(terms, 0) ** w.view(-1, 1, 1), dim=0)
    return val.mean() if size_average else val.mean(1)


def scale_01(x: Tensor) -> Tensor:  # MAE_ViT_Shared.py:94-95 — over the WHOLE tensor
    return (x - x.min()) / (x.max() - x.min() + 1.0e-6)


def ssim_family_loss(kind: str, target: Tensor, pred: Tensor, mask: Optional[Tensor], p: int, c: int) -> Tensor:
    """MAE_ViT_Shared.py:165-267.  `target`/`pred` are [N, L, p*p*c]; both are min-max scaled separately,
    un-patchified, multiplied by the patch mask (hard-coded 3 channels in the reference, :194-196), and
    compared with ssim(data_range=1, nonnegative_ssim=True) or ms_ssim(data_range=1).  The `mse_*` kinds add
    0.1 x that to the masked mse."""
    if kind.startswith("mse_"):
        return loss_fn("mse", target, pred, mask) + 0.1 * ssim_family_loss(kind[4:], target, pred, mask, p, c)
    t, q = unpatchify(scale_01(target), p, c), unpatchify(scale_01(pred), p, c)
    if mask is not None:
        m = unpatchify(mask.unsqueeze(-1).repeat(1, 1, p * p * 3), p, c)
        t, q = t * m, q * m
    if kind == "ssim":
        return 1 - ssim(q, t, data_range=1, size_average=True, nonnegative_ssim=True)
    return 1 - ms_ssim(q, t, data_range=1, size_average=True)


def recon_target(imgs: Tensor, p: int, c: int, norm_pix: bool) -> Tensor:
    t = patchify(imgs, p, c)
    if norm_pix:  # MAE_ViT_Shared.py:106-109 (unbiased variance)
        t = (t - t.mean(-1, keepdim=True)) / (t.var(-1, keepdim=True) + 1.0e-6) ** 0.5
    return t


# --------------------------------------------- timm 0.4.12 Block, eps 1e-6
def vit_block(x: Tensor, sd: Dict[str, Tensor], pre: str, heads: int) -> Tensor:
    B, T, D = x.shape
    hd = D // heads
    y = F.layer_norm(x, (D,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-6)
    qkv = F.linear(y, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"])
    q, k, v = qkv.reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    att = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, T, D)
    x = x + F.linear(o, sd[pre + "attn.proj.weight"], sd[pre + "attn.proj.bias"])
    y = F.layer_norm(x, (D,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-6)
    h = F.gelu(F.linear(y, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])


# --------------------------------------------- models_mae/MAE_ViT_Baseline.py:243-320
def encoder(sd, cfg, imgs: Tensor, noise: Tensor, mask_ratio: float):
    x = F.conv2d(imgs, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=cfg["p"])
    x = x.flatten(2).transpose(1, 2) + sd["encoder_pos_embed"][:, 1:, :]
    keep = keep_count(cfg["L"], mask_ratio)
    ids_shuffle, ids_restore, mask = masking_indices(noise, keep)
    x = torch.gather(x, 1, ids_shuffle[:, :keep].unsqueeze(-1).expand(-1, -1, x.shape[2]))
    cls = (sd["cls_token"] + sd["encoder_pos_embed"][:, :1, :]).expand(x.shape[0], -1, -1)
    x = torch.cat([cls, x], dim=1)
    for i in range(cfg["Ne"]):
        x = vit_block(x, sd, f"encoder.{i}.", cfg["He"])
    # MAE_ViT_Baseline.py:264 — encoder_norm(x) is evaluated and thrown away: latent is NOT normalised.
    return x, mask, ids_restore


def decoder(sd, cfg, latent: Tensor, ids_restore: Tensor):
    x = F.linear(latent, sd["decoder_embed.weight"], sd["decoder_embed.bias"])
    n, L, Dd = x.shape[0], ids_restore.shape[1], x.shape[2]
    filler = sd["mask_token"].expand(n, L + 1 - x.shape[1], Dd)
    seq = torch.cat([x[:, 1:, :], filler], dim=1)
    seq = torch.gather(seq, 1, ids_restore.unsqueeze(-1).expand(-1, -1, Dd))
    x = torch.cat([x[:, :1, :], seq], dim=1) + sd["decoder_pos_embed"]
    for i in range(cfg["Nd"]):
        x = vit_block(x, sd, f"decoder.{i}.", cfg["Hd"])
    emb = F.layer_norm(x, (Dd,), sd["decoder_norm.weight"], sd["decoder_norm.bias"], 1e-6)
    pred = F.linear(emb, sd["decoder_pred.weight"], sd["decoder_pred.bias"])[:, 1:, :]
    return pred, emb


def baseline(sd, cfg, imgs, noise, mask_ratio=0.75):
    latent, mask, ids_restore = encoder(sd, cfg, imgs, noise, mask_ratio)
    pred, emb = decoder(sd, cfg, latent, ids_restore)
    loss = loss_fn(cfg["loss"], recon_target(imgs, cfg["p"], cfg["C"], cfg["norm_pix"]), pred, mask, cfg["p"], cfg["C"])
    return dict(loss=loss, pred=pred, mask=mask, ids_restore=ids_restore, enc=latent, dec=emb)


# ------------------------------------ torchvision 0.15.1 RandomResizedCrop (MAE_ViT_MsLd.py:29-35)
def crop_box(size: int, scale, ratio=(3.0 / 4.0, 4.0 / 3.0)) -> Tuple[int, int, int, int]:
    """<=10 proposals from the global CPU torch RNG, then centre-crop fallback."""
    area = size * size
    lo, hi = math.log(ratio[0]), math.log(ratio[1])
    for _ in range(10):
        tgt = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        ar = math.exp(torch.empty(1).uniform_(lo, hi).item())
        w, h = int(round(math.sqrt(tgt * ar))), int(round(math.sqrt(tgt / ar)))
        if 0 < w <= size and 0 < h <= size:
            i = torch.randint(0, size - h + 1, size=(1,)).item()
            j = torch.randint(0, size - w + 1, size=(1,)).item()
            return i, j, h, w
    return 0, 0, size, size  # square input: in_ratio == 1 lies inside [3/4, 4/3] -> whole image


def aa_axis_weights(in_size: int, out_size: int, offset: int, full: int):
    """Per-output-index taps of ATen's separable anti-aliased bilinear filter
    (`_upsample_bilinear2d_aa`: triangle filter, support = max(scale,1), weights renormalised)."""
    scale = in_size / out_size
    support = scale if scale >= 1.0 else 1.0
    inv = 1.0 / scale if scale >= 1.0 else 1.0
    W = np.zeros((out_size, full), dtype=np.float32)
    for o in range(out_size):
        center = scale * (o + 0.5)
        lo = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - lo
        w = np.array([max(0.0, 1.0 - abs((t + lo - center + 0.5) * inv)) for t in range(n)], dtype=np.float32)
        tot = np.float32(w.sum(dtype=np.float32))
        w = w / tot
        W[o, offset + lo: offset + lo + n] = w
    return W


def crop_resize(imgs: Tensor, box, size: int) -> Tensor:
    i, j, h, w = box
    return F.interpolate(imgs[..., i:i + h, j:j + w], size=(size, size), mode="bilinear",
                         align_corners=False, antialias=True)


# ----------------------------------------------------- util/contrast_loss.py:44-101 (cos_sim=True)
def ntxent(f1: Tensor, f2: Tensor, tau: float = 0.5, eps: float = 1e-8) -> Tensor:
    bs = f1.shape[0]
    z = torch.cat([F.normalize(f1, dim=1), F.normalize(f2, dim=1)], dim=0)
    sim = torch.exp(F.cosine_similarity(z.unsqueeze(1), z.unsqueeze(0), dim=-1) / tau)
    idx = torch.arange(2 * bs)
    partner = (idx + bs) % (2 * bs)
    pos = sim[idx, partner]
    keep = torch.ones(2 * bs, 2 * bs, dtype=torch.bool)
    keep[idx, idx] = False
    keep[idx, partner] = False  # positives are excluded from the denominator (contrast_loss.py:28,94-99)
    neg = (sim * keep).sum(dim=1)
    return (-torch.log(pos / (neg + eps))).mean()


# ------------------------------------- models_mae/MLP.py:4-10 (BatchNorm1d over the TOKEN axis)
def predictor(sd, x: Tensor, bn: Optional[dict] = None, training: bool = True, momentum=0.1, eps=1e-5):
    """x [N, L, Dd].  `bn` (running_mean, running_var, num_batches_tracked) is updated in place when given."""
    u = F.linear(x, sd["predictor.0.weight"], sd["predictor.0.bias"])  # [N, L, Hp]
    if training:
        mean = u.mean(dim=(0, 2))
        var = u.var(dim=(0, 2), unbiased=False)
        if bn is not None:
            with torch.no_grad():
                cnt = u.shape[0] * u.shape[2]
                bn["running_mean"].mul_(1 - momentum).add_(momentum * mean)
                bn["running_var"].mul_(1 - momentum).add_(momentum * var * cnt / (cnt - 1))
                bn["num_batches_tracked"] += 1
    else:
        mean, var = bn["running_mean"], bn["running_var"]
    u = (u - mean[None, :, None]) / torch.sqrt(var[None, :, None] + eps)
    u = u * sd["predictor.1.weight"][None, :, None] + sd["predictor.1.bias"][None, :, None]
    return F.linear(F.relu(u), sd["predictor.3.weight"], sd["predictor.3.bias"])


# --------------------- models_mae/MAE_ViT_MsLd.py:37-77 + MsLd{Le,Cd,LeCd,CeCd}.py forward bodies
def cross_scale(sd, cfg, imgs, noise_orig, noise_crop, box, mask_ratio=0.75, bn=None, training=True, second=None):
    """Two-view forward with every loss term of the `variant` in cfg.  RNG is external: the caller
    supplies the crop box and both noise tensors (reference draw order: box, rand(N,L), rand(N,L)).
    `second`: the second view given explicitly (MAE_ViT_MsLd_PAIRED, MAE_ViT_MsLd.py:79-146) instead of the crop."""
    variant = cfg["variant"]
    crop = crop_resize(imgs, box, cfg["S"]) if second is None else second
    vo = baseline(sd, cfg, imgs, noise_orig, mask_ratio)
    vc = baseline(sd, cfg, crop, noise_crop, mask_ratio)
    out = dict(imgs_crop=crop, loss_orig=vo["loss"], loss_crop=vc["loss"], pred=vo["pred"], mask=vo["mask"],
               mask_crop=vc["mask"], ids_restore=vo["ids_restore"], ids_restore_crop=vc["ids_restore"],
               enc_orig=vo["enc"], enc_crop=vc["enc"], dec_orig=vo["dec"], dec_crop=vc["dec"], pred_crop=vc["pred"])
    loss_d = vo["loss"] + vc["loss"]
    if cfg["reduction"] == "mean":
        loss_d = loss_d / 2
    total = loss_d
    out["loss_d"] = loss_d
    if variant in ("MsLdLe", "MsLdLeCd"):  # MAE_ViT_MsLdLe.py:44 — full [N,T,D] incl. cls, no mask
        out["loss_e"] = loss_fn(cfg.get("loss_e", cfg["loss"]), vo["enc"], vc["enc"])
        total = total + out["loss_e"]
    if variant in ("MsLdCd", "MsLdLeCd", "MsLdCeCd"):  # MAE_ViT_MsLdCeCd.py:56-59 — target NOT detached
        cross_pred = predictor(sd, vc["dec"][:, 1:, :], bn=bn, training=training)   # model.eval(): BatchNorm1d normalises with the running statistics
        out["cross_pred"] = cross_pred
        out["loss_cd"] = loss_fn(cfg.get("loss_cd", cfg["loss"]), vo["dec"][:, 1:, :], cross_pred)
        total = total + out["loss_cd"]
    if variant == "MsLdCeCd":  # MAE_ViT_MsLdCeCd.py:61-69
        f1 = vo["enc"][:, 1:, :].mean(dim=1)
        f2 = vc["enc"][:, 1:, :].mean(dim=1)
        out["loss_ce"] = ntxent(f1, f2, tau=0.5)
        total = total + out["loss_ce"]
    out["loss"] = total
    return out


def forward(sd, cfg, imgs, noise_orig, noise_crop=None, box=None, mask_ratio=0.75, bn=None, training=True, second=None):
    if cfg["variant"] == "Baseline":
        return baseline(sd, cfg, imgs, noise_orig, mask_ratio)
    return cross_scale(sd, cfg, imgs, noise_orig, noise_crop, box, mask_ratio, bn, training, second)


# -------------------------------------------------- main_pretrain.py:426-427 (timm add_weight_decay)
def adamw_groups(named_params, weight_decay: float):
    decay, no_decay = [], []
    for name, p in named_params:
        if not p.requires_grad:
            continue
        (no_decay if p.ndim == 1 or name.endswith(".bias") else decay).append(p)
    return [dict(params=no_decay, weight_decay=0.0), dict(params=decay, weight_decay=weight_decay)]


FROZEN = ("encoder_pos_embed", "decoder_pos_embed")
BUFFERS = ("predictor.1.running_mean", "predictor.1.running_var", "predictor.1.num_batches_tracked")


def trainable_copy(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """float32 CPU leaf tensors with requires_grad following the reference (pos-embeds frozen)."""
    out = {}
    for k, v in sd.items():
        t = v.detach().to("cpu").clone()
        if t.is_floating_point():
            t = t.float()
            t.requires_grad_(k not in FROZEN and k not in BUFFERS)
        out[k] = t
    return out


def train_step(sd, cfg, imgs, noise_orig, noise_crop, box, opt: torch.optim.Optimizer, mask_ratio=0.75, bn=None):
    """fwd + bwd + AdamW on CPU (engine_pretrain.py:50-70 without autocast/GradScaler)."""
    opt.zero_grad(set_to_none=True)
    out = forward(sd, cfg, imgs, noise_orig, noise_crop, box, mask_ratio, bn)
    out["loss"].backward()
    opt.step()
    return out


# --------------------------------------------------------------------------------------------------------------- input step
def train_transform(img_u8, params, mean, std, S):
    """Reference training transform on one decoded image (util/datasets.py:120-136) with the random decisions given:
    ToTensor -> Normalize -> hflip -> vflip -> resized_crop(bicubic, antialias).  img_u8 [H, W, C] uint8 -> [C, S, S] float32.
    `params` = (H, W, i, j, h, w, hflip, vflip), the box in the coordinates of the flipped image (torchvision F.resized_crop)."""
    import torch.nn.functional as F
    H, W, i, j, h, w, hf, vf = [int(v) for v in params]
    x = torch.as_tensor(img_u8)[:H, :W].permute(2, 0, 1).to(torch.float32) / 255.0
    x = (x - torch.tensor(mean, dtype=torch.float32)[:, None, None]) / torch.tensor(std, dtype=torch.float32)[:, None, None]
    if hf:
        x = x.flip(-1)
    if vf:
        x = x.flip(-2)
    x = x[:, i:i + h, j:j + w]
    return F.interpolate(x[None], size=(S, S), mode="bicubic", align_corners=False, antialias=True)[0]
