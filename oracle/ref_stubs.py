"""DEV-ONLY TEST INFRASTRUCTURE — never imported by the product path.

Stand-in modules that let the *reference* (`/root/reference`, read-only) be imported in the
build container so `oracle/gen_golden.py` can emit golden vectors.  The reference depends on
packages that are not installed here (timm 0.4.12, torchvision 0.15.1, xformers, pytorch_msssim,
wandb) and on four in-tree modules that were never committed (`models_mae/__init__.py:16-19`).

Only the two pieces of third-party *arithmetic* that sit on the hot path are restated:

* timm 0.4.12 ``vision_transformer.Block`` / ``PatchEmbed`` (call sites
  `models_mae/MAE_ViT_Baseline.py:75-77,160-188`): pre-norm block, fused qkv Linear, softmax
  attention scaled by head_dim**-0.5, exact-erf GELU MLP.
* torchvision 0.15.1 ``transforms.RandomResizedCrop`` tensor path (call site
  `models_mae/MAE_ViT_MsLd.py:29-35`): <=10 box proposals from the CPU torch RNG, centre-crop
  fallback, then bilinear anti-aliased resize.

* pytorch-msssim 0.2.1 ``ssim`` / ``ms_ssim`` (call sites `models_mae/MAE_ViT_Shared.py:204,247`), in a
  formulation that is deliberately *different* from the one in `csmae_oracle.py` (float64, dense 2-D window,
  explicit 2x2 block means) so that the two check each other through the golden fixture.

Everything else is an inert placeholder.  This file is never shipped to / needed on the GPU box.
"""
from __future__ import annotations

import math
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"


# ----------------------------------------------------------------------------- timm 0.4.12
class _Mlp(nn.Module):
    def __init__(self, in_features, hidden_features, act_layer, drop):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class _Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias, attn_drop, proj_drop):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        B, T, C = x.shape
        qkv = self.qkv(x).reshape(B, T, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = (q @ k.transpose(-2, -1)) * self.scale
        a = self.attn_drop(a.softmax(dim=-1))
        x = (a @ v).transpose(1, 2).reshape(B, T, C)
        return self.proj_drop(self.proj(x))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        assert drop_path == 0.0, "stub supports drop_path=0 only (reference default)"
        self.norm1 = norm_layer(dim)
        self.attn = _Attention(dim, num_heads, qkv_bias, attn_drop, drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio), act_layer, drop)

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        return x + self.drop_path(self.mlp(self.norm2(x)))


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.Identity()

    def forward(self, x):
        assert x.shape[2] == self.img_size[0] and x.shape[3] == self.img_size[1]
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))


# ----------------------------------------------------------------------- torchvision 0.15.1
def rrc_get_params(height, width, scale, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """Box proposal loop of torchvision 0.15.1 RandomResizedCrop.get_params (consumes the global CPU torch RNG): the log-ratio bounds and
    the exponential are float32 tensor operations there."""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        ar = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        w = int(round(math.sqrt(target * ar)))
        h = int(round(math.sqrt(target / ar)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,)).item()
            j = torch.randint(0, width - w + 1, size=(1,)).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


class InterpolationMode:
    """torchvision.transforms.InterpolationMode: the two members the reference names (util/datasets.py:111, MAE_ViT_MsLd.py default)."""
    BILINEAR = "bilinear"
    BICUBIC = "bicubic"


class RandomResizedCrop(nn.Module):
    """torchvision 0.15.1 RandomResizedCrop on tensors: get_params, then F.resized_crop = crop + `interpolate(mode, align_corners=False,
    antialias)` on the float image (no clamp for float inputs).  Used by the models (bilinear, MAE_ViT_MsLd.py:29-35) and by the
    dataset transform (bicubic, util/datasets.py:124-131)."""
    last_box = None  # recorded for fixtures

    def __init__(self, size, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), interpolation=InterpolationMode.BILINEAR, antialias=None, **_):
        super().__init__()
        size = (int(size), int(size)) if isinstance(size, (int, float)) else tuple(size)
        self.size, self.scale, self.ratio, self.interpolation, self.antialias = size, tuple(scale), tuple(ratio), interpolation, antialias

    def forward(self, img):
        i, j, h, w = rrc_get_params(img.shape[-2], img.shape[-1], self.scale, self.ratio)
        RandomResizedCrop.last_box = (i, j, h, w)
        batched = img.dim() == 4
        x = img[..., i:i + h, j:j + w]
        y = F.interpolate(x if batched else x[None], size=self.size, mode=self.interpolation, align_corners=False, antialias=bool(self.antialias))
        return y if batched else y[0]


class ToTensor:
    """torchvision F.to_tensor for a PIL image of mode RGB / L: uint8 HWC -> float32 CHW / 255."""

    def __call__(self, pic):
        import numpy as np
        a = np.array(pic, np.uint8, copy=True)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(a).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


class Normalize:
    def __init__(self, mean, std, inplace=False):
        self.mean, self.std = mean, std

    def __call__(self, t):
        mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return t.clone().sub_(mean).div_(std)


class _RandomFlip(nn.Module):
    """RandomHorizontalFlip / RandomVerticalFlip: `if torch.rand(1) < p: flip` (one draw from the global CPU generator each)."""
    dim = -1
    last = None  # recorded for fixtures

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def forward(self, img):
        hit = bool(torch.rand(1) < self.p)
        type(self).last = hit
        return img.flip(self.dim) if hit else img


class RandomHorizontalFlip(_RandomFlip):
    dim = -1


class RandomVerticalFlip(_RandomFlip):
    dim = -2


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


def _eval_only(name):
    def ctor(*a, **k):
        raise NotImplementedError(f"torchvision.transforms.{name}: the eval transform (util/datasets.py:138-158) is outside the hot path — not restated")
    return ctor


# --------------------------------------------------------------------------------- install
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# ----------------------------------------------------------------------- pytorch-msssim 0.2.1
def _win2d(size, sigma):
    c = torch.arange(size, dtype=torch.float64) - size // 2
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    return torch.outer(g, g)


def _ssim_cs_64(x, y, data_range, size, sigma, k):
    """Per-(n, c) spatial means of the SSIM map and of its contrast-structure factor, float64."""
    ch = x.shape[1]
    w = _win2d(size, sigma).view(1, 1, size, size).repeat(ch, 1, 1, 1)
    blur = lambda t: F.conv2d(t, w, groups=ch)
    c1, c2 = (k[0] * data_range) ** 2, (k[1] * data_range) ** 2
    mx, my = blur(x), blur(y)
    vx, vy, cxy = blur(x * x) - mx * mx, blur(y * y) - my * my, blur(x * y) - mx * my
    cs = (2 * cxy + c2) / (vx + vy + c2)
    lum = (2 * mx * my + c1) / (mx * mx + my * my + c1)
    return (lum * cs).mean((2, 3)), cs.mean((2, 3))


def _halve(t):
    """avg_pool2d(kernel 2, stride 2, padding = size % 2, pad counted) written as explicit block means."""
    ph, pw = t.shape[2] % 2, t.shape[3] % 2
    t = F.pad(t, (pw, pw, ph, ph))
    h2, w2 = t.shape[2] // 2, t.shape[3] // 2
    t = t[:, :, :h2 * 2, :w2 * 2]
    return t.reshape(t.shape[0], t.shape[1], h2, 2, w2, 2).mean((3, 5))


def msssim_ssim(X, Y, data_range=255, size_average=True, win_size=11, win_sigma=1.5, win=None, K=(0.01, 0.03), nonnegative_ssim=False):
    assert win is None and X.dim() == 4 and X.shape == Y.shape
    s, _ = _ssim_cs_64(X.double(), Y.double(), data_range, win_size, win_sigma, K)
    if nonnegative_ssim:
        s = torch.relu(s)
    return (s.mean() if size_average else s.mean(1)).to(X.dtype)


def msssim_ms_ssim(X, Y, data_range=255, size_average=True, win_size=11, win_sigma=1.5, win=None, weights=None, K=(0.01, 0.03)):
    assert win is None and X.dim() == 4 and X.shape == Y.shape
    assert min(X.shape[-2:]) > (win_size - 1) * 2 ** 4
    weights = [0.0448, 0.2856, 0.3001, 0.2363, 0.1333] if weights is None else weights
    x, y = X.double(), Y.double()
    val = None
    for lvl, wgt in enumerate(weights):
        s, cs = _ssim_cs_64(x, y, data_range, win_size, win_sigma, K)
        term = torch.relu(s if lvl == len(weights) - 1 else cs) ** wgt
        val = term if val is None else val * term
        if lvl < len(weights) - 1:
            x, y = _halve(x), _halve(y)
    return (val.mean() if size_average else val.mean(1)).to(X.dtype)


def install():
    """Put the stand-ins into sys.modules and the reference root on sys.path."""
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    dummy = type("_Unused", (), {})
    timm = _mod("timm")
    timm.models = _mod("timm.models")
    timm.models.vision_transformer = _mod("timm.models.vision_transformer", Block=Block, PatchEmbed=PatchEmbed)
    timm.loss = _mod("timm.loss", SoftTargetCrossEntropy=dummy)
    xf = _mod("xformers")
    xf.factory = _mod("xformers.factory", xFormer=dummy, xFormerConfig=dummy)
    _mod("pytorch_msssim", ssim=msssim_ssim, ms_ssim=msssim_ms_ssim)
    _mod("wandb", log=lambda *a, **k: None)
    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms", RandomResizedCrop=RandomResizedCrop, InterpolationMode=InterpolationMode, ToTensor=ToTensor,
                         Normalize=Normalize, RandomHorizontalFlip=RandomHorizontalFlip, RandomVerticalFlip=RandomVerticalFlip, Compose=Compose,
                         Resize=_eval_only("Resize"), CenterCrop=_eval_only("CenterCrop"))
    # util/datasets.py:10-21 imports the multi-band readers' packages at module level; nothing on the RGB path calls into them
    import logging as _logging
    rio = _mod("rasterio", logging=_logging)
    rio.transform = _mod("rasterio.transform", Affine=dummy)
    rio.crs = _mod("rasterio.crs", CRS=dummy)
    rio.features = _mod("rasterio.features", rasterize=dummy)
    fio = _mod("fiona")
    fio.errors = _mod("fiona.errors", FionaValueError=type("FionaValueError", (ValueError,), {}))
    fio.transform = _mod("fiona.transform", transform_geom=dummy)
    for missing in ("models_mae_cross", "models_mae_crossv2", "models_mae_shunted", "models_mae_shunted_cross"):
        _mod("models_mae." + missing)
    torch.cuda.synchronize = lambda *a, **k: None  # engine_pretrain.py:72 hard-requires a GPU otherwise
