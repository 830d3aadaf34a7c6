#!/usr/bin/env python3
"""DEV-ONLY: generate `tests/golden/*` by running the REFERENCE itself in the build container.

Usage (build container only; `/root/reference` does not exist on the GPU box):

    python oracle/gen_golden.py            # rewrites tests/golden/

The reference is imported through `oracle/ref_stubs.py` (inert stand-ins for missing packages and a
restatement of timm 0.4.12 Block/PatchEmbed + torchvision 0.15.1 RandomResizedCrop).  Fixtures are
data only: inputs, RNG draws (noise tensors, crop boxes) and the reference's outputs.  No reference
source text is stored.
"""
from __future__ import annotations

import contextlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()
_print = print
import builtins  # noqa: E402

builtins.print = lambda *a, **k: None  # the reference prints from constructors
import models_mae  # noqa: E402  (reference)
import engine_pretrain  # noqa: E402  (reference)
import util.contrast_loss as ref_contrast  # noqa: E402
import util.lr_sched as ref_lr_sched  # noqa: E402
import util.misc as ref_misc  # noqa: E402
import util.pos_embed as ref_pos  # noqa: E402
from models_mae.MLP import MLP as ref_MLP  # noqa: E402

builtins.print = _print
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)

MICRO = dict(dim_model=128, encoder_num_layers=2, encoder_num_heads=2, decoder_embed_dim=64,
             decoder_num_layers=2, decoder_num_heads=2)
MICRO_HP = 128


def npy(t):
    return t.detach().cpu().numpy().copy() if isinstance(t, torch.Tensor) else np.array(t)


@contextlib.contextmanager
def quiet():
    builtins.print = lambda *a, **k: None
    try:
        yield
    finally:
        builtins.print = _print


@contextlib.contextmanager
def record_rand(store, inject=None):
    """Record (or replace) every torch.rand draw — the masking noise (MAE_ViT_Shared.py:66)."""
    orig = torch.rand
    queue = list(inject) if inject is not None else None

    def rand(*a, **k):
        r = orig(*a, **k)
        if queue:
            r = queue.pop(0).to(r.device).reshape(r.shape)
        store.append(r.clone())
        return r

    torch.rand = rand
    try:
        yield
    finally:
        torch.rand = orig


# ------------------------------------------------------------------------------- G1 sincos
def g_sincos():
    d = {}
    for dim, grid in [(128, 4), (64, 4), (768, 4), (512, 4)]:
        d[f"full_{dim}_{grid}"] = ref_pos.get_2d_sincos_pos_embed(dim, grid, cls_token=True)
    rng = np.random.RandomState(0)
    for dim, grid in [(768, 14), (512, 14), (1024, 16), (1280, 16), (1024, 14)]:
        tab = ref_pos.get_2d_sincos_pos_embed(dim, grid, cls_token=True)
        idx = rng.randint(0, tab.size, size=4096)
        d[f"idx_{dim}_{grid}"] = idx
        d[f"val_{dim}_{grid}"] = tab.reshape(-1)[idx]
        d[f"sum_{dim}_{grid}"] = np.array([tab.sum(), np.abs(tab).sum(), (tab * np.arange(tab.size).reshape(tab.shape)).sum()])
    np.savez_compressed(os.path.join(OUT, "sincos.npz"), **d)


# ------------------------------------------------------------------------------ G2 masking
def g_masking(model):
    d = {}
    g = torch.Generator().manual_seed(1234)
    for L in (16, 196, 256):
        N, D = 8, 4
        noise = torch.rand(N, L, generator=g)
        tie_rows = np.zeros(N, dtype=np.int64)
        noise[5, 3] = noise[5, 1]                       # one exact tie
        noise[6, : L // 2] = noise[6, L // 2: 2 * (L // 2)]  # many ties
        noise[7, :] = 0.5                                # all equal
        tie_rows[5:] = 1
        x = torch.randn(N, L, D, generator=g)
        for mr in (0.75, 0.5):
            rec = []
            with record_rand(rec, inject=[noise]):
                xm, mask, ids_restore = model.random_masking(x, mr)
            assert torch.equal(rec[0], noise)
            tag = f"L{L}_mr{int(mr * 100)}"
            d[f"x_masked_{tag}"] = npy(xm)
            d[f"mask_{tag}"] = npy(mask)
            d[f"ids_restore_{tag}"] = npy(ids_restore)
        d[f"noise_L{L}"] = npy(noise)
        d[f"x_L{L}"] = npy(x)
        d[f"tie_rows_L{L}"] = tie_rows
    np.savez_compressed(os.path.join(OUT, "masking.npz"), **d)


# --------------------------------------------------------------- G3/G4 patchify and losses
def g_patch_loss():
    d = {}
    g = torch.Generator().manual_seed(7)
    with quiet():
        shared = {k: models_mae.MAE_ViT_Baseline.__mro__[1](loss=k) for k in ("mse", "l2", "mae", "l1", "bce")}
        shared_np = models_mae.MAE_ViT_Baseline.__mro__[1](loss="mse", norm_pix_loss=True)
    m = shared["mse"]
    for p, c, s in [(16, 3, 32), (16, 4, 32), (14, 3, 28), (4, 3, 16)]:
        imgs = torch.randn(2, c, s, s, generator=g)
        pt = m.patchify(imgs, p, c)
        d[f"imgs_p{p}c{c}"] = npy(imgs)
        d[f"patches_p{p}c{c}"] = npy(pt)
        assert torch.equal(m.unpatchify(pt, p, c), imgs)
    imgs = torch.randn(3, 3, 32, 32, generator=g)
    pred = torch.randn(3, 4, 16 * 16 * 3, generator=g)
    mask = (torch.rand(3, 4, generator=g) > 0.4).float()
    d["loss_imgs"], d["loss_pred"], d["loss_mask"] = npy(imgs), npy(pred), npy(mask)
    for k, mod in shared.items():
        d[f"loss_{k}_masked"] = npy(mod.forward_loss(imgs, pred, mask, 16, 3))
        d[f"loss_{k}_nomask"] = npy(mod.forward_loss(imgs, pred, None, 16, 3))
        tgt = torch.randn(3, 4, 20, generator=g)
        prd = torch.randn(3, 4, 20, generator=g)
        d[f"raw_t_{k}"], d[f"raw_p_{k}"] = npy(tgt), npy(prd)
        d[f"loss_{k}_raw"] = npy(mod.forward_loss(tgt, prd))
    d["target_normpix"] = npy(shared_np.process_target(imgs, 16, 3))
    d["loss_mse_normpix"] = npy(shared_np.forward_loss(imgs, pred, mask, 16, 3))
    np.savez_compressed(os.path.join(OUT, "patch_loss.npz"), **d)


# ---------------------------------------------------------------- G4b ssim-family losses (§8 f-4)
def g_ssim_loss():
    """The reference's own `forward_loss_{ssim,ms_ssim,mse_ssim,mse_ms_ssim}` (scale_01, unpatchify, mask,
    1 - ssim, the 0.1 weight) around ref_stubs' float64 formulation of pytorch-msssim 0.2.1; value and d/dpred."""
    d = {}
    g = torch.Generator().manual_seed(11)
    kinds = ("ssim", "ms_ssim", "mse_ssim", "mse_ms_ssim")
    with quiet():
        mods = {k: models_mae.MAE_ViT_Baseline.__mro__[1](loss=k) for k in kinds}
        mod_np = models_mae.MAE_ViT_Baseline.__mro__[1](loss="mse_ssim", norm_pix_loss=True)
    # (tag, N, S, p): 64^2 for the single-scale kinds; 168^2 / p=8 for the five-scale ones (168 > 160; levels 168, 84, 42, 21, 11:
    # level 3 is odd, so the padded pooling is covered).  Gradients are kept for the masked calls only (fixture size).
    for tag, n, s, p, ks in [("s64", 2, 64, 16, ("ssim", "mse_ssim")), ("s168", 1, 168, 8, kinds)]:
        L = (s // p) ** 2
        imgs = torch.randn(n, 3, s, s, generator=g)
        # a smooth image-like prediction plus noise, so that the structure terms are well away from the relu clamps
        pred = (mods["ssim"].patchify(imgs, p, 3) * 0.7 + 0.5 * torch.randn(n, L, p * p * 3, generator=g)).requires_grad_(True)
        mask = (torch.rand(n, L, generator=g) > 0.3).float()
        d[f"{tag}_imgs"], d[f"{tag}_pred"], d[f"{tag}_mask"] = npy(imgs), npy(pred), npy(mask)
        d[f"{tag}_p"] = np.array(p)
        for k in ks:
            loss = mods[k].forward_loss(imgs, pred, mask, p, 3)
            d[f"{tag}_{k}_masked"] = npy(loss)
            if not (tag == "s168" and k in ("ssim", "mse_ssim")):
                (gr,) = torch.autograd.grad(loss, pred)
                d[f"{tag}_{k}_masked_grad"] = npy(gr).astype(np.float32)
            d[f"{tag}_{k}_nomask"] = npy(mods[k].forward_loss(imgs, pred, None, p, 3))
        if tag == "s64":
            loss = mod_np.forward_loss(imgs, pred, mask, p, 3)
            (gr,) = torch.autograd.grad(loss, pred)
            d[f"{tag}_mse_ssim_normpix"], d[f"{tag}_mse_ssim_normpix_grad"] = npy(loss), npy(gr)
    np.savez_compressed(os.path.join(OUT, "ssim_loss.npz"), **d)


# ------------------------------------------------------------------------------- G5 ntxent
def g_ntxent():
    d = {}
    g = torch.Generator().manual_seed(11)
    for bs, D in [(1, 8), (2, 8), (4, 32), (128, 32), (16, 768)]:
        f1 = torch.randn(bs, D, generator=g) * 2.7
        f2 = f1 * 0.5 + torch.randn(bs, D, generator=g)
        f1.requires_grad_(True)
        f2.requires_grad_(True)
        loss = ref_contrast.NTXentLoss(bs, 0.5, cos_sim=True)(f1, f2)
        loss.backward()
        d[f"f1_{bs}"], d[f"f2_{bs}"], d[f"loss_{bs}"] = npy(f1), npy(f2), npy(loss)
        d[f"g1_{bs}"], d[f"g2_{bs}"] = npy(f1.grad), npy(f2.grad)
    np.savez_compressed(os.path.join(OUT, "ntxent.npz"), **d)


# ---------------------------------------------------------------------------- G6 predictor
def g_predictor():
    d = {}
    torch.manual_seed(3)
    mlp = ref_MLP(16, 6, 32)
    with torch.no_grad():
        mlp[1].weight.uniform_(0.5, 1.5)
        mlp[1].bias.uniform_(-0.5, 0.5)
    for k, v in mlp.state_dict().items():
        d["sd_" + k] = npy(v)
    mlp.train()
    x1 = torch.randn(5, 6, 16, requires_grad=True)
    x2 = torch.randn(5, 6, 16)
    y1 = mlp(x1)
    (y1 ** 2).mean().backward()
    d["x1"], d["y1"], d["gx1"] = npy(x1), npy(y1), npy(x1.grad)
    for name, p in mlp.named_parameters():
        d["g_" + name] = npy(p.grad)
    d["rm1"], d["rv1"] = npy(mlp[1].running_mean), npy(mlp[1].running_var)
    y2 = mlp(x2)
    d["x2"], d["y2"] = npy(x2), npy(y2)
    d["rm2"], d["rv2"], d["nbt2"] = npy(mlp[1].running_mean), npy(mlp[1].running_var), npy(mlp[1].num_batches_tracked)
    np.savez_compressed(os.path.join(OUT, "predictor.npz"), **d)


# -------------------------------------------------------------------------------- G7 block
def seeded_state(shapes, seed, scale=0.05):
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(*shapes[k], generator=g) * scale + (1.0 if k.endswith("norm1.weight") or k.endswith("norm2.weight") else 0.0)
            for k in sorted(shapes)}


def g_block():
    from functools import partial
    d = {}
    meta = {}
    for tag, (T, D, H, B, store_w) in {"s128": (5, 128, 2, 3, True), "s64": (17, 64, 2, 2, True),
                                       "b768": (5, 768, 12, 2, False), "b512": (17, 512, 16, 2, False),
                                       "b1280": (7, 1280, 16, 1, False)}.items():
        blk = ref_stubs.Block(D, H, 4, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
        shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
        seed = 100 + D
        sd = seeded_state(shapes, seed)
        blk.load_state_dict(sd)
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(B, T, D, generator=g, requires_grad=True)
        go = torch.randn(B, T, D, generator=g)
        y = blk(x)
        (y * go).sum().backward()
        meta[tag] = dict(T=T, D=D, H=H, B=B, seed=seed, store_w=store_w)
        d[f"{tag}_x"], d[f"{tag}_go"], d[f"{tag}_y"], d[f"{tag}_gx"] = npy(x), npy(go), npy(y), npy(x.grad)
        for name, p in blk.named_parameters():
            if store_w:
                d[f"{tag}_w_{name}"] = npy(p)
                d[f"{tag}_g_{name}"] = npy(p.grad)
            else:
                gg = p.grad.double()
                d[f"{tag}_gn_{name}"] = np.array([gg.pow(2).sum().item(), gg.sum().item()])
                d[f"{tag}_gs_{name}"] = npy(p.grad.reshape(-1)[:64])
    np.savez_compressed(os.path.join(OUT, "block.npz"), **d)
    json.dump(meta, open(os.path.join(OUT, "block_meta.json"), "w"), indent=1)


# --------------------------------------------------------------------------------- G8 crop
def g_crop():
    d = {}
    g = torch.Generator().manual_seed(5)
    imgs = torch.randn(2, 2, 64, 64, generator=g)
    d["imgs64"] = npy(imgs)
    boxes = [(7, 2, 45, 48), (0, 0, 64, 64), (10, 20, 30, 40), (3, 5, 33, 17), (31, 0, 33, 64)]
    d["boxes64"] = np.array(boxes)
    for n, (i, j, h, w) in enumerate(boxes):
        d[f"out64_{n}"] = npy(torch.nn.functional.interpolate(imgs[..., i:i + h, j:j + w], size=(64, 64), mode="bilinear",
                                                               align_corners=False, antialias=True))
    big = torch.randn(1, 3, 224, 224, generator=g)
    d["seed224"] = np.array([77])
    big = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(77))
    boxes = [(12, 40, 131, 150), (100, 3, 97, 120), (0, 0, 193, 224)]
    d["boxes224"] = np.array(boxes)
    rng = np.random.RandomState(1)
    idx = rng.randint(0, big.numel(), size=8192)
    d["idx224"] = idx
    for n, (i, j, h, w) in enumerate(boxes):
        o = torch.nn.functional.interpolate(big[..., i:i + h, j:j + w], size=(224, 224), mode="bilinear",
                                            align_corners=False, antialias=True)
        d[f"val224_{n}"] = npy(o.reshape(-1)[idx])
        d[f"sum224_{n}"] = np.array([o.double().sum().item(), o.double().abs().sum().item()])
    # RandomResizedCrop box sequence from the CPU RNG (torchvision get_params restated in ref_stubs)
    seq = []
    torch.manual_seed(0)
    for _ in range(16):
        seq.append(ref_stubs.rrc_get_params(224, 224, (0.25, 0.75)))
    d["rrc_seed0_224"] = np.array(seq)
    seq = []
    torch.manual_seed(123)
    for _ in range(16):
        seq.append(ref_stubs.rrc_get_params(64, 64, (0.25, 0.75)))
    d["rrc_seed123_64"] = np.array(seq)
    np.savez_compressed(os.path.join(OUT, "crop.npz"), **d)


# --------------------------------------------------------------------- G9 full micro model
SEL = ["cls_token", "mask_token", "patch_embed.proj.bias", "patch_embed.proj.weight", "encoder.0.attn.qkv.weight",
       "encoder.0.attn.qkv.bias", "encoder.0.norm1.weight", "encoder.1.mlp.fc2.bias", "encoder.1.mlp.fc1.weight",
       "decoder_embed.weight", "decoder.1.mlp.fc1.weight", "decoder.0.attn.proj.weight", "decoder.1.norm2.bias",
       "decoder_pred.bias", "decoder_pred.weight", "decoder_norm.weight", "predictor.0.weight", "predictor.1.weight",
       "predictor.1.bias", "predictor.3.bias", "predictor.3.weight"]


def run_variant(cls_name, sd_src, imgs, extra_kw, noise=None, box=None, steps=1, loss="mse", norm_pix=False,
                reduction="sum"):
    cls = getattr(models_mae, cls_name) if hasattr(models_mae, cls_name) else models_mae.MAE_ViT_Baseline
    kw = dict(MICRO, input_size=imgs.shape[-1], patch_size="16", input_channels=imgs.shape[1], loss=loss,
              norm_pix_loss=norm_pix, **extra_kw)
    if cls_name != "MAE_ViT_Baseline":
        kw["ms_decoder_loss_reduction"] = reduction
    torch.manual_seed(0)
    with quiet():
        model = cls(**kw)
    if sd_src is not None:
        model.load_state_dict({k: v for k, v in sd_src.items() if k in model.state_dict()}, strict=True)
    model.train()
    rec = dict(recon=[], ce=[], cd=[], e=[])
    orig_fl = model.forward_loss

    def fl(*a, **k):
        r = orig_fl(*a, **k)
        rec["recon"].append(r.detach().clone())
        return r

    model.forward_loss = fl
    for attr, key in (("_MAE_ViT_MsLdCeCd__forward_loss_cd", "cd"), ("_MAE_ViT_MsLdCd__forward_loss_cd", "cd"),
                      ("_MAE_ViT_MsLdLeCd__forward_loss_cd", "cd"), ("_MAE_ViT_MsLdLe__forward_loss_e", "e"),
                      ("_MAE_ViT_MsLdLeCd__forward_loss_e", "e")):
        if hasattr(model, attr):
            def mk(orig, key):
                def f(*a, **k):
                    r = orig(*a, **k)
                    rec[key].append(r.detach().clone())
                    return r
                return f
            setattr(model, attr, mk(getattr(model, attr), key))
    orig_nt = ref_contrast.NTXentLoss.forward

    def nt(self, zi, zj):
        r = orig_nt(self, zi, zj)
        rec["ce"].append(r.detach().clone())
        return r

    ref_contrast.NTXentLoss.forward = nt
    params = [(n, p) for n, p in model.named_parameters()]
    decay = [p for n, p in params if p.requires_grad and not (p.ndim == 1 or n.endswith(".bias"))]
    no_decay = [p for n, p in params if p.requires_grad and (p.ndim == 1 or n.endswith(".bias"))]
    opt = torch.optim.AdamW([dict(params=no_decay, weight_decay=0.0), dict(params=decay, weight_decay=0.05)],
                            lr=1e-3, betas=(0.9, 0.95))
    res = dict(steps=[])
    noises, boxes = [], []
    try:
        for s in range(steps):
            opt.zero_grad()
            draws = []
            if box is not None:
                bx = box[s]
                orig_gp = ref_stubs.rrc_get_params
                ref_stubs.rrc_get_params = lambda *a, **k: bx
            with record_rand(draws, inject=None if noise is None else noise[s]):
                out = model(imgs, mask_ratio=0.75, return_embeds=True)
            if box is not None:
                ref_stubs.rrc_get_params = orig_gp
            loss_t = out[0]
            loss_t.backward()
            st = dict(loss=loss_t.detach().clone(), pred=out[1].detach().clone(), mask=out[2].detach().clone(),
                      recon=[r for r in rec["recon"]], ce=list(rec["ce"]), cd=list(rec["cd"]), e=list(rec["e"]),
                      noise=draws, box=ref_stubs.RandomResizedCrop.last_box,
                      grads={n: p.grad.detach().clone() for n, p in params if p.grad is not None},
                      nograd=[n for n, p in params if p.requires_grad and p.grad is None])
            if cls_name == "MAE_ViT_Baseline":
                st["enc"], st["dec"] = out[3].detach().clone(), out[4].detach().clone()
            else:
                st["enc"] = [t.detach().clone() for t in out[3]]
                st["dec"] = [t.detach().clone() for t in out[4]]
            for k in rec:
                rec[k].clear()
            opt.step()
            st["params_after"] = {n: p.detach().clone() for n, p in params}
            st["buffers_after"] = {n: b.detach().clone() for n, b in model.named_buffers()}
            res["steps"].append(st)
    finally:
        ref_contrast.NTXentLoss.forward = orig_nt
    res["model"] = model
    return res


def pack_step(d, tag, st, full_sd_keys, level="lite"):
    d[f"{tag}_loss"] = npy(st["loss"])
    d[f"{tag}_mask"] = npy(st["mask"])
    d[f"{tag}_pred_head"] = npy(st["pred"][:, :3, :32])
    d[f"{tag}_pred_sum"] = np.array([st["pred"].double().sum().item(), st["pred"].double().abs().sum().item()])
    if level == "full":
        d[f"{tag}_pred"] = npy(st["pred"])
    d[f"{tag}_recon"] = np.array([r.item() for r in st["recon"]])
    for k in ("ce", "cd", "e"):
        if st[k]:
            d[f"{tag}_loss_{k}"] = np.array([r.item() for r in st[k]])
    for n, t in enumerate(st["noise"]):
        d[f"{tag}_noise{n}"] = npy(t)
    if st["box"] is not None:
        d[f"{tag}_box"] = np.array(st["box"])
    if level == "full":
        if isinstance(st["enc"], list):
            d[f"{tag}_enc_orig"], d[f"{tag}_enc_crop"] = npy(st["enc"][0]), npy(st["enc"][1])
            d[f"{tag}_dec_orig"], d[f"{tag}_dec_crop"] = npy(st["dec"][0]), npy(st["dec"][1])
        else:
            d[f"{tag}_enc_orig"], d[f"{tag}_dec_orig"] = npy(st["enc"]), npy(st["dec"])
    names = sorted(st["grads"])
    d[f"{tag}_gradnames"] = np.array(names)
    d[f"{tag}_gradsq"] = np.array([st["grads"][n].double().pow(2).sum().item() for n in names])
    d[f"{tag}_gradsum"] = np.array([st["grads"][n].double().sum().item() for n in names])
    d[f"{tag}_nograd"] = np.array(st["nograd"])
    for n in SEL:
        if n in st["grads"] and (level == "full" or st["grads"][n].numel() <= 2048):
            d[f"{tag}_g_{n}"] = npy(st["grads"][n])
            if st["grads"][n].numel() <= 16384:
                d[f"{tag}_p_{n}"] = npy(st["params_after"][n])
    d[f"{tag}_paramsum_after"] = np.array([st["params_after"][n].double().sum().item() for n in names])
    for n, b in st["buffers_after"].items():
        d[f"{tag}_buf_{n}"] = npy(b)


def g_model_micro():
    g = torch.Generator().manual_seed(42)
    imgs = torch.randn(4, 3, 64, 64, generator=g)
    d = {"imgs": npy(imgs)}
    # superset state_dict: CeCd
    res = run_variant("MAE_ViT_MsLdCeCd", None, imgs, dict(predictor_hidden_size=MICRO_HP), steps=2)
    # state BEFORE training = fresh seeded init; rebuild to store it
    torch.manual_seed(0)
    with quiet():
        fresh = models_mae.MAE_ViT_MsLdCeCd(**MICRO, input_size=64, patch_size="16", predictor_hidden_size=MICRO_HP)
    sd0 = {k: v.clone() for k, v in fresh.state_dict().items()}
    for k, v in sd0.items():
        d["sd_" + k] = npy(v)
    for s, st in enumerate(res["steps"]):
        pack_step(d, f"cecd_s{s}", st, sd0, level="full" if s == 0 else "lite")
    noise = [st["noise"] for st in res["steps"]]
    box = [st["box"] for st in res["steps"]]
    for cls_name, extra in [("MAE_ViT_MsLd", {}), ("MAE_ViT_MsLdLe", {}), ("MAE_ViT_MsLdCd", dict(predictor_hidden_size=MICRO_HP)),
                            ("MAE_ViT_MsLdLeCd", dict(predictor_hidden_size=MICRO_HP))]:
        r = run_variant(cls_name, sd0, imgs, extra, noise=noise[:1], box=box[:1], steps=1)
        pack_step(d, cls_name.replace("MAE_ViT_", "").lower() + "_s0", r["steps"][0], sd0)
    r = run_variant("MAE_ViT_Baseline", sd0, imgs, {}, noise=[noise[0][:1]], steps=1)
    pack_step(d, "baseline_s0", r["steps"][0], sd0)
    # loss / option coverage on the headline model (same weights, same draws)
    for loss in ("l2", "mae", "l1", "bce"):
        r = run_variant("MAE_ViT_MsLdCeCd", sd0, imgs, dict(predictor_hidden_size=MICRO_HP), noise=noise[:1], box=box[:1], loss=loss)
        st = r["steps"][0]
        d[f"cecd_{loss}_loss"] = npy(st["loss"])
        d[f"cecd_{loss}_recon"] = np.array([x.item() for x in st["recon"]])
        d[f"cecd_{loss}_loss_cd"] = np.array([x.item() for x in st["cd"]])
        names = sorted(st["grads"])
        d[f"cecd_{loss}_gradsq"] = np.array([st["grads"][n].double().pow(2).sum().item() for n in names])
    r = run_variant("MAE_ViT_MsLdCeCd", sd0, imgs, dict(predictor_hidden_size=MICRO_HP), noise=noise[:1], box=box[:1],
                    norm_pix=True, reduction="mean")
    st = r["steps"][0]
    d["cecd_normpix_mean_loss"] = npy(st["loss"])
    d["cecd_normpix_mean_recon"] = np.array([x.item() for x in st["recon"]])
    names = sorted(st["grads"])
    d["cecd_normpix_mean_gradsq"] = np.array([st["grads"][n].double().pow(2).sum().item() for n in names])
    # 4-band 128^2 variant (config-4 shape idea at micro width): p=16, C=4, L=64
    g2 = torch.Generator().manual_seed(43)
    imgs4 = torch.randn(2, 4, 128, 128, generator=g2)
    r = run_variant("MAE_ViT_MsLdCeCd", None, imgs4, dict(predictor_hidden_size=MICRO_HP), steps=1)
    torch.manual_seed(0)
    with quiet():
        fresh4 = models_mae.MAE_ViT_MsLdCeCd(**MICRO, input_size=128, input_channels=4, patch_size="16", predictor_hidden_size=MICRO_HP)
    d4 = {"imgs": npy(imgs4)}
    for k, v in fresh4.state_dict().items():
        if k not in sd0 or tuple(v.shape) != tuple(sd0[k].shape) or not torch.equal(v, sd0[k]):
            d4["sd_" + k] = npy(v)
    pack_step(d4, "cecd4_s0", r["steps"][0], None)
    for k in list(d4):
        if k.startswith("cecd4_s0_p_"):
            del d4[k]
    np.savez_compressed(os.path.join(OUT, "model_micro.npz"), **d)
    np.savez_compressed(os.path.join(OUT, "model_micro_c4.npz"), **d4)


# ------------------------------------------------------------- G9b/G10/G11 ViT-B anchors, engine
def checksum(t):
    t = t.double()
    flat = t.reshape(-1)
    w = torch.arange(flat.numel(), dtype=torch.float64) % 97 + 1
    return [float(flat.sum()), float(flat.abs().sum()), float((flat * w).sum()), [float(x) for x in flat[:4]]]


def g_vitb():
    out = {}
    factories = ["mae_vit_base", "mae_vit_base_MsLd", "mae_vit_base_MsLdLe", "mae_vit_base_MsLdCd", "mae_vit_base_MsLdCe",
                 "mae_vit_base_MsLdLeCd", "mae_vit_base_MsLdCeCd", "mae_vit_large", "mae_vit_huge"]
    manifest = {}
    for f in factories:
        torch.manual_seed(0)
        with quiet():
            m = getattr(models_mae, f)(input_size=64, patch_size="16", loss="mse", device="cpu")
        sd = m.state_dict()
        manifest[f] = dict(keys=list(sd.keys()), shapes=[list(v.shape) for v in sd.values()],
                           dtypes=[str(v.dtype) for v in sd.values()],
                           trainable=int(sum(p.numel() for p in m.parameters() if p.requires_grad)),
                           param_order=[n for n, _ in m.named_parameters()],
                           requires_grad=[bool(p.requires_grad) for _, p in m.named_parameters()])
        if f in ("mae_vit_base", "mae_vit_base_MsLdCeCd"):
            manifest[f]["checksums"] = {k: checksum(v) for k, v in sd.items() if v.is_floating_point()}
    torch.manual_seed(0)
    with quiet():
        m = models_mae.mae_vit_base_MsLdCeCd(input_size=224, patch_size="16", loss="mse", device="cpu")
    sd = m.state_dict()
    manifest["mae_vit_base_MsLdCeCd@224"] = dict(keys=list(sd.keys()), shapes=[list(v.shape) for v in sd.values()],
                                                 trainable=int(sum(p.numel() for p in m.parameters() if p.requires_grad)))
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"))

    # anchor A: Baseline ViT-B @64^2, N=2 (BASELINE.json configs[0]) through the reference engine
    d = {}
    torch.manual_seed(0)
    with quiet():
        m = models_mae.mae_vit_base(input_size=64, patch_size="16", loss="mse", device="cpu")
    x = torch.randn(2, 3, 64, 64)
    d["cfg1_imgs"] = npy(x)
    draws = []
    args = types.SimpleNamespace(accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=0, epochs=1, mask_ratio=0.75,
                                 local_rank=0, wandb_project=None)
    decay = [p for n, p in m.named_parameters() if p.requires_grad and not (p.ndim == 1 or n.endswith(".bias"))]
    no_decay = [p for n, p in m.named_parameters() if p.requires_grad and (p.ndim == 1 or n.endswith(".bias"))]
    opt = torch.optim.AdamW([dict(params=no_decay, weight_decay=0.0), dict(params=decay, weight_decay=0.05)], lr=1e-3, betas=(0.9, 0.95))
    with quiet(), record_rand(draws):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            stats = engine_pretrain.train_one_epoch(m, [(x, None)], opt, torch.device("cpu"), 0,
                                                    ref_misc.NativeScalerWithGradNormCount(), log_writer=None, args=args)
    d["cfg1_noise"] = npy(draws[0])
    out["cfg1_stats"] = {k: float(v) for k, v in stats.items()}
    sd_after = m.state_dict()
    out["cfg1_after_checksums"] = {k: checksum(sd_after[k]) for k in ("cls_token", "patch_embed.proj.weight", "encoder.0.attn.qkv.weight",
                                                                      "encoder.11.mlp.fc2.bias", "decoder.7.mlp.fc2.weight", "decoder_pred.bias",
                                                                      "mask_token", "decoder_norm.weight")}
    # anchor B: CeCd ViT-B @64^2, N=4
    torch.manual_seed(0)
    with quiet():
        m = models_mae.mae_vit_base_MsLdCeCd(input_size=64, patch_size="16", loss="mse", device="cpu")
    x = torch.randn(4, 3, 64, 64)
    d["cecd64_imgs"] = npy(x)
    res_rec = dict(recon=[], ce=[])
    ofl = m.forward_loss
    m.forward_loss = lambda *a, **k: (lambda r: (res_rec["recon"].append(float(r)), r)[1])(ofl(*a, **k))
    orig_nt = ref_contrast.NTXentLoss.forward
    ref_contrast.NTXentLoss.forward = lambda self, a, b: (lambda r: (res_rec["ce"].append(float(r)), r)[1])(orig_nt(self, a, b))
    draws = []
    with record_rand(draws):
        o = m(x, return_embeds=True)
    ref_contrast.NTXentLoss.forward = orig_nt
    o[0].backward()
    d["cecd64_noise0"], d["cecd64_noise1"] = npy(draws[0]), npy(draws[1])
    d["cecd64_pred_head"] = npy(o[1][:, :2, :64])
    d["cecd64_mask"] = npy(o[2])
    out["cecd64"] = dict(loss=float(o[0]), recon=res_rec["recon"], ce=res_rec["ce"][0], box=list(ref_stubs.RandomResizedCrop.last_box),
                         cd=float(o[0]) - sum(res_rec["recon"]) - res_rec["ce"][0],
                         gradsq={n: float(p.grad.double().pow(2).sum()) for n, p in m.named_parameters() if p.grad is not None},
                         nograd=[n for n, p in m.named_parameters() if p.requires_grad and p.grad is None])
    # LR schedule table
    tab = []
    for (lr, min_lr, wu, ep) in [(1e-3, 0.0, 40, 400), (1.5e-4, 1e-6, 5, 100), (1e-3, 0.0, 0, 1)]:
        a = types.SimpleNamespace(lr=lr, min_lr=min_lr, warmup_epochs=wu, epochs=ep)
        for e in [0.0, 0.5, 1.0, wu * 0.5, float(wu), wu + 0.25, ep * 0.5, ep - 0.5]:
            grp = [dict(lr=0.0), dict(lr=0.0, lr_scale=0.5)]
            o2 = types.SimpleNamespace(param_groups=grp)
            r = ref_lr_sched.adjust_learning_rate(o2, e, a)
            tab.append([lr, min_lr, wu, ep, e, r, grp[0]["lr"], grp[1]["lr"]])
    out["lr_table"] = tab
    json.dump(out, open(os.path.join(OUT, "vitb_anchor.json"), "w"))
    np.savez_compressed(os.path.join(OUT, "vitb_anchor.npz"), **d)



# ------------------------------------------------------------------------------- G10 full-size geometries of BASELINE.json configs[1..4]
FULLSIZE = {  # tag -> (geometry kwargs, input size, patch, channels, N)
    "vitb16_224": (dict(dim_model=768, encoder_num_layers=12, encoder_num_heads=12, decoder_embed_dim=512, decoder_num_layers=8, decoder_num_heads=16), 224, 16, 3, 4),
    "vitl16_224": (dict(dim_model=1024, encoder_num_layers=24, encoder_num_heads=16, decoder_embed_dim=512, decoder_num_layers=8, decoder_num_heads=16), 224, 16, 3, 2),
    "vitl16_256c4": (dict(dim_model=1024, encoder_num_layers=24, encoder_num_heads=16, decoder_embed_dim=512, decoder_num_layers=8, decoder_num_heads=16), 256, 16, 4, 2),
    "vith14_224": (dict(dim_model=1280, encoder_num_layers=32, encoder_num_heads=16, decoder_embed_dim=512, decoder_num_layers=8, decoder_num_heads=16), 224, 14, 3, 2),
}
FULLSIZE_WEIGHTS = ("patch_embed.proj.weight", "encoder.0.attn.qkv.weight", "decoder.7.mlp.fc2.weight", "predictor.0.weight", "mask_token")


def fullsize_inputs(tag):
    """Seeded inputs of the full-size fixtures (regenerated, not stored: 2.4 MB of incompressible noise per config).  The fixture
    carries a checksum so that a torch whose CPU generator differs is reported as such."""
    geom, S, p, C, N = FULLSIZE[tag]
    g = torch.Generator().manual_seed(1000 + sorted(FULLSIZE).index(tag))
    L = (S // p) ** 2
    return torch.randn(N, C, S, S, generator=g), [torch.rand(N, L, generator=g), torch.rand(N, L, generator=g)]


def g_fullsize(tags=None):
    """MAE_ViT_MsLdCeCd at the geometries BASELINE.json's configs[1..4] are quoted on (ViT-B/16 224^2; ViT-L/16 224^2; ViT-L/16 256^2
    4-band; ViT-H/14 224^2), small batches, seeded default init, one forward + backward of the REFERENCE: every loss term, the masks,
    and sum / sum of squares of every parameter gradient."""
    d, meta = {}, {}
    for tag in (tags or FULLSIZE):
        geom, S, p, C, N = FULLSIZE[tag]
        torch.manual_seed(0)
        with quiet():
            m = models_mae.MAE_ViT_MsLdCeCd(**geom, input_size=S, patch_size=str(p), input_channels=C, loss="mse", device="cpu")
        m.train()
        imgs, noise = fullsize_inputs(tag)
        rec = dict(recon=[], ce=[], cd=[])
        ofl = m.forward_loss
        m.forward_loss = lambda *a, _o=ofl, **k: (lambda r: (rec["recon"].append(float(r)), r)[1])(_o(*a, **k))
        ocd = getattr(m, "_MAE_ViT_MsLdCeCd__forward_loss_cd")
        setattr(m, "_MAE_ViT_MsLdCeCd__forward_loss_cd", lambda *a, _o=ocd, **k: (lambda r: (rec["cd"].append(float(r)), r)[1])(_o(*a, **k)))
        orig_nt = ref_contrast.NTXentLoss.forward
        ref_contrast.NTXentLoss.forward = lambda self, a, b: (lambda r: (rec["ce"].append(float(r)), r)[1])(orig_nt(self, a, b))
        draws = []
        try:
            torch.manual_seed(4242)   # the crop box comes from the global CPU generator (MAE_ViT_MsLd.py:52)
            with record_rand(draws, inject=noise):
                out = m(imgs, mask_ratio=0.75, return_embeds=True)
        finally:
            ref_contrast.NTXentLoss.forward = orig_nt
        out[0].backward()
        assert len(draws) == 2 and torch.equal(draws[0], noise[0]) and torch.equal(draws[1], noise[1])
        names = sorted(n for n, q in m.named_parameters() if q.grad is not None)
        grads = dict(m.named_parameters())
        d[f"{tag}_mask"] = npy(out[2]).astype(np.uint8)
        d[f"{tag}_noise0"], d[f"{tag}_noise1"] = npy(noise[0]), npy(noise[1])
        d[f"{tag}_gradnames"] = np.array(names)
        d[f"{tag}_gradsq"] = np.array([grads[n].grad.double().pow(2).sum().item() for n in names])
        d[f"{tag}_gradsum"] = np.array([grads[n].grad.double().sum().item() for n in names])
        d[f"{tag}_pred_head"] = npy(out[1][:, :2, :48])
        d[f"{tag}_g_mask_token"] = npy(m.mask_token.grad)
        d[f"{tag}_g_decoder_pred_bias"] = npy(m.decoder_pred.bias.grad)
        d[f"{tag}_g_cls_token"] = npy(m.cls_token.grad)
        meta[tag] = dict(geom=geom, input_size=S, patch=p, channels=C, N=N, loss=float(out[0]), recon=rec["recon"], ce=rec["ce"][0], cd=rec["cd"][0],
                         box=[int(v) for v in ref_stubs.RandomResizedCrop.last_box],
                         pred_sum=[out[1].double().sum().item(), out[1].double().abs().sum().item()],
                         enc_sum=[t.double().abs().sum().item() for t in out[3]], dec_sum=[t.double().abs().sum().item() for t in out[4]],
                         nograd=[n for n, q in m.named_parameters() if q.requires_grad and q.grad is None],
                         imgs_checksum=checksum(imgs)[:3], weights={k: checksum(v)[:3] for k, v in m.state_dict().items() if k in FULLSIZE_WEIGHTS},
                         bn_running_mean_sum=float(m.predictor[1].running_mean.double().sum()))
        _print(f"fullsize {tag}: loss {float(out[0]):.6f} recon {rec['recon']} ce {rec['ce'][0]:.6f} cd {rec['cd'][0]:.6f} box {meta[tag]['box']}")
        del m, out, grads
    if tags:   # partial regeneration: merge into the existing files
        old = dict(np.load(os.path.join(OUT, "fullsize.npz")))
        old.update(d)
        d = old
        om = json.load(open(os.path.join(OUT, "fullsize.json")))
        om.update(meta)
        meta = om
    np.savez_compressed(os.path.join(OUT, "fullsize.npz"), **d)
    json.dump(meta, open(os.path.join(OUT, "fullsize.json"), "w"), indent=1)

N128_SMALL = ("cls_token", "mask_token", "decoder_pred.bias", "predictor.1.weight", "predictor.1.bias", "decoder_norm.weight", "encoder.11.norm2.bias",
              "encoder.0.attn.qkv.bias", "decoder.0.mlp.fc2.bias", "patch_embed.proj.bias")


def n128_inputs():
    """Seeded inputs of the N = 128 fixture (same recipe as tests/test_model_gpu.py::test_full_size_vitb_224_n128_*: nothing stored)."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(128, 3, 224, 224, generator=g)
    return x, [torch.rand(128, 196, generator=g), torch.rand(128, 196, generator=g)]


def g_fullsize_n128():
    """BASELINE.json configs[1] at the batch the bench runs: the REFERENCE's MAE_ViT_MsLdCeCd ViT-B/16 224^2 at N = 128 (BatchNorm over the
    batch and the NT-Xent negatives couple the samples, so the N = 4 fixture cannot stand in for it), one forward + backward on CPU
    (about 2 min on 8 threads, ~35 GB): the four loss terms, the mask (bit-packed), sum / sum of squares of every parameter gradient
    and a few small gradients in full."""
    geom = FULLSIZE["vitb16_224"][0]
    torch.manual_seed(0)
    with quiet():
        m = models_mae.MAE_ViT_MsLdCeCd(**geom, input_size=224, patch_size="16", input_channels=3, loss="mse", device="cpu")
    m.train()
    imgs, noise = n128_inputs()
    rec = dict(recon=[], ce=[], cd=[])
    ofl = m.forward_loss
    m.forward_loss = lambda *a, _o=ofl, **k: (lambda r: (rec["recon"].append(float(r)), r)[1])(_o(*a, **k))
    ocd = getattr(m, "_MAE_ViT_MsLdCeCd__forward_loss_cd")
    setattr(m, "_MAE_ViT_MsLdCeCd__forward_loss_cd", lambda *a, _o=ocd, **k: (lambda r: (rec["cd"].append(float(r)), r)[1])(_o(*a, **k)))
    orig_nt = ref_contrast.NTXentLoss.forward
    ref_contrast.NTXentLoss.forward = lambda self, a, b: (lambda r: (rec["ce"].append(float(r)), r)[1])(orig_nt(self, a, b))
    draws = []
    try:
        torch.manual_seed(4242)
        with record_rand(draws, inject=noise):
            out = m(imgs, mask_ratio=0.75, return_embeds=True)
    finally:
        ref_contrast.NTXentLoss.forward = orig_nt
    out[0].backward()
    assert len(draws) == 2 and torch.equal(draws[0], noise[0]) and torch.equal(draws[1], noise[1])
    grads = {n: q.grad for n, q in m.named_parameters() if q.grad is not None}
    names = sorted(grads)
    d = dict(mask_bits=np.packbits(npy(out[2]).astype(np.uint8), axis=1), gradnames=np.array(names),
             gradsq=np.array([grads[n].double().pow(2).sum().item() for n in names]),
             gradsum=np.array([grads[n].double().sum().item() for n in names]), pred_head=npy(out[1][:2, :2, :48]))
    for n in N128_SMALL:
        d["g_" + n] = npy(grads[n])
    meta = dict(geom=geom, input_size=224, patch=16, channels=3, N=128, loss=float(out[0]), recon=rec["recon"], ce=rec["ce"][0], cd=rec["cd"][0],
                box=[int(v) for v in ref_stubs.RandomResizedCrop.last_box], imgs_checksum=checksum(imgs)[:3],
                noise_checksum=[checksum(noise[0])[:3], checksum(noise[1])[:3]],
                pred_sum=[out[1].double().sum().item(), out[1].double().abs().sum().item()],
                nograd=[n for n, q in m.named_parameters() if q.requires_grad and q.grad is None],
                weights={k: checksum(v)[:3] for k, v in m.state_dict().items() if k in FULLSIZE_WEIGHTS})
    _print(f"fullsize n128: loss {float(out[0]):.6f} recon {rec['recon']} ce {rec['ce'][0]:.6f} cd {rec['cd'][0]:.6f} box {meta['box']}")
    np.savez_compressed(os.path.join(OUT, "fullsize_n128.npz"), **d)
    json.dump(meta, open(os.path.join(OUT, "fullsize_n128.json"), "w"), indent=1)


# ------------------------------------------------------------------ G14 input transform
def g_input_transform():
    """The reference's own training transform (util/datasets.py:107-136 `BaseDataset.build_transform(True, ...)`) behind its own RGB dataset
    class (util/datasets.py:161-210: csv -> PIL -> transform) on seeded uint8 images of ragged sizes written as PNG files; the random
    decisions of every sample (flips, crop box) are recorded next to the output so that the GPU input step can be handed the same ones."""
    import tempfile
    from PIL import Image
    import util.datasets as ref_datasets
    S = 64
    mean, std = ref_datasets.Dataset_fmow_rgb.mean, ref_datasets.Dataset_fmow_rgb.std
    g = torch.Generator().manual_seed(77)
    sizes = [(97, 120), (150, 130), (64, 64), (120, 200), (33, 47), (80, 64), (200, 90)]
    d = dict(S=np.array(S), mean=np.array(mean), std=np.array(std), sizes=np.array(sizes))
    with tempfile.TemporaryDirectory() as tmp:
        rows = []
        for n, (h, w) in enumerate(sizes):
            # smooth + noise content: a bicubic anti-aliased resize of pure noise has little signal left to compare
            yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
            base = torch.stack([127 + 90 * torch.sin(yy / (5 + c) + n) * torch.cos(xx / (7 - c)) for c in range(3)], -1)
            img = (base + 30 * torch.randn(h, w, 3, generator=g)).clamp(0, 255).to(torch.uint8)
            Image.fromarray(img.numpy(), "RGB").save(os.path.join(tmp, f"im{n}.png"))
            d[f"img{n}"] = npy(img)
            rows.append((n % 5, f"im{n}.png"))
        with open(os.path.join(tmp, "train.csv"), "w") as f:
            f.write("category,image_path\n" + "".join(f"{c},{pth}\n" for c, pth in rows))
        tf = ref_datasets.BaseDataset.build_transform(True, S, mean, std)
        with quiet():
            ds = ref_datasets.Dataset_fmow_rgb(os.path.join(tmp, "train.csv"), tf)
        params, outs, seeds = [], [], []
        for rep in range(2):                      # two passes with different seeds: different flips / boxes per image
            for n, (h, w) in enumerate(sizes):
                seed = 1000 * rep + n
                torch.manual_seed(seed)
                x, label = ds[n]
                assert label == n % 5 and tuple(x.shape) == (3, S, S)
                i, j, bh, bw = ref_stubs.RandomResizedCrop.last_box
                params.append((h, w, i, j, bh, bw, int(ref_stubs.RandomHorizontalFlip.last), int(ref_stubs.RandomVerticalFlip.last)))
                outs.append(npy(x))
                seeds.append(seed)
    d["params"], d["out"], d["seeds"], d["index"] = np.array(params), np.stack(outs), np.array(seeds), np.array([k % len(sizes) for k in range(len(outs))])
    np.savez_compressed(os.path.join(OUT, "input_transform.npz"), **d)


# ------------------------------------------------------------------ G15 two explicit views (MAE_ViT_MsLd.py:79-146)
def g_paired():
    """MAE_ViT_MsLd_PAIRED at micro size on the seeded weights of model_micro.npz: two explicit views instead of the crop."""
    from models_mae.MAE_ViT_MsLd import MAE_ViT_MsLd_PAIRED
    g = torch.Generator().manual_seed(44)
    imgs1, imgs2 = torch.randn(4, 3, 64, 64, generator=g), torch.randn(4, 3, 64, 64, generator=g)
    torch.manual_seed(0)
    with quiet():
        fresh = models_mae.MAE_ViT_MsLdCeCd(**MICRO, input_size=64, patch_size="16", predictor_hidden_size=MICRO_HP)
    sd0 = {k: v.clone() for k, v in fresh.state_dict().items()}
    d = dict(imgs1=npy(imgs1), imgs2=npy(imgs2))
    for reduction, kw, tag in (("sum", {}, "sum"), ("mean", dict(mask_seed=11), "mean_seed11")):
        torch.manual_seed(0)
        with quiet():
            model = MAE_ViT_MsLd_PAIRED(**MICRO, input_size=64, patch_size="16", ms_decoder_loss_reduction=reduction)
        model.load_state_dict({k: v for k, v in sd0.items() if k in model.state_dict()}, strict=True)
        model.train()
        params = list(model.named_parameters())
        draws = []
        torch.manual_seed(5)
        with record_rand(draws):
            out = model(imgs1, imgs2, mask_ratio=0.75, return_embeds=True, **kw)
        out[0].backward()
        grads = {n: p.grad.detach().clone() for n, p in params if p.grad is not None}
        names = sorted(grads)
        d[f"{tag}_loss"], d[f"{tag}_pred"], d[f"{tag}_mask"] = npy(out[0]), npy(out[1]), npy(out[2])
        d[f"{tag}_enc_orig"], d[f"{tag}_enc_crop"] = npy(out[3][0]), npy(out[3][1])
        d[f"{tag}_dec_orig"], d[f"{tag}_dec_crop"] = npy(out[4][0]), npy(out[4][1])
        for n, t in enumerate(draws):
            d[f"{tag}_noise{n}"] = npy(t)
        d[f"{tag}_gradnames"] = np.array(names)
        d[f"{tag}_gradsq"] = np.array([grads[n].double().pow(2).sum().item() for n in names])
        d[f"{tag}_gradsum"] = np.array([grads[n].double().sum().item() for n in names])
        d[f"{tag}_nograd"] = np.array([n for n, p in params if p.requires_grad and p.grad is None])
        for n in SEL:
            if n in grads and grads[n].numel() <= 16384:
                d[f"{tag}_g_{n}"] = npy(grads[n])
    d["state_keys"] = np.array(sorted(model.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "paired.npz"), **d)


def main():
    torch.set_num_threads(8)
    with quiet():
        tiny = models_mae.MAE_ViT_Baseline(**MICRO, input_size=64, patch_size="16")
    g_sincos()
    g_masking(tiny)
    g_patch_loss()
    g_ssim_loss()
    g_ntxent()
    g_predictor()
    g_block()
    g_crop()
    g_model_micro()
    g_vitb()
    g_fullsize()
    g_fullsize_n128()
    g_input_transform()
    g_paired()
    for f in sorted(os.listdir(OUT)):
        _print(f"{f:28s} {os.path.getsize(os.path.join(OUT, f)) / 1024:9.1f} KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1:  # e.g. `python oracle/gen_golden.py g_ssim_loss`: regenerate one fixture
        torch.set_num_threads(8)
        for name in sys.argv[1:]:
            globals()[name]()
    else:
        main()
