"""Reconstruction helpers around the pre-training path (reference util/viz.py:27-206): checkpoint -> model, image file -> normalised
array, one masked forward of a single image -> (original, masked input, reconstruction, reconstruction of the masked patches,
reconstruction pasted into the visible patches).

The model forward is the HIP path (`model(x, mask_ratio=, mask_seed=)`); everything else here is host-side array handling.
The reference's matplotlib figures (`plot_image`, `plot_reconstruction`, `add_noise`, the dataset-wide sweeps, util/viz.py:123-138,
208-316) are plotting UI — SURVEY §2 row 19, out of scope — and are not rebuilt: `run_one_image` + `util.metrics.calc_metric` give the
arrays and scores a figure would show."""
import os
import re
from typing import Optional

import numpy as np
import torch

import models_mae
from util.gpu_input import resized_crop_box

# per-channel statistics the reference hard-codes for its plots (util/viz.py:23-24)
image_mean = np.array([0.40558367, 0.43378946, 0.43175863])
image_std = np.array([0.19208308, 0.19136319, 0.19783947])


def title_to_fname(title: str) -> str:
    """File-name form of a plot title (reference util/misc.py:428-436)."""
    s = re.sub(r"\s+", "_", re.sub(r"[^\w\s]", "_", title.replace("-", "")))
    while "__" in s:
        s = s.replace("__", "_")
    return s.strip("_")


def prepare_model(chkpt_dir, chkpt_basedir="../Model_Saving", chkpt_name=None, map_location="cpu"):
    """`<chkpt_basedir>/<chkpt_dir>/checkpoint-<epoch>.pth` (latest epoch when `chkpt_name` is None) -> the model its `args` describe,
    weights loaded (strict=False), moved to `model.device` when that is set (util/viz.py:27-89)."""
    folder = os.path.join(chkpt_basedir, chkpt_dir)
    if chkpt_name is None:
        names = [f for f in os.listdir(folder) if f.endswith(".pth")]
        if not names:
            raise IndexError(f"no checkpoint-*.pth under {folder}")
        chkpt_name = max(names, key=lambda f: int(f.split("-")[1].split(".")[0]))
    chkpt_name = str(chkpt_name)
    if not chkpt_name.endswith(".pth"):
        chkpt_name += ".pth"
    if not chkpt_name.startswith("checkpoint-"):
        chkpt_name = "checkpoint-" + chkpt_name
    path = os.path.join(folder, chkpt_name)
    print("Loading checkpoint: ", path)
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    args = dict(vars(ckpt["args"]))
    try:
        model = getattr(models_mae, args["model"])(**args)
    except AssertionError as e:  # an architecture option this build (like the reference) refuses
        print("Error: ", e)
        return None
    print(model.load_state_dict(ckpt["model"], strict=False))
    if model.device is not None:
        model = model.to(model.device)
    return model


def prepare_image(image_uri, img_size, random_crop=False, crop_seed=None, resample=None, **kwargs):
    """Image file -> [img_size, img_size, C] float64, scaled to [0, 1] and normalised with the plot statistics (util/viz.py:92-120).
    `random_crop`: torchvision's RandomResizedCrop(scale 0.25-1, bicubic) of the PIL image first, seeded by `crop_seed`."""
    from PIL import Image
    img = Image.open(image_uri)
    if random_crop:
        if crop_seed is not None:
            torch.manual_seed(crop_seed)
        i, j, h, w = resized_crop_box(img.height, img.width, scale=(0.25, 1.0))
        img = img.resize((img_size, img_size), Image.BICUBIC, box=(j, i, j + w, i + h))
    img = img.resize((img_size, img_size), resample=resample)
    return (np.array(img) / 255.0 - image_mean) / image_std


@torch.no_grad()
def run_one_image(img, model, mask_seed: Optional[int] = None, **kwargs):
    """One [H, W, C] normalised image through `model(x, mask_ratio=model.mask_ratio, mask_seed=)` -> five [1, H, W, C] CPU tensors in
    un-normalised image space: x, x with the masked patches blanked, y (reconstruction), y on the masked patches only, and x on the
    visible patches + y on the masked ones (util/viz.py:141-206)."""
    p, c = model.patch_size, model.input_channels
    x = torch.as_tensor(img).unsqueeze(0).permute(0, 3, 1, 2)
    mask_ratio = getattr(model, "mask_ratio", 0.75)
    xf = x.float()
    xf = xf.to(model.device if model.device is not None else next(model.parameters()).device)
    _, y, mask = model(xf, mask_ratio=mask_ratio, mask_seed=mask_seed)
    y = model.unpatchify(y, p=p, c=c).permute(0, 2, 3, 1).detach().cpu()
    mask = mask.detach().unsqueeze(-1).repeat(1, 1, model.patch_embed.patch_size[0] ** 2 * 3)   # 3 channels, as the reference (:186-188)
    mask = model.unpatchify(mask, p=p, c=c).permute(0, 2, 3, 1).cpu()                              # 1 = removed, 0 = kept
    x = x.permute(0, 2, 3, 1)
    std, mean = torch.as_tensor(image_std), torch.as_tensor(image_mean)
    x = x * std + mean
    y = y * std + mean
    xm = x * (1 - mask)
    ym = y * mask
    return x, xm, y, ym, xm + ym
