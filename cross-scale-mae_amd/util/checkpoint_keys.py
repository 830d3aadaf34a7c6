"""Checkpoint key mapping between the pre-training model and a plain timm-style ViT encoder.

`to_vit_keys` restates what the reference's fine-tuning / linear-probe entry points do to a pre-training checkpoint when
`--transform_checkpoint_keys` is given (main_finetune.py:553-586, main_linprobe.py same block), for the timm code path:

    encoder_pos_embed            -> pos_embed
    encoder_norm.{weight,bias}   -> norm.{weight,bias}
    encoder.<i>.<rest>           -> blocks.<i>.<rest>
    cls_token, patch_embed.proj.{weight,bias} kept as they are
    everything else (decoder, mask token, predictor, ...) dropped

`from_vit_keys` is the inverse (initialise the pre-training encoder from a ViT checkpoint); `load_pretrain_checkpoint` reads the
reference's checkpoint dict layout {"model", "optimizer", "epoch", "scaler", "args"} (util/misc.py:364-370)."""
from __future__ import annotations

from collections import OrderedDict

_KEPT = ("cls_token", "patch_embed.proj.weight", "patch_embed.proj.bias")


def to_vit_keys(state_dict):
    out = OrderedDict()
    for key, value in state_dict.items():
        if "encoder" in key:
            if "encoder_" in key:            # encoder_pos_embed, encoder_norm.*
                name = key.replace("encoder_", "")
            else:                            # encoder.<i>.*
                name = key.replace("encoder", "blocks")
            out[name] = value
        elif key in _KEPT:
            out[key] = value
    return out


def from_vit_keys(state_dict):
    out = OrderedDict()
    for key, value in state_dict.items():
        if key.startswith("blocks."):
            out["encoder." + key[len("blocks."):]] = value
        elif key == "pos_embed":
            out["encoder_pos_embed"] = value
        elif key.startswith("norm."):
            out["encoder_norm." + key[len("norm."):]] = value
        elif key in _KEPT:
            out[key] = value
    return out


def load_pretrain_checkpoint(path, map_location="cpu"):
    """-> (model_state_dict, rest) for a file written by util.misc.save_model here or by the reference."""
    import torch
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    if not isinstance(ckpt, dict) or "model" not in ckpt:
        raise ValueError(f"{path}: not a pre-training checkpoint (expected a dict with a 'model' entry)")
    return ckpt["model"], {k: v for k, v in ckpt.items() if k != "model"}
