"""Shared by the CPU (oracle) and GPU (HIP path) tests of tests/golden/fullsize.{json,npz}: MAE_ViT_MsLdCeCd at the geometries
BASELINE.json's configs[1..4] are quoted on, produced by the reference itself (oracle/gen_golden.py:g_fullsize).  The inputs are
regenerated from their seeds (2.4 MB of incompressible noise per config is not committed); the fixture's checksums tell a wrong
regeneration apart from a wrong result."""
import json
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
TAGS = ("vitb16_224", "vitl16_224", "vitl16_256c4", "vith14_224")   # configs[1], [2], [3], [4]


def checksum(t):
    flat = t.double().reshape(-1)
    w = torch.arange(flat.numel(), dtype=torch.float64) % 97 + 1
    return [float(flat.sum()), float(flat.abs().sum()), float((flat * w).sum())]


def load(tag):
    meta = json.load(open(os.path.join(G, "fullsize.json")))[tag]
    d = np.load(os.path.join(G, "fullsize.npz"), allow_pickle=False)
    return meta, {k[len(tag) + 1:]: d[k] for k in d.files if k.startswith(tag + "_")}


def inputs(tag, meta):
    """Same recipe as oracle/gen_golden.py:fullsize_inputs (seed 1000 + index of the tag in sorted order)."""
    S, p, C, N = meta["input_size"], meta["patch"], meta["channels"], meta["N"]
    g = torch.Generator().manual_seed(1000 + sorted(TAGS).index(tag))
    imgs = torch.randn(N, C, S, S, generator=g)
    got = checksum(imgs)
    assert np.allclose(got, meta["imgs_checksum"], rtol=1e-12, atol=1e-9), "this torch's CPU generator does not reproduce the fixture inputs"
    return imgs


def seeded_model(meta, device="cpu"):
    """The drop-in class with the reference's seeded default initialisation (bit-identical: checked against the fixture)."""
    import models_mae
    torch.manual_seed(0)
    m = models_mae.MAE_ViT_MsLdCeCd(**meta["geom"], input_size=meta["input_size"], patch_size=str(meta["patch"]), input_channels=meta["channels"],
                                    loss="mse", device=device)
    sd = m.state_dict()
    for k, c in meta["weights"].items():
        assert np.allclose(checksum(sd[k]), c, rtol=1e-12, atol=1e-9), f"seeded init of {k} differs from the reference's"
    return m
