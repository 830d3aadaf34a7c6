"""2-D sin-cos positional table (reference util/pos_embed.py:16-63): float64 numpy, "w goes first", cls row = 0."""
import numpy as np


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    assert embed_dim % 4 == 0
    omega = 1.0 / 10000 ** (np.arange(embed_dim // 4, dtype=float) / (embed_dim / 4.0))
    hh, ww = np.divmod(np.arange(grid_size * grid_size), grid_size)
    aw, ah = np.outer(ww.astype(float), omega), np.outer(hh.astype(float), omega)
    table = np.concatenate([np.sin(aw), np.cos(aw), np.sin(ah), np.cos(ah)], axis=1)
    return np.concatenate([np.zeros([1, embed_dim]), table], axis=0) if cls_token else table
