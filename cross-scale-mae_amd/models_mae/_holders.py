"""Parameter holders with timm 0.4.12's names and construction order (Block / PatchEmbed — call sites
MAE_ViT_Baseline.py:75-77,160-188 of the reference).  They own weights only: every FLOP runs in libcsmae_hip,
their `forward` is never used.  Same registration order => same `state_dict` keys and the same seeded init."""
import torch.nn as nn


class _Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, norm_layer):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = _Attention(dim, num_heads)
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, *a, **k):
        raise RuntimeError("Block is a parameter holder; compute runs in csmae_hip.Engine")


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, *a, **k):
        raise RuntimeError("PatchEmbed is a parameter holder; compute runs in csmae_hip.Engine")
