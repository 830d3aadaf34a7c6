"""Shared base of the MAE family (reference models_mae/MAE_ViT_Shared.py): loss selection, patchify helpers,
random_masking.  The step itself runs in csmae_hip.Engine (HIP); helpers here are layout/API surface only."""
import torch
import torch.nn as nn

SUPPORTED_LOSSES = ("mse", "l2", "mae", "l1", "bce")
SSIM_LOSSES = ("ssim", "ms_ssim", "mse_ssim", "mse_ms_ssim")  # MAE_ViT_Shared.py:165-267: reconstruction head only


def check_loss(name, what="loss"):
    """Loss names as the reference resolves them (`getattr(self, f"forward_loss_{name}")`, MAE_ViT_Shared.py:19).  The ssim family works
    on images (un-patchify with the patch size, a 3-channel mask): as the cross-decoder / latent / cross-encoder loss the reference
    fails inside its first forward (`unpatchify(x, None, None)`); here the constructor says so."""
    name = name.lower()
    if name in SSIM_LOSSES:
        if what != "loss":
            raise ValueError(f"{what}={name!r}: the ssim family compares images and only serves the reconstruction loss "
                             f"(MAE_ViT_Shared.py:181-185 un-patchifies both operands); pass an explicit per-element {what} (mse, l2, mae, l1)")
        return name
    if name not in SUPPORTED_LOSSES:
        raise AttributeError(f"forward_loss_{name}")  # what getattr() raises in the reference (MAE_ViT_Shared.py:19)
    return name


class MAE_ViT_Shared(nn.Module):
    def __init__(self, norm_pix_loss=False, loss="mse", **kwargs):
        super().__init__()
        self.loss = check_loss(loss)
        self.norm_pix_loss = norm_pix_loss

    # ---- layout helpers (MAE_ViT_Shared.py:24-55): "nchpwq->nhwpqc" and back
    def patchify(self, imgs, p, c):
        assert imgs.shape[2] == imgs.shape[3] and imgs.shape[2] % p == 0
        n, g = imgs.shape[0], imgs.shape[2] // p
        return imgs.reshape(n, c, g, p, g, p).permute(0, 2, 4, 3, 5, 1).reshape(n, g * g, p * p * c)

    def unpatchify(self, x, p, c):
        g = int(x.shape[1] ** 0.5)
        assert g * g == x.shape[1]
        return x.reshape(x.shape[0], g, g, p, p, c).permute(0, 5, 1, 3, 2, 4).reshape(x.shape[0], c, g * p, g * p)

    def random_masking(self, x, mask_ratio):
        """MAE_ViT_Shared.py:57-84.  Indices come from the HIP rank-sort (stable ascending == argsort on tie-free rows), the kept rows
        from a HIP row gather (inside the step both are fused into patch_gather).  The HIP gather is inference-only (no autograd node):
        inputs that require grad go through torch.gather."""
        from csmae_hip import ops
        N, L, D = x.shape
        keep = int(L * (1 - mask_ratio))
        noise = torch.rand(N, L, device=x.device)
        ids_restore = torch.empty(N, L, device=x.device, dtype=torch.long)
        mask = torch.empty(N, L, device=x.device)
        ids_keep = torch.empty(N, max(keep, 1), device=x.device, dtype=torch.int32)
        ops.mask_sort(noise, keep, ids_restore, mask, ids_keep)
        if keep < 1:
            return x.new_empty(N, 0, D), mask, ids_restore
        if torch.is_grad_enabled() and x.requires_grad:
            #  the HIP gather is a raw kernel outside autograd: a caller that differentiates through the masked tokens — as the
            # reference's forward_encoder does — takes torch.gather with the same indices instead of silently losing the gradient
            idx = ids_keep[:, :keep].long().unsqueeze(-1).expand(-1, -1, D)
            return torch.gather(x, 1, idx), mask, ids_restore
        x32 = x.contiguous().float()
        x_masked = ops.rows_gather_idx(x32, ids_keep, keep, torch.empty(N, keep, D, device=x.device, dtype=torch.float32))
        return x_masked.to(x.dtype), mask, ids_restore

    def scale_01(self, x):
        return (x - x.min()) / (x.max() - x.min() + 1.0e-6)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {}
