"""Multi-scale wrapper (reference models_mae/MAE_ViT_MsLd.py:8-77): one RandomResizedCrop box per batch, two views, summed
(or averaged) reconstruction losses.  Both views run as one 2N batch on the MI355X."""
import math

import torch

from .MAE_ViT_Baseline import MAE_ViT_Baseline


def sample_crop_box(size, scale, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """Box of torchvision 0.15.1 RandomResizedCrop.get_params on a size x size image, drawn from the global CPU torch RNG
    exactly like the reference does (MAE_ViT_MsLd.py:29-35,52): up to 10 (area, log-ratio) proposals, then a centre crop."""
    area = float(size * size)
    lo, hi = math.log(ratio[0]), math.log(ratio[1])
    for _ in range(10):
        target = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect = math.exp(torch.empty(1).uniform_(lo, hi).item())
        w = int(round(math.sqrt(target * aspect)))
        h = int(round(math.sqrt(target / aspect)))
        if 0 < w <= size and 0 < h <= size:
            top = torch.randint(0, size - h + 1, size=(1,)).item()
            left = torch.randint(0, size - w + 1, size=(1,)).item()
            return top, left, h, w
    return 0, 0, size, size  # aspect 1 is inside [3/4, 4/3]: the fallback is the whole (square) image


class MAE_ViT_MsLd(MAE_ViT_Baseline):
    """Masked Autoencoder with VisionTransformer backbone"""

    VARIANT = "MsLd"

    def __init__(self, ms_range=(0.25, 0.75), ms_decoder_loss_reduction: str = "sum", **kwargs):
        super().__init__(**kwargs)
        self.ms_decoder_loss_reduction = ms_decoder_loss_reduction.lower()
        self.allowed_reductions = ["mean", "sum"]
        assert self.ms_decoder_loss_reduction in self.allowed_reductions, f"ms_decoder_loss_reduction must be one of: {self.allowed_reductions}"
        self.ms_range = tuple(ms_range)
        self.crop = torch.nn.Sequential()  # placeholder child keeping the reference's module registration order (no parameters)
        self._box_host = torch.zeros(4, dtype=torch.int32)

    def _draw(self, imgs, mask_ratio, mask_seed, consistent_mask=False):
        """Draw order of the reference (MAE_ViT_MsLd.py:45-61): [seed] -> crop box (CPU RNG) -> [seed] rand(N,L) -> [seed] rand(N,L)."""
        N, L = imgs.shape[0], self.num_patches
        hook, self._test_draws = self._test_draws, None
        if mask_seed is not None:
            torch.manual_seed(mask_seed)
        elif consistent_mask:
            mask_seed = torch.randint(0, 2 ** 32 - 1, (1,)).item()
        box = hook["box"] if hook else sample_crop_box(self.input_size, self.ms_range)
        self.last_crop_box = tuple(int(v) for v in box)
        noises = []
        for v in range(2):
            if mask_seed is not None:
                torch.manual_seed(mask_seed)
            noises.append(hook["noise"][v].to(imgs.device) if hook else torch.rand(N, L, device=imgs.device))
        self._box_host = torch.tensor(self.last_crop_box, dtype=torch.int32)
        return torch.cat(noises, dim=0), self._box_host

    def _outputs(self, eng, ws, N):
        c = eng.cfg
        pred = ws.pred.view(ws.B2, ws.Td, c["P"])
        lat = (ws.lat32 if ws.lat32 is not None else ws.enc["x"][c["Ne"]]).view(ws.B2, ws.Te, c["D"])   # fp32 (copy of a bf16 stream's latent)
        emb = ws.emb32.view(ws.B2, ws.Td, c["Dd"])
        return (ws.losses[0].clone(), pred[:N, 1:, :], ws.mask[:N], lat[:N], lat[N:], emb[:N], emb[N:])

    def _forward_ms(self, imgs, mask_ratio, mask_seed, return_embeds, consistent_mask):
        noise, box = self._draw(imgs, mask_ratio, mask_seed, consistent_mask)
        loss, pred, mask, eo, ec, do, dc = self._run(imgs, mask_ratio, noise, box)
        return (loss, pred, mask) if not return_embeds else (loss, pred, mask, (eo, ec), (do, dc))

    def forward(self, imgs, mask_ratio=0.75, mask_seed: int = None, return_embeds=False, consistent_mask=False):
        return self._forward_ms(imgs, mask_ratio, mask_seed, return_embeds, consistent_mask)


class MAE_ViT_MsLd_PAIRED(MAE_ViT_Baseline):
    """Two explicit views instead of the random crop (reference MAE_ViT_MsLd.py:79-146): `forward(imgs1, imgs2, ...)`, the same summed (or
    averaged) reconstruction loss; returns the first view's prediction and mask.  No `crop` child (commented out in the reference), so the
    module tree and `state_dict` are MAE_ViT_Baseline's."""

    VARIANT = "MsLd"

    def __init__(self, ms_range=(0.2, 0.8), ms_decoder_loss_reduction: str = "sum", **kwargs):
        super().__init__(**kwargs)
        self.ms_decoder_loss_reduction = ms_decoder_loss_reduction.lower()
        self.allowed_reductions = ["mean", "sum"]
        assert self.ms_decoder_loss_reduction in self.allowed_reductions, f"ms_decoder_loss_reduction must be one of: {self.allowed_reductions}"

    _outputs = MAE_ViT_MsLd._outputs

    def _draw(self, imgs, mask_ratio, mask_seed, consistent_mask=False):
        """Draw order of the reference (MAE_ViT_MsLd.py:119-133): [seed] -> [seed] rand(N,L) -> [seed] rand(N,L); no crop box."""
        N, L = imgs.shape[0], self.num_patches
        hook, self._test_draws = self._test_draws, None
        if mask_seed is not None:
            torch.manual_seed(mask_seed)
        elif consistent_mask:
            mask_seed = torch.randint(0, 2 ** 32 - 1, (1,)).item()
        noises = []
        for v in range(2):
            if mask_seed is not None:
                torch.manual_seed(mask_seed)
            noises.append(hook["noise"][v].to(imgs.device) if hook else torch.rand(N, L, device=imgs.device))
        return torch.cat(noises, dim=0), None

    def forward(self, imgs1, imgs2, mask_ratio=0.75, mask_seed: int = None, return_embeds=False, consistent_mask=False):
        noise, _ = self._draw(imgs1, mask_ratio, mask_seed, consistent_mask)
        loss, pred, mask, eo, ec, do, dc = self._run(imgs1, mask_ratio, noise, None, img1=imgs2)
        return (loss, pred, mask) if not return_embeds else (loss, pred, mask, (eo, ec), (do, dc))
