"""MsLd + latent loss between the two views' encoder outputs (reference models_mae/MAE_ViT_MsLdLe.py:44-47)."""
from .MAE_ViT_MsLd import MAE_ViT_MsLd
from .MAE_ViT_Shared import check_loss


class MAE_ViT_MsLdLe(MAE_ViT_MsLd):
    VARIANT = "MsLdLe"

    def __init__(self, loss_e=None, **kwargs):
        super().__init__(**kwargs)
        self.loss_e = check_loss(loss_e if loss_e is not None else self.loss, "loss_e")

    def forward(self, imgs, mask_ratio=0.75, mask_seed: int = None, return_embeds=False, consistent_mask=False, targets=None):
        return self._forward_ms(imgs, mask_ratio, mask_seed, return_embeds, consistent_mask)
