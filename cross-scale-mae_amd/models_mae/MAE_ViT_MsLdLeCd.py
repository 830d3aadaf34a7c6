"""MsLd + latent loss + cross-decoder predictor loss (reference models_mae/MAE_ViT_MsLdLeCd.py:57-65)."""
from .MAE_ViT_MsLd import MAE_ViT_MsLd
from .MAE_ViT_Shared import check_loss
from .MLP import MLP


class MAE_ViT_MsLdLeCd(MAE_ViT_MsLd):
    VARIANT = "MsLdLeCd"

    def __init__(self, loss_e=None, loss_cd=None, predictor_hidden_size=2048, **kwargs):
        super().__init__(**kwargs)
        self.loss_e = check_loss(loss_e if loss_e is not None else self.loss, "loss_e")
        self.loss_cd = check_loss(loss_cd if loss_cd is not None else self.loss, "loss_cd")
        self.predictor_hidden_size = predictor_hidden_size
        self.predictor = MLP(self.decoder_embed_dim, self.num_patches, predictor_hidden_size)

    def forward(self, imgs, mask_ratio=0.75, mask_seed: int = None, return_embeds=False, consistent_mask=False, targets=None):
        return self._forward_ms(imgs, mask_ratio, mask_seed, return_embeds, consistent_mask)
