"""MsLd + cross-decoder predictor loss (reference models_mae/MAE_ViT_MsLdCd.py:49-54)."""
from .MAE_ViT_MsLd import MAE_ViT_MsLd
from .MAE_ViT_Shared import check_loss
from .MLP import MLP


class MAE_ViT_MsLdCd(MAE_ViT_MsLd):
    VARIANT = "MsLdCd"

    def __init__(self, loss_cd=None, predictor_hidden_size=2048, **kwargs):
        super().__init__(**kwargs)
        self.loss_cd = check_loss(loss_cd if loss_cd is not None else self.loss, "loss_cd")
        self.predictor_hidden_size = predictor_hidden_size
        self.predictor = MLP(self.decoder_embed_dim, self.num_patches, predictor_hidden_size)
