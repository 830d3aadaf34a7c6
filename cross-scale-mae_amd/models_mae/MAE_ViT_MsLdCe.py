"""MsLd + cross-encoder predictor (reference models_mae/MAE_ViT_MsLdCe.py).  In the reference this variant cannot run with any
mask_ratio > 0: BatchNorm1d(num_patches) is fed len_keep channels and raises.  Constructor, parameters
and signature are kept; forward raises the equivalent error up front."""
from .MAE_ViT_MsLd import MAE_ViT_MsLd
from .MAE_ViT_Shared import check_loss
from .MLP import MLP


class MAE_ViT_MsLdCe(MAE_ViT_MsLd):
    VARIANT = "MsLdCe"

    def __init__(self, loss_ce=None, predictor_hidden_size=2048, **kwargs):
        super().__init__(**kwargs)
        self.loss_ce = check_loss(loss_ce if loss_ce is not None else self.loss, "loss_ce")
        self.predictor_hidden_size = predictor_hidden_size
        self.predictor = MLP(self.dim_model, self.num_patches, predictor_hidden_size)

    def forward(self, imgs, mask_ratio=0.75, mask_seed: int = None, return_embeds=False, consistent_mask=False):
        keep = int(self.num_patches * (1 - mask_ratio))
        raise RuntimeError(f"running_mean should contain {keep} elements not {self.num_patches}: MAE_ViT_MsLdCe's BatchNorm1d(num_patches) "
                           "predictor cannot consume the kept encoder tokens — the reference raises the same (MAE_ViT_MsLdCe.py:21,46); "
                           "use MAE_ViT_MsLdCeCd")
