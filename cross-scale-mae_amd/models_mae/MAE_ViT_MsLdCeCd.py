"""Headline model: MsLd + cross-decoder predictor loss + encoder NT-Xent contrastive loss
(reference models_mae/MAE_ViT_MsLdCeCd.py:7-84; tau = 0.5, cosine similarity, per-GPU negatives)."""
from .MAE_ViT_MsLd import MAE_ViT_MsLd, MAE_ViT_MsLd_PAIRED  # noqa: F401  (the reference imports both here, MAE_ViT_MsLdCeCd.py:2)
from .MAE_ViT_Shared import check_loss
from .MLP import MLP


class MAE_ViT_MsLdCeCd(MAE_ViT_MsLd):
    VARIANT = "MsLdCeCd"

    def __init__(self, loss_cd=None, predictor_hidden_size=2048, **kwargs):
        super().__init__(**kwargs)
        self.loss_cd = check_loss(loss_cd if loss_cd is not None else self.loss, "loss_cd")
        self.predictor_hidden_size = predictor_hidden_size
        self.predictor = MLP(self.decoder_embed_dim, self.num_patches, predictor_hidden_size)

    def forward(self, imgs, mask_ratio=0.75, contr_bs=None, mask_seed: int = None, return_embeds=False, consistent_mask=False, **kwargs):
        if contr_bs and contr_bs != imgs.shape[0]:
            raise ValueError("contr_bs must equal the batch size (the reference's NTXentLoss masks only fit bs == N)")
        return self._forward_ms(imgs, mask_ratio, mask_seed, return_embeds, consistent_mask)


class MAE_ViT_MsLdCeCd_PAIRED:
    def __init__(self, *a, **k):
        raise TypeError("MAE_ViT_MsLdCeCd_PAIRED cannot run in the reference either (NTXentLoss gets multiple values for 'cos_sim', "
                        "MAE_ViT_MsLdCeCd.py:152): out of scope (SURVEY.md Appendix E8)")
