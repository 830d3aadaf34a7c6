"""Drop-in `models_mae` package for the Cross-Scale MAE pre-training path on MI355X.

Same class / factory names, constructor keywords (unknown keys are swallowed, so `models_mae.__dict__[args.model](**vars(args))`
works — main_pretrain.py:398), presets and `state_dict` keys as the reference's `models_mae/__init__.py:23-162`.
The reference's `*_cross*` / `*shunted*` factories reference modules that were never committed (`__init__.py:16-19`) and are
omitted.  `mae_vit_{large,huge}_MsLdCeCd` are build-side additions for BASELINE.json configs 3-5."""
from .MAE_ViT_Baseline import MAE_ViT_Baseline
from .MAE_ViT_MsLd import MAE_ViT_MsLd, MAE_ViT_MsLd_PAIRED
from .MAE_ViT_MsLdCd import MAE_ViT_MsLdCd
from .MAE_ViT_MsLdCe import MAE_ViT_MsLdCe
from .MAE_ViT_MsLdCeCd import MAE_ViT_MsLdCeCd
from .MAE_ViT_MsLdLe import MAE_ViT_MsLdLe
from .MAE_ViT_MsLdLeCd import MAE_ViT_MsLdLeCd

args_mae_vit_tiny = {"dim_model": 128, "encoder_num_layers": 4, "encoder_num_heads": 8, "decoder_embed_dim": 256,
                     "decoder_num_layers": 4, "decoder_num_heads": 8}
args_mae_vit_small = {"dim_model": 512, "encoder_num_layers": 8, "encoder_num_heads": 8, "decoder_embed_dim": 512,
                      "decoder_num_layers": 8, "decoder_num_heads": 16}
args_mae_vit_base = {"dim_model": 768, "encoder_num_layers": 12, "encoder_num_heads": 12, "decoder_embed_dim": 512,
                     "decoder_num_layers": 8, "decoder_num_heads": 16}
args_mae_vit_large = {"dim_model": 1024, "encoder_num_layers": 24, "encoder_num_heads": 16, "decoder_embed_dim": 512,
                      "decoder_num_layers": 8, "decoder_num_heads": 16}
args_mae_vit_huge = {"dim_model": 1280, "encoder_num_layers": 32, "encoder_num_heads": 16, "decoder_embed_dim": 512,
                     "decoder_num_layers": 8, "decoder_num_heads": 16}


def mae_vit_base(**kwargs):
    return MAE_ViT_Baseline(**args_mae_vit_base, **kwargs)


def mae_vit_base_MsLd(**kwargs):
    return MAE_ViT_MsLd(**args_mae_vit_base, **kwargs)


def mae_vit_base_MsLdLe(**kwargs):
    return MAE_ViT_MsLdLe(**args_mae_vit_base, **kwargs)


def mae_vit_base_MsLdCd(**kwargs):
    return MAE_ViT_MsLdCd(**args_mae_vit_base, **kwargs)


def mae_vit_base_MsLdCe(**kwargs):
    return MAE_ViT_MsLdCe(**args_mae_vit_base, **kwargs)


def mae_vit_base_MsLdLeCd(**kwargs):
    return MAE_ViT_MsLdLeCd(**args_mae_vit_base, **kwargs)


def mae_vit_base_MsLdCeCd(**kwargs):
    return MAE_ViT_MsLdCeCd(**args_mae_vit_base, **kwargs)


def mae_vit_large(**kwargs):
    return MAE_ViT_Baseline(dim_model=1024, **kwargs)  # as the reference: constructor defaults are the ViT-L geometry


def mae_vit_huge(**kwargs):
    return MAE_ViT_Baseline(**args_mae_vit_huge, **kwargs)


def mae_vit_large_MsLdCeCd(**kwargs):
    return MAE_ViT_MsLdCeCd(**args_mae_vit_large, **kwargs)


def mae_vit_huge_MsLdCeCd(**kwargs):
    return MAE_ViT_MsLdCeCd(**args_mae_vit_huge, **kwargs)
