"""ctypes binding of libcsmae_hip.so (the gfx950 C ABI declared in include/csmae.h).

There is deliberately NO fallback: if the shared library is missing or a kernel launch fails, the
product path raises.  PyTorch is used only for device memory, streams and torch.distributed.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_void_p

import torch  # noqa: F401  -- must come first: libcsmae_hip.so has to bind to the HIP runtime PyTorch already loaded (one runtime per process)

F32, BF16 = 0, 1
ABI_VERSION = 7
EPI_NONE, EPI_GELU, EPI_RESID, EPI_DGELU, EPI_ATOMIC = 0, 1, 2, 3, 4
LOSS_KINDS = {"mse": 0, "l2": 1, "mae": 2, "l1": 3, "bce": 4, "none": 5}
# the ssim family: kind -> (per-patch kind, pyramid levels, weight of the ssim term)  MAE_ViT_Shared.py:165-267
SSIM_KINDS = {"ssim": ("none", 1, 1.0), "ms_ssim": ("none", 5, 1.0), "mse_ssim": ("mse", 1, 0.1), "mse_ms_ssim": ("mse", 5, 0.1)}

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSMAE_LIB_PATH") or os.path.join(_HERE, "libcsmae_hip.so")   # (override: A/B builds of tools/)

I, L, P, F = c_int, c_longlong, c_void_p, c_float
_SIGNATURES = {
    "csmae_gemm": [I, I, I, L, L, L, P, L, P, L, P, L, I, P, I, P, L, P, L, I, P],
    "csmae_gemm_ks": [I, L, L, L, P, L, P, L, P, L, P, L, I, P, I, P, L, P, L, P],
    "csmae_weights_kslab": [I, P, I, P, P, P],
    "csmae_gemm_ln_supported": [L, L, L],
    "csmae_gemm_ln_fwd": [L, L, L, P, L, P, L, P, P, L, P, L, P, P, F, P, L, P, P, P],
    "csmae_gemm_ln_bwd": [L, L, L, P, L, P, L, P, L, P, P, P, P, L, P, L, P, L, P],
    "csmae_gemm_k2_mode": [I, I],
    "csmae_gemm_dw_mode": [I],
    "csmae_gemm_dw": [I, L, L, L, P, L, P, L, P, P, P, L, P],
    "csmae_gemm_dw_group": [I, I, L, P, P, P, P, P, P, P, P, I, P, L, P],
    "csmae_gemm_dw_group_fp8": [I, L, P, P, P, P, P, P, P, P, P, P, I, P, L, P],
    "csmae_fp8_amax": [I, L, I, P, L, P, P],
    "csmae_fp8_quantize": [I, I, I, L, I, P, L, P, L, P, P, P, P],
    "csmae_fp8_weights": [I, P, P, P, P, P, P, P],
    "csmae_gemm_fp8": [I, L, L, L, P, L, P, L, P, L, I, P, I, P, L, P, L, P, P, P, L, I, P, P, P, P],
    "csmae_gemm_force_tile": [I],
    "csmae_attn_fwd": [I, L, I, I, I, P, P, P, P],
    "csmae_attn_bwd": [I, L, I, I, I, P, P, P, P, P, P],
    "csmae_attn_resident": [I, I, I],
    "csmae_attn_fwd_q": [I, L, I, I, I, P, P, P, P, I, P, P, P, P],
    "csmae_attn_bwd_q": [I, L, I, I, I, P, P, P, P, P, P, I, P, P, P, P],
    "csmae_layernorm_fwd": [I, I, L, I, P, P, P, F, P, P, P, P, P, I, P, P, P, P],
    "csmae_layernorm_bwd": [I, I, I, L, I, P, P, P, P, P, P, P, P, P, P, P, L, P, I, P, P, P, P],
    "csmae_ln_param_reduce": [I, L, I, P, L, L, P, P, P],
    "csmae_ln_param_reduce_rows": [I, I, I, P, L, P, P, P],
    "csmae_bnrelu_fwd": [I, I, I, I, P, P, P, F, F, P, P, P, P, P, P, I, P],
    "csmae_bnrelu_bwd": [I, I, I, I, P, P, P, P, P, P, P, P, P, P],
    "csmae_crop_resize": [L, I, P, P, P, P],
    "csmae_mask_sort": [L, I, I, P, P, P, P, P, P],
    "csmae_patch_gather": [I, L, I, I, I, I, I, P, P, P, P, L, P],
    "csmae_embed_assemble": [I, L, I, I, P, P, P, P, P, P],
    "csmae_embed_assemble_bwd": [I, I, L, I, I, P, P, P, P],
    "csmae_unshuffle_fwd": [I, L, I, I, I, P, P, P, P, P, P],
    "csmae_unshuffle_bwd": [I, I, L, I, I, I, P, P, P, P, P],
    "csmae_rows_gather": [I, L, I, P, L, L, L, P, P],
    "csmae_rows_scatter_add": [I, L, I, P, F, L, L, L, P, P],
    "csmae_rows_scatter_add2": [I, L, I, P, F, L, P, F, L, L, L, P, P],
    "csmae_spec_fixup": [P, P, L, P, L, P, L, P, P, P, P, I, P],
    "csmae_rows_gather_idx": [L, I, I, I, P, P, L, P, P],
    "csmae_target_minmax": [I, L, I, I, I, I, P, P, P, P, P],
    "csmae_recon_loss_fwd": [I, I, I, L, I, I, I, I, P, P, P, L, P, P, P, P],
    "csmae_recon_loss_bwd": [I, I, I, I, L, I, I, I, I, P, P, P, L, P, P, P, P, F, P, P, L, P],
    "csmae_ssim_workspace_floats": [L, I, I, I, I, P],
    "csmae_ssim_fwd": [I, I, I, L, I, I, I, I, P, P, P, L, P, P, P, P],
    "csmae_ssim_apply": [I, I, F, F, P, P, P],
    "csmae_ssim_bwd": [I, L, I, I, I, I, P, L, P, P, F, P, P, P],
    "csmae_pair_loss_fwd": [I, L, I, P, L, L, L, P, L, L, L, P, P],
    "csmae_pair_loss_bwd": [I, I, L, I, P, L, L, L, P, L, L, L, P, F, P, P, P, P],
    "csmae_ntxent_fwd": [I, I, I, I, P, F, F, P, P, P, P, P, P],
    "csmae_ntxent_bwd": [I, I, P, P, P, P, F, F, P, P, P],
    "csmae_latent_grad_finish": [I, L, I, I, P, P, F, P, P],
    "csmae_loss_finalize": [L, I, P, P, F, P, F, P, F, P, I, P, P],
    "csmae_augment_u8": [L, I, I, I, I, P, P, P, P, P, P],
    "csmae_next_launch_event": [P],
    "csmae_flush_launch_event": [P],
    "csmae_adamw": [L, P, P, P, P, P, P, P, F, F, F, F, F, F, P, P, P, P, P],
    "csmae_adamw_fp8": [L, P, F, P, P, P, P, F, F, F, F, F, F, P, P, P, P, P, P, P, P],
    "csmae_gate_accumulate": [P, P, I, P],
    "csmae_clip_grad_norm": [L, P, F, P, P, P],
    "csmae_cast_f32_to_bf16": [L, P, P, P],
    "csmae_cast_bf16_to_f32": [L, P, P, P],
    "csmae_colsum": [I, L, I, P, L, P, P],
    "csmae_stream_create_cu_mask": [I, P, P],
    "csmae_stream_destroy": [P],
}

def debug_opt(name: str, default=None):
    """One gate for the settled A/B aids and experiment knobs: CSMAE_DEBUG="key[=value],key[=value],..." (INTEGRATION.md lists the keys; the C
    library reads the same variable).  Returns the value ("1" for a bare key) or `default`."""
    for item in os.environ.get("CSMAE_DEBUG", "").split(","):
        k, _, v = item.strip().partition("=")
        if k == name:
            return v or "1"
    return default


# variables that used to be read on their own and are now keys of CSMAE_DEBUG: setting one changes nothing any more, so say so once
_RETIRED_ENV = {"CSMAE_DW_SLOTS": "dw_slots", "CSMAE_DW_CUS": "dw_cus", "CSMAE_MAIN_CUS": "main_cus", "CSMAE_ZERO_MAIN": "zero_main",
                "CSMAE_BWD_MAIN_CUS": "bwd_main_cus", "CSMAE_DW_GROUP": "dw_group", "CSMAE_K2_STAGGER": "k2_stagger"}


def _warn_retired_env():
    import warnings
    for old, key in _RETIRED_ENV.items():
        if os.environ.get(old) is not None:
            warnings.warn(f"{old} is no longer read: use CSMAE_DEBUG={key}=... (INTEGRATION.md); this run uses the default", RuntimeWarning, stacklevel=3)


_lib = None


class CsmaeError(RuntimeError):
    pass


def load():
    """Load the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CsmaeError(f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()); "
                         "the MI355X path has no CPU/eager fallback")
    lib = ctypes.CDLL(LIB_PATH)
    lib.csmae_last_error.restype = ctypes.c_char_p
    lib.csmae_source_hash.restype = ctypes.c_char_p
    lib.csmae_abi_version.restype = c_int
    for name, sig in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = sig
        fn.restype = c_int
    if lib.csmae_abi_version() != ABI_VERSION:
        raise CsmaeError("libcsmae_hip ABI version mismatch")
    _warn_retired_env()
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        raise CsmaeError(f"{what} failed ({rc}): {load().csmae_last_error().decode()}")


def exported_symbols():
    return ["csmae_last_error", "csmae_abi_version", "csmae_source_hash"] + list(_SIGNATURES)


def source_hash() -> str:
    """sha256 of the kernel sources the loaded library was built from."""
    return load().csmae_source_hash().decode()
