"""ctypes binding of libskg.so (the C ABI declared in include/skg.h).

The library is the product's only compute path.  If it is missing or does not export a declared
symbol this module raises at import time: there is no CPU / PyTorch fallback.
"""
from __future__ import annotations

import ctypes
import os

# Load order matters: PyTorch-ROCm ships its own libamdhip64 under torch/lib, libskg.so is linked against the SONAME
# libamdhip64.so.7.  Whichever is loaded first decides which HIP runtime the process gets; if libskg.so came first, its
# kernels would be launched through a second runtime instance that never saw torch's device / streams
# ("no ROCm-capable device is detected" at the first launch).  Importing torch first makes libskg.so bind to the runtime
# torch already loaded - the one that owns the tensors and streams every launch uses.
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
# SKG_LIB: another build of the same ABI (same-box A/B measurements of kernel variants); default = the in-tree library
LIB_PATH = os.environ.get("SKG_LIB") or os.path.join(_HERE, "libskg.so")

# include/skg.h SKG_ABI_VERSION: bumped whenever an entry point changes its signature (2: skg_attn_bwd_dq / _dkv lost
# their transposed-operand pointers, skg_set_workspace became per stream; 3: the accuracy-mode entry points with GroupNorm
# statistics - skg_*_hilo_gn, skg_groupnorm_fwd_hilo / _from_partial_hilo, skg_ff_block_f16_hilo - and the pair offset of
# skg_cfg_ddim_step / skg_cfg_dpmpp2m_step; 5: skg_box_probe_mfma, skg_conv3x3_sc_f16 declines instead of failing on the 2 GiB
# operand limit), so that a stale build selected through
# SKG_LIB fails at load instead of receiving shifted arguments
ABI_VERSION = 5

# spec letters: p = device/host pointer, i = int, f = float, u = unsigned, z = size_t (return only)
SIGNATURES = {
    "skg_abi_version": ("i", ""),
    "skg_last_error": ("s", ""),
    "skg_gemm_f16": ("i", "pipipiiiippifup"),
    "skg_conv3x3_up2_f16": ("i", "pippiiiiiipp"),
    "skg_conv3x3_up2_f16_hilo": ("i", "pipppiiiiiipp"),
    "skg_conv3x3_up2_f16_pairout": ("i", "pipppiiiiiipp"),
    "skg_conv3x3_sc_f16": ("i", "pipiipppiiiiiipupip"),
    "skg_conv4x4s2_f16": ("i", "pippiiiiiipp"),
    "skg_conv3x3_wino_v_bytes": ("z", "iiii"),
    "skg_conv3x3_wino_f16": ("i", "pippppiiiiiipppiup"),
    "skg_groupnorm_wino_fwd": ("i", "pipiiiiifppipp"),
    "skg_gemm_f16_rows": ("i", "pipipiiiipiip"),
    "skg_gemm_f16_hilo": ("i", "pipippiiiipppifup".replace(" ", "")),
    "skg_conv3x3_f16_hilo": ("i", "pipppiiiiiiipppifup"),
    "skg_gemm_f16_hilo_gn": ("i", "pipippiiiipppifupiip"),
    "skg_conv3x3_f16_hilo_gn": ("i", "pipppiiiiiiipppifupip"),
    "skg_groupnorm_fwd_hilo": ("i", "ppippiiiiifppippp"),
    "skg_groupnorm_from_partial_hilo": ("i", "ppippiiiiiifppippipiip"),
    "skg_ff_block_f16_hilo": ("i", "ppippiiiippfpppppiip"),
    "skg_groupnorm_apply_hilo": ("i", "ppipiiiiippp ip".replace(" ", "")),
    "skg_layernorm_fwd_hilo": ("i", "ppipiiippfpp"),
    "skg_gemm_f16_gn": ("i", "pipipiiiippifupiip"),
    "skg_gemm_gn_fused": ("i", "iiiiiii"),
    "skg_gemm_f16_geglu_keep": ("i", "pipipipiiiipp"),
    "skg_ff_block_f16": ("i", "pipiiiippfppppp"),
    "skg_ff_block_f16_keep": ("i", "pipiiiippfpppppiip"),
    "skg_ff_block_proj_f16": ("i", "pipiiiippfpppppippiipiip"),
    "skg_ff_block_proj_f16_hilo": ("i", "ppippiiiippfppppppippiipiip"),
    "skg_xattn_block_f16": ("i", "pipiiiiiippfpppfp"),
    "skg_xattn_block_f16_hilo": ("i", "ppippiiiiiippfpppfp"),
    "skg_xattn_block_f16_hilo_keep": ("i", "ppippiiiiiippfpppfpppipip"),
    "skg_xattn_block_f16_keep": ("i", "pipiiiiiippfpppfpppipip"),
    "skg_gemm_variant": ("i", "iiiii"),
    "skg_set_workspace": ("i", "pzp"),
    "skg_conv3x3_f16": ("i", "pippiiiiiiippifup"),
    "skg_conv3x3_f16_gn": ("i", "pippiiiiiiippifupip"),
    "skg_groupnorm_scratch_floats": ("z", "ii"),
    "skg_groupnorm_stats": ("i", "piiiiifppp"),
    "skg_groupnorm_fwd": ("i", "pipiiiiifppippp"),
    "skg_groupnorm_from_partial": ("i", "pipiiiiifppippip"),
    "skg_groupnorm_from_partial2": ("i", "pipiiiiiifppippipiip"),
    "skg_groupnorm_apply": ("i", "pipiiiiipppip"),
    "skg_groupnorm_bwd": ("i", "pipipipiiiiipppipp"),
    "skg_layernorm_fwd": ("i", "pipiiippfpp"),
    "skg_layernorm_bwd": ("i", "pipipipiiippp"),
    "skg_geglu_fwd": ("i", "pipiiiip"),
    "skg_geglu_bwd": ("i", "pipipiiiip"),
    "skg_attn_fwd": ("i", "pipipipipiiiiiifp"),
    "skg_attn_fwd_causal": ("i", "pipipipipiiiiiifp"),
    "skg_attn_fwd_rowv": ("i", "pipipipipiiiiiifp"),
    "skg_attn_bwd_delta": ("i", "pipipiiiip"),
    "skg_attn_bwd_dq": ("i", "pipipipipppiiiiiiifp"),
    "skg_attn_bwd_dq_delta": ("i", "pipipipipipppiiiiiiifp"),
    "skg_attn_bwd_dkv": ("i", "pipipipipppipiiiiiifp"),
    "skg_transpose_f16": ("i", "pipiiip"),
    "skg_axpby_f16": ("i", "pipipiiiffp"),
    "skg_batch_copy_f16": ("i", "piipiiiiip"),
    "skg_silu_f16": ("i", "pipiiip"),
    "skg_quick_gelu_f16": ("i", "pipiiip"),
    "skg_gelu_f16": ("i", "pipiiip"),
    "skg_sumpool2x2_f16": ("i", "pipiiiiip"),
    "skg_nchw_f32_to_nhwc_f16": ("i", "ppiiiip"),
    "skg_nhwc_f16_to_nchw_f32": ("i", "pipiiip"),
    "skg_lgp_layer0_gather": ("i", "pipippfipiiip"),
    "skg_lgp_layer0_scatter": ("i", "pipiiiip"),
    "skg_bn_scratch_floats": ("z", "ii"),
    "skg_bn_stats": ("i", "piiiiifppppp"),
    "skg_bn_stats_from_running": ("i", "ppiifpp"),
    "skg_bn_apply": ("i", "pipiiiiipppp"),
    "skg_bn_relu_bwd": ("i", "pipipiiiiippipp"),
    "skg_colsum_scratch_floats": ("z", "i"),
    "skg_colsum_f16": ("i", "piiifppp"),
    "skg_bn_param_grads": ("i", "pipiiipfpppp"),
    "skg_lgp_extra_features": ("i", "pfiiipip"),
    "skg_lgp_mse_train": ("i", "pippipiifp"),
    "skg_adamw_step": ("i", "pppppzfffffifp"),
    "skg_lgp_mse_seed": ("i", "pippipiifp"),
    "skg_cfg_ddim_step": ("i", "ppiipppiifffffip"),
    "skg_softmax_rows_f16": ("i", "pipiiip"),
    "skg_image_postprocess": ("i", "pipziffp"),
    "skg_image_to_u8": ("i", "pipziffp"),
    "skg_gaussian_sample": ("i", "pippiiifp"),
    "skg_cfg_dpmpp2m_step": ("i", "ppiippppiiffffffip"),
    "skg_guidance_update": ("i", "pipppiifp"),
    "skg_box_probe_mfma": ("i", "pip"),
}

_CT = {"p": ctypes.c_void_p, "i": ctypes.c_int, "f": ctypes.c_float, "u": ctypes.c_uint,
       "z": ctypes.c_size_t, "s": ctypes.c_char_p}


class SkgTap(ctypes.Structure):
    _fields_ = [("P", ctypes.c_void_p), ("s", ctypes.c_int), ("pad_", ctypes.c_int)]


class SkgError(RuntimeError):
    rc = 0                      # the C ABI's return code (-2 = SKG_E_UNSUPPORTED: nothing was launched)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C sketch2img_amd/csrc`.  sketch2img_amd has no fallback compute path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (ret, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise ImportError(f"libskg.so does not export {name}") from e
        fn.restype = _CT[ret]
        fn.argtypes = [_CT[a] for a in args]
    if lib.skg_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.skg_abi_version()}, this package binds version {ABI_VERSION} "
                          "(include/skg.h SKG_ABI_VERSION) - rebuild with `make -C sketch2img_amd/csrc`")
    return lib


lib = _load()

_ERR = {-1: "SKG_E_BADARG (shape/alignment precondition violated)", -2: "SKG_E_UNSUPPORTED",
        -3: "SKG_E_LAUNCH"}


def check(rc: int, what: str):
    if rc != 0:
        detail = lib.skg_last_error().decode() if rc == -3 else ""
        err = SkgError(f"{what}: {_ERR.get(rc, rc)} {detail}")
        err.rc = rc
        raise err
