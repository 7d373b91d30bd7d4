"""ctypes binding of libinklayer_hip.so (the C ABI declared in include/inklayer_hip.h).

The product path has NO fallback: if the library is missing or fails to load,
`lib()` raises.  Nothing under oracle/ is ever imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libinklayer_hip.so"
_lib = None

c_void_p, c_int, c_i64, c_float = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class InkGemm(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("bias", c_void_p), ("col_scale", c_void_p),
        ("residual", c_void_p), ("row_map", c_void_p), ("C", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_int), ("ldw", c_int), ("ldr", c_int), ("ldc", c_int),
        ("act", c_int), ("c_f16", c_int),
        ("C_lo", c_void_p), ("res_hi", c_void_p), ("res_lo", c_void_p), ("stats_out", c_void_p),
        ("ln_stats", c_void_p), ("ln_colsum", c_void_p),
        ("stats_parts", c_int), ("ln_parts", c_int), ("ln_dim", c_int), ("ln_eps", c_float),
    ]


class InkAttn(C.Structure):
    _fields_ = [
        ("Q", c_void_p), ("K", c_void_p), ("V", c_void_p), ("O", c_void_p),
        ("ldq", c_i64), ("ldk", c_i64), ("ldv", c_i64), ("ldo", c_i64),
        ("n_batch", c_int), ("n_heads", c_int), ("n_q", c_int), ("n_k", c_int),
        ("head_dim", c_int), ("scale", c_float), ("bias_mode", c_int), ("grid_w", c_int),
        ("q_batch_rows", c_void_p), ("kv_batch_rows", c_void_p),
        ("rel_h", c_void_p), ("rel_w", c_void_p), ("rel_aug", c_void_p),
        ("dense_bias", c_void_p), ("dense_mask", c_void_p), ("n_mask", c_int), ("rel_f16", c_int),
        ("tok_rows", c_void_p), ("pad_k", c_void_p), ("pad_v", c_void_p),
    ]


# name -> argtypes; every function returns int (0 ok / 1 bad argument / 2 launch failure)
SIGNATURES = {
    "ink_abi_version": [],
    "ink_gemm_f16": [C.POINTER(InkGemm), c_void_p],
    "ink_gemm_set_variant": [c_int],
    "ink_gemm_query_variant": [c_int, c_int, c_int],
    "ink_gemm_query_stats_chunk": [c_int, c_int, c_int],
    "ink_hilo_split_stats": [c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_int, c_void_p],
    "ink_hilo_join": [c_void_p, c_void_p, c_i64, c_void_p, c_void_p],
    "ink_layernorm_rows": [c_void_p, c_i64, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int,
                           c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_i64, c_void_p, c_int, c_void_p],
    "ink_add_split_f16": [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p],
    "ink_add_cvt_f16": [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p],
    "ink_add_f32": [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p],
    "ink_flash_attn": [C.POINTER(InkAttn), c_void_p],
    "ink_sam_patchify": [c_void_p, c_int, c_int, c_int, c_int, C.POINTER(c_float), C.POINTER(c_float),
                         c_int, c_int, c_void_p, c_void_p],
    "ink_im2col3x3_f16": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "ink_resize_bilinear_u8": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int,
                               c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "ink_sam_pe_encode": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p],
    "ink_sam_mask_logits": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "ink_sam_postprocess": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                            c_void_p, c_void_p, c_void_p],
    "ink_ms_deform_attn_forward": [c_void_p, C.POINTER(c_i64), C.POINTER(c_i64), c_void_p, c_void_p, c_int,
                                   c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "ink_msda_fused": [c_void_p, c_void_p, c_i64, c_void_p, c_int, c_i64, c_i64, C.POINTER(c_int), c_int,
                       c_int, c_int, c_void_p, c_void_p],
    "ink_swin_patchify": [c_void_p, c_int, c_int, C.POINTER(c_float), C.POINTER(c_float), c_void_p, c_void_p],
    "ink_layernorm_merge4": [c_void_p, c_i64, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p,
                             c_void_p],
    "ink_groupnorm_nhwc": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p,
                           c_void_p, c_i64, c_void_p],
    "ink_gather_rows": [c_void_p, c_i64, c_i64, c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_void_p,
                        c_void_p],
    "ink_biattn_fusion": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                          c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    "ink_biattn_colstats": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ink_fusion_fold_workspace": [c_int, c_int, C.POINTER(c_i64)],
    "ink_relpos_bias64_f16": [c_void_p, c_i64, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    "ink_proj256_ln_pack": [c_void_p, c_void_p, c_void_p],
    "ink_proj256_ln": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_i64, c_void_p,
                       c_void_p, c_void_p],
    "ink_sam_upscale_pack": [c_void_p, c_void_p, c_void_p],
    "ink_sam_upscale_tail": [c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p],
    "ink_ffn256_pack_bytes": [c_int, c_int, C.POINTER(c_i64)],
    "ink_ffn256_pack": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    "ink_ffn256_fused": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int,
                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "ink_fusion_fold": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_i64, c_int, c_void_p,
                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                        c_void_p, c_void_p, c_void_p],
    "ink_attn_fewkeys": [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int,
                         c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_i64, c_void_p],
    "ink_attn_fewq": [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int,
                      c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_i64, c_void_p],
    "ink_topk_rowmax": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "ink_sine_embed4": [c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "ink_box_refine": [c_void_p, c_i64, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "ink_mask_cleanup_workspace_ints": [c_int, c_int, c_int, c_int, C.POINTER(c_i64)],
    "ink_mask_cleanup": [c_void_p, c_int, c_int, c_int, c_int, c_int, C.c_double, c_void_p, c_void_p, c_void_p,
                         c_void_p, c_void_p],
    "ink_mask_sketch_iou_counts": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ink_refine_sketch_planes": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ink_bitplane_pack": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "ink_refine_depth_samples": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ink_refine_pair_tables": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p],
    "ink_refine_composite": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ink_refine_relabel_clean": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "ink_refine_grow_workspace": [c_int, c_int, C.POINTER(c_i64), C.POINTER(c_i64)],
    "ink_refine_grow": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                        c_int, c_void_p, c_void_p],
    "ink_refine_query_dists": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "ink_refine_finalize": [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "ink_host_sparse_sample": [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p],
    "ink_host_assign_unlabeled": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "ink_depth_patchify": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                           c_int, c_void_p, c_void_p],
    "ink_resize_bilinear_ac_nhwc": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ink_im2col3x3_ex_f16": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "ink_relpos_bias": [c_void_p, c_i64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
}


class InkLayerHipError(RuntimeError):
    pass


def lib_path() -> Path:
    return Path(os.environ.get("INKLAYER_HIP_LIB", _LIB_PATH))


def lib() -> C.CDLL:
    """Load (once) and return the HIP library; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not p.exists():
            raise InkLayerHipError(
                f"{p} not found: build it with `python -m inklayer_amd.build` "
                "(there is no CPU/eager fallback for the InkLayer hot path)")
        # In a process that also runs PyTorch-ROCm, torch's bundled HIP runtime must be the one that is loaded: the
        # library links against libamdhip64 by SONAME, and loading it first would bring in the system ROCm runtime,
        # on which torch's own launches then fail ("HIP launch error" on the first kernel).  Import torch first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(str(p))
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the symbol is missing
            fn.argtypes = argtypes
            fn.restype = C.c_int
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise InkLayerHipError(f"{what} failed: " + {1: "bad argument", 2: "HIP launch error"}.get(rc, str(rc)))
