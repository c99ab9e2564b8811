"""ctypes binding of libsparsify_hip.so (the C ABI declared in include/sparsify_hip.h).

There is no CPU fallback: if the library is missing, or an entry point reports an error,
the caller gets an exception.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p
from pathlib import Path

LIB_NAME = "libsparsify_hip.so"
LIB_PATH = Path(__file__).resolve().parent / LIB_NAME
ABI_VERSION = 3

F32, BF16 = 0, 1


SPHIP_EUNSUPPORTED = 2        # include/sparsify_hip.h


class HipLibraryError(RuntimeError):
    """An entry point of libsparsify_hip.so returned a non-zero status."""


_P, _I, _L, _F, _Z, _D = c_void_p, c_int, c_int64, c_float, c_size_t, c_double

# name -> (restype, argtypes); mirrors include/sparsify_hip.h one to one
SIGNATURES: dict[str, tuple] = {
    "sphip_abi_version": (_I, []),
    "sphip_last_error": (c_char_p, []),
    "sphip_center_cast": (_I, [_P, _I, _P, _P, _P, _L, _L, _P]),
    "sphip_cast_f32_to_bf16": (_I, [_P, _P, _L, _P]),
    "sphip_encoder_preacts": (_I, [_P, _P, _P, _P, _P, _I, _L, _L, _L, _I, _P]),
    "sphip_encoder_preacts_uses_tile256": (_I, [_I, _L, _L, _L]),
    "sphip_encoder_preacts_block": (_I, [_P, _P, _P, _P, _P, _I, _L, _L, _L, _L, _L, _I, _P]),
    "sphip_topk": (_I, [_P, _I, _P, _P, _L, _L, _L, _P, _P, _P]),
    "sphip_groupmax": (_I, [_P, _I, _L, _L, _L, _P, _P, _P]),
    "sphip_exclude_selected": (_I, [_P, _P, _L, _P, _P, _L, _I, _L, _P, _P]),
    "sphip_lazy_workspace_bytes": (_Z, [_L, _L]),
    "sphip_row_min": (_I, [_P, _L, _L, _P, _P]),
    "sphip_encoder_preacts_emit": (_I, [_P, _P, _P, _L, _L, _L, _L, _P, _L, _I, _P, _L, _P]),
    "sphip_topk_candidates": (_I, [_P, _P, _P, _L, _L, _L, _P, _L, _P, _P, _P]),
    "sphip_encoder_preacts_if": (_I, [_P, _P, _P, _P, _P, _P, _L, _L, _L, _I, _P]),
    "sphip_topk_if": (_I, [_P, _P, _I, _P, _P, _L, _L, _L, _P, _P, _P]),
    "sphip_gather_rows_weighted": (_I, [_P, _P, _I, _P, _P, _P, _L, _L, _L, _L, _P]),
    "sphip_gather_l2_workspace_bytes": (_Z, [_L, _L, _L, _L]),
    "sphip_gather_rows_weighted_l2": (_I, [_P, _P, _I, _P, _P, _P, _P, _L, _L, _L, _L, _P, _Z, _P]),
    "sphip_slice_major_f32": (_I, [_P, _P, _L, _L, _P]),
    "sphip_segments_workspace_bytes": (_Z, [_L, _L, _L]),
    "sphip_build_latent_segments": (_I, [_P, _L, _L, _L, _P, _P, _P, _Z, _P]),
    "sphip_weight_grad_workspace_bytes": (_Z, [_L, _L, _L]),
    "sphip_segmented_weight_grad": (_I, [_P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _L, _L, _L, _L, _P, _Z, _P]),
    "sphip_weight_grad_accumulates": (_I, [_L, _L, _L, _L, _I, _I]),
    "sphip_segmented_weight_grad_acc": (_I, [_P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _L, _L, _L, _L, _P, _Z, _I, _P]),
    "sphip_reduce_workspace_bytes": (_Z, [_L, _L]),
    "sphip_total_variance": (_I, [_P, _I, _L, _L, _P, _P, _Z, _P]),
    "sphip_total_variance_static": (_I, [_P, _I, _L, _L, _P, _P]),
    "sphip_residual_sumsq": (_I, [_P, _I, _P, _P, _P, _L, _L, _P, _Z, _P]),
    "sphip_scale_by_ratio": (_I, [_P, _P, _L, _F, _P, _P, _P]),
    "sphip_colsum": (_I, [_P, _I, _L, _L, _F, _P, _P, _Z, _P]),
    "sphip_vec_mat": (_I, [_P, _P, _L, _L, _F, _P, _P, _Z, _P]),
    "sphip_unit_norm_rows": (_I, [_P, _L, _L, _P]),
    "sphip_remove_parallel_grad": (_I, [_P, _P, _L, _L, _P]),
    "sphip_signsgd_step": (_I, [_P, _P, _L, _F, _P, _P]),
    "sphip_signsgd_step_rows": (_I, [_P, _P, _L, _L, _F, _P, _P]),
    "sphip_decoder_project_signsgd": (_I, [_P, _P, _L, _L, _F, _I, _P, _P]),
    "sphip_schedulefree_signsgd_step": (_I, [_P, _P, _P, _L, _F, _F, _F, _P, _P]),
    "sphip_decoder_schedulefree_signsgd_step": (_I, [_P, _P, _P, _L, _L, _F, _F, _F, _I, _P, _P]),
    "sphip_schedulefree_signsgd_step_rows": (_I, [_P, _P, _P, _L, _L, _F, _F, _F, _P, _P]),
    "sphip_dead_compact": (_I, [_P, _L, _P, _P, _L, _P]),
    "sphip_dead_scatter_acts": (_I, [_P, _P, _P, _L, _L, _L, _L, _L, _L, _P, _P, _P]),
    "sphip_split_bf16_transposed": (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _P]),
    "sphip_gemm_bf16_f32acc": (_I, [_P, _P, _P, _L, _L, _L, _L, _I, _P]),
    "sphip_scatter_rows": (_I, [_P, _P, _L, _L, _L, _P, _P]),
    "sphip_scatter_add_rows": (_I, [_P, _P, _L, _L, _L, _P, _P]),
    "sphip_broadcast_rows": (_I, [_P, _L, _L, _P, _P]),
    "sphip_gather_rows_bf16": (_I, [_P, _P, _L, _L, _L, _P, _P]),
    "sphip_schedulefree_apply_signs": (_I, [_P, _P, _P, _L, _F, _F, _F, _P, _P]),
    "sphip_decoder_schedulefree_apply_signs": (_I, [_P, _P, _P, _L, _L, _F, _F, _F, _I, _P, _L, _L, _P]),
    "sphip_segmented_weight_grad_signsgd": (_I, [_P, _P, _P, _I, _P, _I, _P, _L, _L, _L, _L, _P, _F, _P, _P, _P, _P, _Z, _P]),
    "sphip_segmented_weight_grad_schedulefree": (_I, [_P, _P, _P, _I, _P, _I, _P, _L, _L, _L, _L, _P, _P, _F, _F, _F, _P, _P, _P, _P, _Z, _P]),
    "sphip_adam_step": (_I, [_P, _P, _P, _P, _L, _D, _D, _D, _D, _L, _P, _P]),
    "sphip_segment_coeff_sums": (_I, [_P, _P, _P, _I, _L, _P, _P]),
    "sphip_sign_i8": (_I, [_P, _L, _P, _P]),
    "sphip_decoder_project_sign_i8": (_I, [_P, _P, _L, _L, _P, _P]),
    "sphip_apply_signs": (_I, [_P, _P, _L, _F, _P, _P]),
    "sphip_decoder_apply_signs": (_I, [_P, _P, _L, _L, _F, _I, _P, _L, _L, _P]),
    "sphip_mark_fired": (_I, [_P, _L, _L, _P, _P]),
    "sphip_dead_counter_step": (_I, [_P, _P, _L, _L, _P]),
    "sphip_dead_mask": (_I, [_P, _L, _L, _P, _P, _P]),
    "sphip_gemm_bf16_axpy": (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _L, _F, _F, _P]),
    "sphip_muon_workspace_bytes": (_Z, []),
    "sphip_muon_momentum": (_I, [_P, _P, _L, _F, _I, _P, _P, _P, _Z, _P]),
    "sphip_muon_denominator": (_I, [_P, _P]),
    "sphip_muon_normalise": (_I, [_P, _L, _P, _P]),
    "sphip_transpose": (_I, [_P, _P, _L, _L, _L, _I, _P]),
    "sphip_add": (_I, [_P, _P, _I, _P, _L, _P]),
    "sphip_muon_apply": (_I, [_P, _P, _L, _F, _F, _P, _P]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load the library once; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("SPARSIFY_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise ImportError(
            f"{path} not found: the HIP extension is not built. Run "
            "`make -C sparsify_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "sparsify_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = lib.sphip_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {got}, expected {ABI_VERSION}; rebuild the extension")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().sphip_last_error().decode(errors="replace")
        err = HipLibraryError(f"{what} failed (status {status}): {msg}")
        err.status = status
        raise err
