"""ctypes binding of lib/libsmml_hip.so (C-ABI declared in include/smml.h).

The product path has no CPU or eager fallback: if the library is missing, or a tensor is not a
contiguous fp32 CUDA(HIP) tensor, the call fails loudly."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# SMML_LIB: measurement hook - load another build of the same C-ABI (tests/build_variants.py writes them to lib/variants/)
LIB_PATH = os.environ.get("SMML_LIB") or os.path.join(_PKG, "lib", "libsmml_hip.so")

_f = C.c_void_p          # device pointers travel as void*


class DeformOpts(C.Structure):
    """SmmlDeformOpts of include/smml.h: optional behaviour of ONE fused deformable-attention launch, passed with the call."""
    _fields_ = [("seed_offset", C.c_void_p), ("raw_distance", C.c_int), ("mask_table", C.c_void_p), ("mask_table_pmax", C.c_float),
                ("export_masks", C.c_void_p), ("region_lds_cap", C.c_int)]


_o = C.POINTER(DeformOpts)
_i, _ll, _fl, _sz = C.c_int, C.c_longlong, C.c_float, C.c_size_t

ABI_VERSION = 2          # smml_abi_version() of the library this binding is written against

# name -> (restype, argtypes); mirrors include/smml.h one to one
SIGNATURES = {
    "smml_last_error": (C.c_char_p, []),
    "smml_abi_version": (_i, []),
    "smml_device_check": (_i, [_i]),
    "smml_gemm_f32": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _ll, _ll, _ll, _ll, _ll, _ll, _i, _i, _ll, _ll, _ll, _ll,
                           _ll, _ll, _ll, _ll, _i, _i, _ll, _i, _i, _i, _fl, _fl, _f]),
    "smml_gemm_force_generic": (None, [_i]),
    "smml_gemm_set_mode": (None, [_i]),
    "smml_gemm_get_mode": (_i, []),
    "smml_gemm_set_small_tile": (None, [_i]),
    "smml_gemm_b16": (_i, [_f, _f, _f, _f, _i, _i, _i, _ll, _ll, _ll, _i, _i, _i, _f]),
    "smml_gemm_b16_set_tile": (None, [_i]),
    "smml_gemm_b16_set_slice_major": (None, [_i]),
    "smml_gemm_b16_batched": (_i, [_f, _f, _f, _f, _i, _i, _i, _ll, _ll, _ll, _i, _i, _i, _i, _ll, _ll, _ll, _f]),
    "smml_attn16_fwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "smml_attn16_fwd_f32": (_i, [_f, _f, _f, _f, _f, _f, _sz, _i, _i, _i, _i, _fl, _i, _i, _i, _f]),
    "smml_attn16_set_fewkeys": (None, [_i]),
    "smml_attn16_set_query_blocks": (None, [_i]),
    "smml_attn16_fwd_b16": (_i, [_f, _f, _f, _f, _f, _f, _f, _sz, _i, _i, _i, _i, _fl, _i, _ll, _ll, _ll, _ll, _ll, _ll, _f]),
    "smml_attn16_bwd_b16": (_i, [_f] * 11 + [_sz, _i, _i, _i, _i, _fl, _i] + [_ll] * 9 + [_i, _f]),
    "smml_attn16_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "smml_attn16_bwd_f32": (_i, [_f] * 10 + [_f, _sz, _i, _i, _i, _i, _fl, _i, _i, _f]),
    "smml_layernorm_fwd_f32": (_i, [_f, _f, _f, _f, _f, _f, _ll, _i, _fl, _f]),
    "smml_layernorm_bwd_f32": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _ll, _i, _ll, _fl, _i, _f]),
    "smml_colsum_f32": (_i, [_f, _f, _i, _ll, _i, _fl, _f]),
    "smml_orth_loss_f32": (_i, [_f] * 10 + [_i, _i, _fl, _f]),
    "smml_batchloss_tail_f32": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _fl, _f]),
    "smml_relu_bwd_f32": (_i, [_f, _f, _f, _ll, _f]),
    "smml_offsets_out_len": (_i, [_i, _i, _i]),
    "smml_offsets_fwd_f32": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _f]),
    "smml_offsets_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "smml_offsets_bwd_f32": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _i, _f]),
    "smml_bilinear_sample_fwd_f32": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f]),
    "smml_bilinear_sample_bwd_f32": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f]),
    "smml_bilinear_corners_f32": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _f]),
    "smml_deform_attn_nst": (_i, [_i]),
    "smml_deform_attn_fwd_f32": (_i, [_f] * 15 + [_i, _i, _i, _i, _i, _i, _fl, _fl, C.c_ulonglong, _f, _f, _f] + [_o]),
    "smml_deform_attn_dropout_mask_f32": (_i, [_f, _i, _i, _i, _i, _fl, C.c_ulonglong, _f] + [_o]),
    "smml_deform_attn_relu1_masks": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f] + [_o]),
    "smml_deform_attn_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "smml_deform_attn_bwd_f32": (_i, [_f] * 27 + [_f, _sz, _i, _i, _i, _i, _i, _i, _fl, _fl, C.c_ulonglong, _f, _f, _f] + [_o]),
    "smml_cpb_regions_bytes": (_sz, []),
    "smml_cpb_regions_build": (_i, [_f] * 6 + [_fl, _f, _sz, _f]),
    "smml_deform_attn_region_fwd_f32": (_i, [_f] * 16 + [_i, _i, _i, _i, _fl, _fl, C.c_ulonglong, _f, _f, _f] + [_o]),
    "smml_deform_attn_region_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "smml_deform_attn_region_bwd_f32": (_i, [_f] * 28 + [_f, _sz, _i, _i, _i, _i, _fl, _fl, C.c_ulonglong, _f, _f, _f] + [_o]),
    "smml_deform_attn16_region_fwd": (_i, [_f] * 16 + [_i, _i, _i, _i, _fl, _fl, C.c_ulonglong, _i, _f, _f, _f] + [_o]),
    "smml_deform_attn16_region_bwd": (_i, [_f] * 28 + [_f, _sz, _i, _i, _i, _i, _fl, _fl, C.c_ulonglong, _i, _f, _f, _f] + [_o]),
    "smml_deform_attn16_fwd": (_i, [_f] * 15 + [_i, _i, _i, _i, _i, _i, _fl, _fl, C.c_ulonglong, _i, _f, _f, _f] + [_o]),
    "smml_deform_attn16_bwd": (_i, [_f] * 27 + [_f, _sz, _i, _i, _i, _i, _i, _i, _fl, _fl, C.c_ulonglong, _i, _f, _f, _f] + [_o]),
    "smml_cpb_mask_table_cells": (_i, [_i]),
    "smml_cpb_mask_table": (_i, [_f, _f, _f, _f, _f, _i, _fl, _f]),
    "smml_deform_attn_table_points": (_i, [_i]),
    "smml_deform_attn_table_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "smml_deform_attn_table_fwd": (_i, [_f] * 9 + [_i, _i, _i, _i, _i, _i, _i, _fl, _fl, _fl, C.c_ulonglong, _i, _f, _f, _f] + [_o]),
    "smml_deform_attn_table_bwd": (_i, [_f] * 16 + [_f, _sz, _i, _i, _i, _i, _i, _i, _i, _fl, _i, _i, _fl, _fl, C.c_ulonglong, _i, _f, _f, _f] + [_o]),
    "smml_softmax_fwd_f32": (_i, [_f, _f, _ll, _i, _f]),
    "smml_softmax_bwd_f32": (_i, [_f, _f, _f, _ll, _i, _f]),
    "smml_tile_rows_f32": (_i, [_f, _f, _ll, _i, _i, _fl, _f]),
    "smml_segment_mean_b16": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f]),
    "smml_segment_mean_bwd_add_b16": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f]),
    "smml_resconv_b16": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _i, _f]),
    "smml_resconv_wgrad_b16": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _f]),
    "smml_newton_schulz_set_fast": (None, [_i]),
    "smml_newton_schulz_saved_floats": (_sz, [_i, _i, _i, _i]),
    "smml_newton_schulz_scratch_floats": (_sz, [_i, _i, _i, _i]),
    "smml_newton_schulz_fwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _f]),
    "smml_newton_schulz_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _f]),
    "smml_resconv_fwd_f32": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f]),
    "smml_resconv_bwd_f32": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f]),
    "smml_dwconv7_fwd_f32": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _f]),
    "smml_dwconv7_bwd_weight_f32": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _f]),
    "smml_grad_modulate_f32": (_i, [_f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _f]),
    "smml_grad_modulate_survival_f32": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _f]),
    "smml_fixdim_indices": (_i, [_f, _ll, _ll, _f]),
    "smml_fixdim_gather_bf16": (_i, [_f, _ll, _f, _i, _ll, _i, _f]),
    "smml_event_create": (C.c_void_p, []),
    "smml_event_destroy": (_i, [_f]),
    "smml_event_record": (_i, [_f, _f]),
    "smml_event_elapsed_ms": (_i, [_f, _f, C.POINTER(C.c_float)]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Loads the shared library once; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU or eager fallback for this path")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.smml_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} implements C-ABI version {handle.smml_abi_version()}, this package binds version {ABI_VERSION}: rebuild it")
        if os.environ.get("SMML_GEMM_MODE"):       # measurement switch: 1 = fp32-MFMA GEMM only, 2 = split-bf16 wherever it applies
            handle.smml_gemm_set_mode(int(os.environ["SMML_GEMM_MODE"]))
        if os.environ.get("SMML_GEMM_SMALL_TILE"):  # measurement switch: 1 = never the 64-row tile, 2 = wherever it applies
            handle.smml_gemm_set_small_tile(int(os.environ["SMML_GEMM_SMALL_TILE"]))
        _lib = handle
    return _lib


def deform_opts(seed_offset: Optional[torch.Tensor] = None, log_distance: bool = True, mask_table: Optional[torch.Tensor] = None,
                mask_table_pmax: float = 0.0, export_masks: Optional[torch.Tensor] = None, region_lds_cap: int = 0):
    """The `opts` argument of a fused deformable-attention entry point (byref of a DeformOpts), or None when every field is at its default.
    The struct is read during the call only; the tensors it points at must outlive the launch (the callers keep them)."""
    if seed_offset is None and log_distance and mask_table is None and export_masks is None and not region_lds_cap:
        return None
    if seed_offset is not None and (seed_offset.dtype != torch.int64 or seed_offset.numel() != 1):
        raise RuntimeError("dropout_seed_offset must be a device int64 tensor with one element")
    o = DeformOpts(ptr(seed_offset), 0 if log_distance else 1, ptr(mask_table), float(mask_table_pmax), ptr(export_masks), int(region_lds_cap))
    return C.byref(o)


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().smml_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"smml call failed ({rc}){': ' + what if what else ''}: {msg}")


def ptr(t: Optional[torch.Tensor]):
    """Raw device pointer of a contiguous fp32/int32/uint8 HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("smml kernels need tensors resident in GPU HBM (got a CPU tensor); "
                           "this path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("smml kernels need contiguous tensors")
    if t.device.index != torch.cuda.current_device():
        # the C entry points launch on the current HIP device and stream() hands them that device's stream: a tensor of another
        # device would be dereferenced by a kernel running elsewhere (one process per GPU: torch.cuda.set_device(LOCAL_RANK))
        raise RuntimeError(f"smml kernels launch on the current device (cuda:{torch.cuda.current_device()}) but got a tensor on "
                           f"{t.device}; call torch.cuda.set_device first (one process per GPU)")
    return C.c_void_p(t.data_ptr())


def fptr(t: Optional[torch.Tensor]):
    if t is not None and t.dtype != torch.float32:
        raise RuntimeError(f"smml fp32 kernel got dtype {t.dtype}")
    return ptr(t)


def stream(device=None):
    """The current HIP stream of `device` (default: the current device) as a void*."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
