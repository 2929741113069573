"""ctypes binding of libp2w_gfx950.so (C ABI: include/p2w.h).

The library is the product path: there is no CPU or eager-PyTorch fallback, so a
missing library is a hard error naming the build command.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libp2w_gfx950.so")

_vp, _i32, _f32, _sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t


class Epilogue(C.Structure):
    _fields_ = [("bias", _vp), ("sc0", _vp), ("sh0", _vp), ("sc1", _vp), ("sh1", _vp), ("residual", _vp),
                ("ldr", _i32), ("relu0", _i32), ("relu1", _i32), ("relu2", _i32), ("relu_final", _i32), ("range", _vp),
                ("interp", _vp), ("interp_rows", _i32)]


# name -> (restype, argtypes); must list every symbol declared in include/p2w.h
SEARCH_X_INDEX_IN_W, SEARCH_Q_ROW_IN_W, SEARCH_BOX = 1, 2, 4   # include/p2w.h P2W_SEARCH_*
PREC_F16X3, PREC_F16, PREC_BF16 = 0, 1, 2                      # include/p2w.h P2W_PREC_*
PREC_OF = {"f16x3": PREC_F16X3, "fp16": PREC_F16, "bf16": PREC_BF16}
GEMM_TILE_128, GEMM_TILE_256, GEMM_GENERIC_EPI, GEMM_ORDER_ROWS, GEMM_ORDER_COLS, GEMM_RESIDUAL_H = 1, 2, 4, 8, 16, 32   # P2W_GEMM_*
GEMM_STREAMK, GEMM_NO_STREAMK = 64, 128
SA_ITEM_256, SA_ITEM_128, SA_PACK8 = 1, 2, 4                                                       # P2W_SA_*

SIGNATURES = {
    "p2w_version": (_i32, []),
    "p2w_strerror": (C.c_char_p, [_i32]),
    "p2w_pack_xyzr": (_i32, [_vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "p2w_voxel_sample_ws_bytes": (_sz, [_i32]),
    "p2w_voxel_sample": (_i32, [_vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "p2w_voxel_sample_table_ws_bytes": (_sz, [_i32, C.c_int64]),
    "p2w_voxel_sample_table": (_i32, [_vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64,
                                      _vp, _sz, _vp]),
    "p2w_voxel_sample_table_prepare": (_i32, [_vp, _sz, _vp]),
    "p2w_voxel_sample_table_prepared": (_i32, [_vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64,
                                               _vp, _sz, _vp]),
    "p2w_knn_grid_indexed": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "p2w_ball_query_grid_indexed": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.c_double, _i32, _vp, _vp, _i32, _vp]),
    "p2w_knn_hint2": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "p2w_knn_grid": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "p2w_ball_query_grid": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.c_double, _i32, _vp, _vp, _i32, _vp]),
    "p2w_index_records": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "p2w_voxel_grid": (_i32, [_vp, _vp, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "p2w_consecutive_cluster": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "p2w_level_gather": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "p2w_ball_query": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, C.c_double, _i32, _vp, _vp, _vp, _i32, _vp]),
    "p2w_knn": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "p2w_morton_order_ws_bytes": (_sz, [_i32]),
    "p2w_morton_order": (_i32, [_vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "p2w_cells_nd": (_i32, [_vp, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "p2w_sort_pairs_u64_ws_bytes": (_sz, [_i32]),
    "p2w_sort_pairs_u64": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "p2w_key_runs_ws_bytes": (_sz, [_i32]),
    "p2w_key_runs": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "p2w_knn_refine_f64": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "p2w_cell_starts_ws_bytes": (_sz, [C.c_int64]),
    "p2w_cell_starts": (_i32, [_vp, _i32, C.c_int64, _vp, _vp, _sz, _vp]),
    "p2w_vote": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _f32, _vp, _vp, _vp]),
    "p2w_tile_bbox": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "p2w_tile_bbox_count": (_i32, [_i32, _i32]),
    "p2w_stem": (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "p2w_packed_dims": (None, [_i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "p2w_gemm": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, C.POINTER(Epilogue), _vp, _i32, _vp]),
    "p2w_sa_conv": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp,
                           _vp, _i32, _vp]),
    "p2w_packed_dims_h": (_i32, [_i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "p2w_gemm_h2": (_i32, [_i32, _vp, _i32, _vp, _f32, _i32, _i32, _i32, C.POINTER(Epilogue), _vp, _i32, _vp, _i32, _i32, _vp]),
    "p2w_gemm_h2_sk_ws_bytes": (_sz, []),
    "p2w_gemm_h2_sk": (_i32, [_i32, _vp, _i32, _vp, _f32, _i32, _i32, _i32, C.POINTER(Epilogue), _vp, _i32, _vp, _i32, _vp, _sz, _i32, _vp]),
    "p2w_gemm_h2_rowdot_ws_bytes": (_sz, [_i32, _i32]),
    "p2w_gemm_h2_rowdot": (_i32, [_i32, _vp, _i32, _vp, _f32, _i32, _i32, _i32, C.POINTER(Epilogue), _vp, _f32, _vp, _vp, _sz, _i32, _vp]),
    "p2w_sa_conv_h_ws_bytes": (_sz, [_i32, _i32]),
    "p2w_sa_conv_h": (_i32, [_i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _f32, _i32, _i32, _vp, _vp,
                             _vp, _vp, _i32, _vp, _i32, _vp, _sz, _i32, _vp]),
    "p2w_stem_h2": (_i32, [_i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "p2w_stem_h2_indexed": (_i32, [_i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "p2w_sa_conv_h_rows": (_i32, [_i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _f32, _i32, _i32, _vp, _vp,
                                  _vp, _vp, _i32, _vp, _i32, _vp, _sz, _i32, _vp, _vp, _vp]),
    "p2w_interp_concat_h2": (_i32, [_i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _i32, _vp]),
    "p2w_interp_weights": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "p2w_concat_xyz_h2": (_i32, [_i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp]),
    "p2w_interp_concat": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _i32, _vp]),
    "p2w_concat_xyz": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp]),
    "p2w_segment_max": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "p2w_rowdot": (_i32, [_vp, _i32, _i32, _vp, _f32, _i32, _vp, _vp]),
    "p2w_fill_batch_nbr": (_i32, [_vp, _i32, _vp, _vp, _vp]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raise loudly when the library is not built."""
    global _lib
    if _lib is None:
        from . import build as _build
        if _build._stale():  # missing, or built from other sources than the ones in the tree
            try:
                _build.build()
            except Exception as e:  # no hipcc, compile error, read-only tree ...
                raise RuntimeError(
                    f"{LIB_PATH} is missing or stale and could not be rebuilt ({e}). The HIP extension is the "
                    "only compute path (no CPU fallback). Build it with `python -m pointstowood_amd.build`.") from e
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError = symbol missing from the build
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = lib().p2w_strerror(code).decode()
        raise RuntimeError(f"{what or 'p2w'} failed: {msg} (code {code})")


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """Raw handle of torch's current HIP stream.  Through torch._C directly when it is there: ``torch.cuda.current_stream()``
    builds a Stream object behind several device-index lookups (8 us per call, 0.45 ms of host time per forward - a third of it;
    small batches are paced by the host)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pointstowood_amd operators need tensors on an MI355X (cuda) device; "
                               "there is no CPU fallback")


def packed_dims(n: int, k: int, prec: int | None = None):
    """(N_pad, K_pad) of a packed weight: fp32 / f16x3 pad K to 32, the single-plane H precisions to 64."""
    a, b = _i32(), _i32()
    if prec is None:
        lib().p2w_packed_dims(n, k, C.byref(a), C.byref(b))
    else:
        check(lib().p2w_packed_dims_h(prec, n, k, C.byref(a), C.byref(b)), "p2w_packed_dims_h")
    return a.value, b.value
