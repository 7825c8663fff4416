"""ctypes binding of libcds_hip.so (include/cds_hip.h) -- the same entry points a JVM host binds.

Nothing here computes: it only marshals numpy arrays / device pointers across the C ABI.  There is no
CPU fallback; when the shared library is missing the import of the symbols fails loudly.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CDS_HIP_LIBRARY") or os.path.join(_HERE, "libcds_hip.so")   # (override: A/B runs of two builds)

CDS_OK = 0
CDS_ERR_INVALID_ARGUMENT = -1
CDS_ERR_NO_DEVICE = -2
CDS_ERR_HIP = -3
CDS_ERR_OUT_OF_MEMORY = -4
CDS_ERR_UNSUPPORTED = -5
CDS_N_TRANSFORMS = 18


class CdsError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[cds {code}] {message}")
        self.code = code
        self.message = message


class CdsInvalidArgument(CdsError, ValueError):
    """Raised where the reference throws IllegalArgumentException."""


class Rect(ctypes.Structure):
    _fields_ = [("x0", ctypes.c_int32), ("y0", ctypes.c_int32), ("x1", ctypes.c_int32), ("y1", ctypes.c_int32)]


class PixMatchParams(ctypes.Structure):
    _fields_ = [("mask_threshold", ctypes.c_int32), ("data_threshold", ctypes.c_int32),
                ("z_tolerance", ctypes.c_double), ("xy_shift", ctypes.c_int32), ("mirror", ctypes.c_int32)]


class ImageInfo(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("channels", ctypes.c_int32), ("bits", ctypes.c_int32)]


_vp = ctypes.c_void_p
_u8p = ctypes.POINTER(ctypes.c_uint8)
_u16p = ctypes.POINTER(ctypes.c_uint16)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_f32p = ctypes.POINTER(ctypes.c_float)
_rp = ctypes.POINTER(Rect)
_i = ctypes.c_int
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64

# name -> (restype, argtypes): exactly the declarations of include/cds_hip.h (+ cds_hip_debug.h)
SIGNATURES = {
    "cds_last_error": (ctypes.c_char_p, []),
    "cds_version": (ctypes.c_char_p, []),
    "cds_context_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "cds_context_destroy": (None, [_vp]),
    "cds_context_set_stream": (_i, [_vp, _vp]),
    "cds_context_reset_stream": (_i, [_vp]),
    "cds_context_synchronize": (_i, [_vp]),
    "cds_context_trim": (_i, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "cds_context_set_cache_limit": (_i, [_vp, ctypes.c_uint64]),
    "cds_library_bytes": (ctypes.c_size_t, [_i32, _i32, _i64]),
    "cds_library_create": (_i, [_vp, _i32, _i32, _i64, ctypes.POINTER(_vp)]),
    "cds_library_wrap": (_i, [_vp, _i32, _i32, _i64, _vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "cds_library_destroy": (None, [_vp]),
    "cds_library_upload": (_i, [_vp, _i64, _i64, _u8p]),
    "cds_library_download": (_i, [_vp, _i64, _i64, _u8p]),
    "cds_library_upload_variants": (_i, [_vp, _i64, _i64, _u16p, _u8p]),
    "cds_library_upload_gradient_u8": (_i, [_vp, _i64, _i64, _u8p]),
    "cds_library_load_files": (_i, [_vp, _i64, _i64, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p), _i32]),
    "cds_variant_shard_create": (_i, [ctypes.c_char_p, _i32, _i32, ctypes.POINTER(_vp)]),
    "cds_variant_shard_append": (_i, [_vp, _i64, _u8p, _u16p, _u8p, _i32]),
    "cds_variant_shard_append_files": (_i, [_vp, _i64, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p), _i32]),
    "cds_variant_shard_finish": (_i, [_vp]),
    "cds_variant_shard_open": (_i, [ctypes.c_char_p, ctypes.POINTER(_vp)]),
    "cds_variant_shard_info": (_i, [_vp, _i32p, _i32p, _i64p, _i64p]),
    "cds_variant_shard_target_info": (_i, [_vp, _i64, _i64, _u32p, _u32p]),
    "cds_variant_shard_close": (None, [_vp]),
    "cds_library_load_variants": (_i, [_vp, _i64, _vp, _i64p, _i64, _i32]),
    "cds_library_load_variant_images": (_i, [_vp, _i64, _vp, _i64p, _i64, _i32]),
    "cds_library_download_variants": (_i, [_vp, _i64, _i64, _u16p, _u8p]),
    "cds_library_generate_synthetic": (_i, [_vp, ctypes.c_uint64, _i64, _i64, _i64, _i32, _i32]),
    "cds_library_generate_synthetic_list": (_i, [_vp, ctypes.c_uint64, _i64p, _i64, _i64, _i32, _i32]),
    "cds_maskset_create": (_i, [_vp, _i32, _u8p, _i32, _i32, ctypes.POINTER(PixMatchParams), _rp, _i32,
                                ctypes.POINTER(_vp)]),
    "cds_maskset_destroy": (None, [_vp]),
    "cds_maskset_query_sizes": (_i, [_vp, _i32p]),
    "cds_search": (_i, [_vp, _vp, _i64, _i64, _i32p, _u8p]),
    "cds_search_device": (_i, [_vp, _vp, _i64, _i64, _vp]),
    "cds_search_counts": (_i, [_vp, _vp, _i64, _i64, _i32p]),
    "cds_score_pair": (_i, [_vp, _i32, _u8p, _i32, _i32, _i32p, _u8p]),
    "cds_pixel_match_normalized": (ctypes.c_float, [_i32, _i32]),
    "cds_is_match": (_i, [_i32, _i32, ctypes.c_double]),
    "cds_maskset_last_kernel_ms": (_i, [_vp, ctypes.POINTER(ctypes.c_double), _i32p]),
    "cds_sparse_library_create": (_i, [_vp, _i32, _i32, _i64, _i64, ctypes.POINTER(_vp)]),
    "cds_sparse_library_destroy": (None, [_vp]),
    "cds_sparse_library_set_growable": (_i, [_vp, _i32]),
    "cds_sparse_library_reserve": (_i, [_vp, _i64, _i64]),
    "cds_sparse_library_capacity": (_i, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i32)]),
    "cds_sparse_library_append": (_i, [_vp, _vp, _i64, _i64]),
    "cds_sparse_library_append_host": (_i, [_vp, _i64, _u8p]),
    "cds_sparse_library_layout": (_i, [_vp, _i32p, _i32p]),
    "cds_sparse_library_set_density_limit": (_i, [_vp, ctypes.c_double]),
    "cds_sparse_library_set_min_threshold": (_i, [_vp, ctypes.c_int32]),
    "cds_sparse_library_min_threshold": (_i, [_vp, ctypes.POINTER(ctypes.c_int32)]),
    "cds_sparse_library_size": (_i64, [_vp]),
    "cds_sparse_library_pixels": (_i64, [_vp]),
    "cds_sparse_library_clear": (_i, [_vp]),
    "cds_sparse_library_image_records": (_i, [_vp, _i64, _u32p, _i64, _i64p, _u32p]),
    "cds_sparse_library_image_offsets": (_i, [_vp, _i64, _i64, ctypes.POINTER(ctypes.c_uint64)]),
    "cds_sparse_library_export": (_i, [_vp, _i64, _i64, ctypes.POINTER(ctypes.c_uint64), _u32p, _u32p, _i64]),
    "cds_sparse_library_import": (_i, [_vp, _i64, ctypes.POINTER(ctypes.c_uint64), _u32p, _u32p, _i32, _i32, _i32]),
    "cds_image_probe": (_i, [_u8p, ctypes.c_size_t, ctypes.POINTER(ImageInfo)]),
    "cds_image_decode": (_i, [_u8p, ctypes.c_size_t, _vp, ctypes.c_size_t]),
    "cds_sparse_library_append_files": (_i, [_vp, ctypes.POINTER(ctypes.c_char_p), _i64, _i32]),
    "cds_search_sparse": (_i, [_vp, _vp, _i64, _i64, _i32p, _u8p]),
    "cds_search_sparse_device": (_i, [_vp, _vp, _i64, _i64, _vp]),
    "cds_search_sparse_counts": (_i, [_vp, _vp, _i64, _i64, _i32p]),
    "cds_search_sparse_match_list": (_i, [_vp, _vp, _i64, _i64, ctypes.c_double, ctypes.POINTER(_vp), _i64p]),
    "cds_match_list_size": (_i64, [_vp]),
    "cds_match_list_fetch": (_i, [_vp, _i64, _i64, _vp]),
    "cds_match_list_fetch_columns": (_i, [_vp, _i64, _i64, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "cds_match_list_destroy": (None, [_vp]),
    "cds_search_sparse_topk": (_i, [_vp, _vp, _i64, _i64, _i32, _i32p, _i32p]),
    "cds_scores_topk": (_i, [_vp, _vp, _i64, _i32, _i32p, _i32p]),
    "cds_search_sparse_matches": (_i, [_vp, _vp, _i64, _i64, ctypes.c_double, _vp, _i64, _i64p]),
    "cds_comm_create": (_i, [ctypes.POINTER(_vp), _i32, ctypes.POINTER(_vp)]),
    "cds_comm_destroy": (None, [_vp]),
    "cds_comm_size": (_i32, [_vp]),
    "cds_comm_uses_rccl": (_i32, [_vp]),
    "cds_search_sparse_sharded": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i64p, _i32p, _u8p]),
    "cds_search_sparse_sharded_topk": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int32, _i64p, _i32p, _i32p]),
    "cds_search_sparse_sharded_matches": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_double, _i64p, _vp, _i64, _i64p]),
    "cds_shapeset_last_call_info": (_i, [_vp, ctypes.POINTER(ctypes.c_int32), _i64p, ctypes.POINTER(ctypes.c_double)]),
    "cds_maskset_size": (ctypes.c_int32, [_vp]),
    "cds_shapeset_size": (ctypes.c_int32, [_vp]),
    "cds_library_dims": (_i, [_vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), _i64p]),
    "cds_sparse_library_dims": (_i, [_vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "cds_gradscore_sharded": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i64p, _i64, _i32p, _i64p, _i64p, _i64p, _i64p, _u8p]),
    "cds_shapeset_create": (_i, [_vp, _i32, _u8p, _i32, _i32, _i32, _i32, _u8p, _rp, _i32, ctypes.POINTER(_vp)]),
    "cds_shapeset_create_bordered": (_i, [_vp, _i32, _u8p, _i32, _i32, _i32, _i32, _i32, _u8p, _rp, _i32, ctypes.POINTER(_vp)]),
    "cds_shapeset_destroy": (None, [_vp]),
    "cds_shapeset_mask_stats": (_i, [_vp, _i64p, _i64p]),
    "cds_gradscore": (_i, [_vp, _vp, _i64, _i32p, _i64p, _i64p, _i64p, _u8p]),
    "cds_shape_score_pair": (_i, [_vp, _i32, _u8p, _u16p, _u8p, _i32, _i32, _i64p, _i64p, _u8p]),
    "cds_shape_score_value": (_i64, [_i64, _i64]),
    "cds_normalized_score": (ctypes.c_double, [_i32, _i64, _i64, _i64]),
    "cds_shapeset_last_kernel_ms": (_i, [_vp, ctypes.POINTER(ctypes.c_double), _i32p]),
    "cds_make_zgap": (_i, [_vp, _u8p, _i32, _i32, _i32, ctypes.c_double, _rp, _i32, _u8p]),
    "cds_library_make_zgap_variants": (_i, [_vp, _i64, _i64, _i32, ctypes.c_double, _rp, _i32]),
    # include/cds_hip_debug.h
    "cds_debug_intervals": (_i, [_f32p, _i32, _i32, _i32, ctypes.c_double, _u32p]),
    "cds_debug_band_bits": (_i, [ctypes.c_uint32, ctypes.c_uint32, _u32p]),
    "cds_debug_band_of": (_i32, [_f32p, _i32, _i32, _i32]),
    "cds_debug_hue_code": (ctypes.c_uint32, [_f32p, _i32, _i32, _i32, _i32]),
    "cds_debug_device_rcp": (_i, [_vp, _f32p]),
    "cds_debug_valu_issue_ns": (_i, [_vp, ctypes.POINTER(ctypes.c_double)]),
    "cds_debug_sparse_load_only": (_i, [_vp, ctypes.c_int32]),
    "cds_debug_sparse_load_only_state": (ctypes.c_int32, [_vp]),
    "cds_debug_maskset_scan_kernel": (_i, [_vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)]),
    "cds_debug_sparse_stream_read": (_i, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "cds_maskset_sparse_groups": (_i, [_vp, _i32p, _i32p, _i32p]),
    "cds_maskset_sparse_layout": (_i, [_vp, _i32, _i32p]),
}

_lib = None


def load():
    """Load libcds_hip.so and bind every declared symbol.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc == CDS_OK:
        return
    msg = load().cds_last_error().decode("utf-8", "replace")
    if rc == CDS_ERR_INVALID_ARGUMENT:
        raise CdsInvalidArgument(rc, msg)
    raise CdsError(rc, msg)


def rect_array(rects):
    rects = list(rects or [])
    arr = (Rect * max(1, len(rects)))()
    for k, r in enumerate(rects):
        arr[k] = Rect(*[int(v) for v in r])
    return arr, len(rects)


def as_u8p(a):
    return a.ctypes.data_as(_u8p)


def as_u16p(a):
    return a.ctypes.data_as(_u16p)


def as_i32p(a):
    return a.ctypes.data_as(_i32p)


def as_i64p(a):
    return a.ctypes.data_as(_i64p)


def rgb_images(a, n=None):
    """Coerce to contiguous uint8[n,H,W,3]; a single image gets a leading axis."""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 3:
        a = a[None]
    if a.ndim != 4 or a.shape[3] != 3:
        raise CdsInvalidArgument(CDS_ERR_INVALID_ARGUMENT, f"expected uint8[n,H,W,3] RGB images, got {a.shape}")
    if n is not None and a.shape[0] != n:
        raise CdsInvalidArgument(CDS_ERR_INVALID_ARGUMENT, f"expected {n} images, got {a.shape[0]}")
    return a
