"""ctypes front-end of the CPU oracle (oracle/cds_oracle.c, oracle/cds_synth_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by anything under colormipsearch_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcds_oracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("cds_oracle.c", "cds_synth_oracle.c", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libcds_oracle.so"])
    return _SO


class Rect(ctypes.Structure):
    _fields_ = [("x0", ctypes.c_int), ("y0", ctypes.c_int), ("x1", ctypes.c_int), ("y1", ctypes.c_int)]


def _rects(rects):
    rects = list(rects or [])
    arr = (Rect * max(1, len(rects)))()
    for i, r in enumerate(rects):
        arr[i] = Rect(*[int(v) for v in r])
    return arr, len(rects)


def standard_label_rects(W=1210):
    """tools cmd/AbstractColorDepthMatchArgs.java:101-119 == test ImageTestUtils.java:12-24."""
    out = []
    if W > 270:
        out.append((W - 270, 0, W, 90))
    out.append((0, 0, 330, 100))
    return out


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = ctypes.CDLL(build())
    u8p = ctypes.POINTER(ctypes.c_uint8)
    u16p = ctypes.POINTER(ctypes.c_uint16)
    ip = ctypes.POINTER(ctypes.c_int)
    i64p = ctypes.POINTER(ctypes.c_int64)
    rp = ctypes.POINTER(Rect)
    L.cdso_pixel_gap.restype = ctypes.c_double
    L.cdso_pixel_gap.argtypes = [ctypes.c_int] * 6
    L.cdso_mask_positions.restype = ctypes.c_int
    L.cdso_mask_positions.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, rp, ctypes.c_int, ip, ip]
    L.cdso_query_create.restype = ctypes.c_void_p
    L.cdso_query_create.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                    ctypes.c_int, ctypes.c_int, rp, ctypes.c_int]
    L.cdso_query_free.argtypes = [ctypes.c_void_p]
    L.cdso_query_size.restype = ctypes.c_int
    L.cdso_query_size.argtypes = [ctypes.c_void_p]
    L.cdso_query_score.argtypes = [ctypes.c_void_p, u8p, ip, ip, ip]
    L.cdso_query_scan.argtypes = [ctypes.c_void_p, u8p, ctypes.c_long, ctypes.c_size_t, ip, u8p, ctypes.c_int,
                                  ctypes.c_int]
    L.cdso_normalized_pixel_score.restype = ctypes.c_float
    L.cdso_normalized_pixel_score.argtypes = [ctypes.c_int, ctypes.c_int]
    L.cdso_is_match.restype = ctypes.c_int
    L.cdso_is_match.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double]
    L.cdso_depth_lut.restype = u8p
    L.cdso_slice_number.restype = ctypes.c_int
    L.cdso_slice_number.argtypes = [ctypes.c_int] * 3
    L.cdso_slice_gap.restype = ctypes.c_int
    L.cdso_slice_gap.argtypes = [ctypes.c_int] * 6
    L.cdso_max_filter_rgb.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_double, u8p]
    L.cdso_mask_rgb.argtypes = [u8p, ctypes.c_long, ctypes.c_int, u8p]
    L.cdso_clear_regions.argtypes = [u8p, ctypes.c_int, ctypes.c_int, rp, ctypes.c_int, u8p]
    L.cdso_make_zgap.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, rp, ctypes.c_int, u8p]
    L.cdso_shape_prepare.argtypes = [u8p, ctypes.c_int, ctypes.c_int, rp, ctypes.c_int, u8p, u8p, u8p]
    L.cdso_shape_prepare_b.argtypes = [u8p, ctypes.c_int, ctypes.c_int, rp, ctypes.c_int, ctypes.c_int, u8p, u8p, u8p]
    L.cdso_shape_score_b.argtypes = [u8p, u8p, u8p, u8p, u8p, u16p, u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, rp, ctypes.c_int, ctypes.c_int, i64p]
    L.cdso_shape_score_value.restype = ctypes.c_int64
    L.cdso_shape_score_value.argtypes = [ctypes.c_int64, ctypes.c_int64]
    L.cdso_shape_score.argtypes = [u8p, u8p, u8p, u8p, u8p, u16p, u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, rp, ctypes.c_int, i64p]
    L.cdso_normalized_score.restype = ctypes.c_double
    L.cdso_normalized_score.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
    L.cdso_match_matrix.argtypes = [u8p, ctypes.c_int, u8p, ctypes.c_int, ctypes.c_int, ctypes.c_double, u8p]
    L.cdso_synth_image.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, u8p]
    L.cdso_synth_variants.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, u8p, u16p]
    _lib = L
    return L


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 3, a.shape
    return a


def pixel_gap(m, t):
    return lib().cdso_pixel_gap(int(m[0]), int(m[1]), int(m[2]), int(t[0]), int(t[1]), int(t[2]))


def match_matrix(mask_colors, target_colors, data_thr, ztol):
    """bool[n,m]: does target colour j match mask colour i (threshold + pixel gap <= ztol)."""
    mc = np.ascontiguousarray(mask_colors, dtype=np.uint8).reshape(-1, 3)
    tc = np.ascontiguousarray(target_colors, dtype=np.uint8).reshape(-1, 3)
    out = np.zeros((len(mc), len(tc)), dtype=np.uint8)
    lib().cdso_match_matrix(_u8(mc), len(mc), _u8(tc), len(tc), int(data_thr), float(ztol), _u8(out))
    return out.astype(bool)


def mask_positions(mask, thr, rects=None):
    mask = _img(mask)
    H, W, _ = mask.shape
    r, n = _rects(rects)
    pos = np.empty(W * H, dtype=np.int32)
    bbox = np.zeros(4, dtype=np.int32)
    P = lib().cdso_mask_positions(_u8(mask), W, H, int(thr), r, n, pos.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                  bbox.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return pos[:P].copy(), bbox


class PixelMatchQuery:
    """One prepared mask == one PixelMatchColorDepthSearchAlgorithm instance of the reference."""

    def __init__(self, mask, mask_thr, data_thr, ztol, xy_shift, mirror, rects=None):
        self.mask = _img(mask)
        self.H, self.W, _ = self.mask.shape
        r, n = _rects(rects)
        self._h = lib().cdso_query_create(_u8(self.mask), self.W, self.H, int(mask_thr), int(data_thr), float(ztol),
                                          int(xy_shift), int(bool(mirror)), r, n)
        if not self._h:
            raise ValueError("xyShift must be 0 or 2")
        self.size = lib().cdso_query_size(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().cdso_query_free(self._h)
            self._h = None

    def score(self, target):
        target = _img(target)
        assert target.shape == self.mask.shape
        s, m = ctypes.c_int(), ctypes.c_int()
        counts = np.zeros(18, dtype=np.int32)
        lib().cdso_query_score(self._h, _u8(target), ctypes.byref(s), ctypes.byref(m),
                               counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        return s.value, bool(m.value), counts

    def scan(self, library, n_threads=1, partition=100):
        """library: uint8[n, H, W, 3] (contiguous).  Returns (scores int32[n], mirrored uint8[n])."""
        library = np.ascontiguousarray(library, dtype=np.uint8)
        n = library.shape[0]
        assert library.shape[1:] == self.mask.shape
        scores = np.zeros(n, dtype=np.int32)
        mir = np.zeros(n, dtype=np.uint8)
        lib().cdso_query_scan(self._h, _u8(library), n, library.strides[0],
                              scores.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _u8(mir), int(n_threads),
                              int(partition))
        return scores, mir


def normalized_pixel_score(score, P):
    return float(lib().cdso_normalized_pixel_score(int(score), int(P)))


def is_match(score, P, pct_positive_pixels):
    return bool(lib().cdso_is_match(int(score), int(P), float(pct_positive_pixels)))


def depth_lut():
    p = lib().cdso_depth_lut()
    return np.ctypeslib.as_array(p, shape=(256, 3)).copy()


def slice_number(rgb):
    return lib().cdso_slice_number(int(rgb[0]), int(rgb[1]), int(rgb[2]))


def max_filter_rgb(img, radius):
    img = _img(img)
    H, W, _ = img.shape
    out = np.empty_like(img)
    lib().cdso_max_filter_rgb(_u8(img), W, H, float(radius), _u8(out))
    return out


def make_zgap(target, thr, radius=10, rects=None):
    target = _img(target)
    H, W, _ = target.shape
    r, n = _rects(rects)
    out = np.empty_like(target)
    lib().cdso_make_zgap(_u8(target), W, H, int(thr), float(radius), r, n, _u8(out))
    return out


class ShapeQuery:
    """One prepared mask == one Shape2DMatchColorDepthSearchAlgorithm instance of the reference."""

    def __init__(self, mask, query_thr, mirror, rects=None, roi=None, border=0):
        mask = _img(mask)
        self.H, self.W, _ = mask.shape
        self.thr, self.mirror, self.rects = int(query_thr), bool(mirror), list(rects or [])
        self.border = int(border)   # queryBorderSize (ColorDepthSearchAlgorithmProviderFactory.java:103)
        r, n = _rects(self.rects)
        self.Q = np.empty_like(mask)
        self.QM = np.empty((self.H, self.W), dtype=np.uint8)
        self.QH = np.empty((self.H, self.W), dtype=np.uint8)
        lib().cdso_shape_prepare_b(_u8(mask), self.W, self.H, r, n, self.border, _u8(self.Q), _u8(self.QM), _u8(self.QH))
        self.roi = None
        if roi is not None:
            roi = np.ascontiguousarray(roi, dtype=np.uint8).copy()
            for (x0, y0, x1, y1) in self.rects:
                roi[max(0, y0):y1, max(0, x0):x1] = 0
            self.roi = roi

    def score(self, target, grad, zgap):
        """Returns (gradientAreaGap, highExpressionArea, score, mirrored)."""
        out = np.zeros(3, dtype=np.int64)
        r, n = _rects(self.rects)
        target = _img(target)
        gp = zp = None
        if grad is not None:
            grad = np.ascontiguousarray(grad, dtype=np.uint16)
            gp = grad.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))
        if zgap is not None:
            zgap = _img(zgap)
            zp = _u8(zgap)
        lib().cdso_shape_score_b(_u8(self.Q), _u8(self.QM), _u8(self.QH), _u8(self.roi) if self.roi is not None else None,
                                 _u8(target), gp, zp, self.W, self.H, self.thr, int(self.mirror), r, n, self.border,
                                 out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        gap, high, mir = int(out[0]), int(out[1]), bool(out[2])
        return gap, high, int(np.int32(lib().cdso_shape_score_value(gap, high))), mir


def shape_score_value(gap, high):
    return int(lib().cdso_shape_score_value(int(gap), int(high)))


def normalized_score(pix, shape, max_pix, max_shape):
    return float(lib().cdso_normalized_score(int(pix), int(shape), int(max_pix), int(max_shape)))


def synth_image(seed, index, kind, W=1210, H=566):
    out = np.empty((H, W, 3), dtype=np.uint8)
    lib().cdso_synth_image(ctypes.c_uint64(seed), ctypes.c_uint64(index), int(kind), W, H, _u8(out))
    return out


def synth_variants(seed, index, W=1210, H=566, want_zgap=True, want_grad=True):
    z = np.empty((H, W, 3), dtype=np.uint8) if want_zgap else None
    g = np.empty((H, W), dtype=np.uint16) if want_grad else None
    lib().cdso_synth_variants(ctypes.c_uint64(seed), ctypes.c_uint64(index), W, H,
                              _u8(z) if z is not None else None,
                              g.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)) if g is not None else None)
    return z, g
