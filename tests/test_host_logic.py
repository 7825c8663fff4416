"""CPU-side tests of the product's host logic (no GPU needed):
 * libcds_hip.so loads and exports every symbol include/*.h declares;
 * the integer restatement of the reference's double predicate (hue codes + acceptance intervals,
   colormipsearch_amd/csrc/cds_host.hip) equals the oracle's calculatePixelGap <= zTolerance for
   every target colour -- with a host reciprocal table standing in for the GPU's v_rcp_f32 (the logic
   only needs host and "device" to share the table)."""
import ctypes
import os
import re

import numpy as np
import pytest

from colormipsearch_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    declared = set()
    for h in ("cds_hip.h", "cds_hip_debug.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        declared |= set(re.findall(r"\b(cds_[a-z0-9_]+)\s*\(", src))
    assert len(declared) >= 35
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/ but not exported"
        assert name in capi.SIGNATURES, f"{name} has no ctypes signature in capi.py"


def test_no_device_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = capi.load()
    h = ctypes.c_void_p()
    rc = lib.cds_context_create(0, ctypes.byref(h))
    assert rc == capi.CDS_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.cds_last_error()


def test_pure_host_arithmetic(oracle):
    lib = capi.load()
    from tests.golden import expected as E
    for pix, gap, high, max_pix, max_neg, shape, norm in E.NORMALIZED:  # GradientAreaGapUtilsTest.scoreCalculator
        s = lib.cds_shape_score_value(gap, high)
        assert s == shape
        assert abs(lib.cds_normalized_score(pix, s, max_pix, max_neg) - norm) <= 0.1
        assert lib.cds_normalized_score(pix, s, max_pix, max_neg) == oracle.normalized_score(pix, s, max_pix, max_neg)
    assert lib.cds_shape_score_value(-1, 5) == -1
    for score, P, pct in ((0, 100, 1.0), (1, 100, 1.0), (2, 100, 1.0), (5, 0, 0.0), (123, 10299, 1.0), (103, 10299, 1.0)):
        assert bool(lib.cds_is_match(score, P, pct)) == oracle.is_match(score, P, pct)
        assert lib.cds_pixel_match_normalized(score, P) == oracle.normalized_pixel_score(score, P)


def _host_rcp():
    t = np.zeros(256, dtype=np.float32)
    t[1:] = (np.float32(1.0) / np.arange(1, 256, dtype=np.float32)).astype(np.float32)
    t[0] = np.inf
    return t


def _codes(rcp, colors, thr):
    lib = capi.load()
    p = rcp.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    return np.array([lib.cds_debug_hue_code(p, int(c[0]), int(c[1]), int(c[2]), thr) for c in colors], dtype=np.uint32)


def _intervals(rcp, color, ztol):
    lib = capi.load()
    out = np.zeros(4, dtype=np.uint32)
    capi.check(lib.cds_debug_intervals(rcp.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(color[0]), int(color[1]),
                                       int(color[2]), float(ztol), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
    return out


def _target_palette(rng, n_random):
    cols = [rng.integers(0, 256, size=(n_random, 3))]
    # every (second, max) ratio in every class at a few brightness levels, ties and zeros included
    for mx in (255, 254, 200, 128, 77, 51, 22, 21, 20, 3):
        md = np.arange(0, mx + 1)
        lo = np.minimum(md, rng.integers(0, 256, size=md.shape))
        for perm in ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)):
            c = np.stack([np.full_like(md, mx), md, lo], axis=1)[:, perm]
            cols.append(c)
    return np.concatenate(cols).astype(np.uint8)


@pytest.mark.parametrize("ztol,thr", [(0.01, 20), (0.02, 100), (0.0, 0), (0.3, 20), (2.0, 20), (-0.05, 20)])
def test_integer_predicate_equals_reference_predicate(oracle, ztol, thr):
    rng = np.random.default_rng(1234)
    rcp = _host_rcp()
    targets = _target_palette(rng, 4000)
    tcodes = _codes(rcp, targets, thr)
    masks = np.concatenate([rng.integers(0, 256, size=(150, 3)),
                            oracle.depth_lut()[::5],
                            np.array([[0, 0, 255], [0, 255, 0], [255, 0, 0], [10, 10, 200], [200, 200, 10], [90, 87, 255],
                                      [0, 254, 255], [254, 255, 0], [255, 0, 200], [127, 0, 255], [1, 0, 2], [255, 254, 253]])]).astype(np.uint8)
    want = oracle.match_matrix(masks, targets, thr, ztol)
    for i, m in enumerate(masks):
        lo1, len1, lo2, len2 = (int(v) for v in _intervals(rcp, m, ztol))
        got = (((tcodes - np.uint32(lo1)) & 0xFFFFFFFF) <= len1) | (((tcodes - np.uint32(lo2)) & 0xFFFFFFFF) <= len2)
        bad = np.flatnonzero(got != want[i])
        assert len(bad) == 0, (m.tolist(), targets[bad[:5]].tolist(), got[bad[:5]], want[i][bad[:5]])


def test_hue_code_is_zero_below_threshold_and_for_ties():
    rcp = _host_rcp()
    c = _codes(rcp, np.array([[0, 0, 0], [20, 20, 20], [20, 3, 1], [200, 200, 10], [21, 3, 1], [0, 0, 21]], dtype=np.uint8), 20)
    assert list(c[:4]) == [0, 0, 0, 0] and c[4] != 0 and c[5] != 0
    assert (c[4] >> 29) == 5 and (c[5] >> 29) == 2      # RG, BG (r <= g)


@pytest.mark.parametrize("ztol", [0.0, 0.01, 0.05, 0.3])
def test_band_map_bits_cover_every_code_of_an_interval(oracle, ztol):
    """Stage 1 of the sparse scan drops a target pixel when its hue band's bit is clear in the cell's word.  For that to be
    exact, every code an interval contains must have its band among cds_debug_band_bits(interval): checked for all 195,840
    (class, second, max) target colours against the intervals of LUT, random and boundary mask colours."""
    lib = capi.load()
    rcp = _host_rcp()
    p = rcp.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    # every record colour: class 1..6 (BR, BG, GB, GR, RG, RB as in cds_internal.hpp), second < max
    orders = {1: (2, 0, 1), 2: (2, 1, 0), 3: (1, 2, 0), 4: (1, 0, 2), 5: (0, 1, 2), 6: (0, 2, 1)}   # (max channel, second, third)
    cols, cls, sec, mxs = [], [], [], []
    for c, order in orders.items():
        for mx in range(1, 256):
            s = np.arange(mx)
            px = np.zeros((mx, 3), dtype=np.uint8)
            px[:, order[0]], px[:, order[1]] = mx, s
            cols.append(px), cls.append(np.full(mx, c)), sec.append(s), mxs.append(np.full(mx, mx))
    cols, cls, sec, mxs = np.concatenate(cols), np.concatenate(cls), np.concatenate(sec), np.concatenate(mxs)
    codes = _codes(rcp, cols, 0)
    # (with second == third == 0 the reference's strict comparisons pick the class, e.g. (0, 0, 9) is BG, not BR: take the
    # class from the code itself; everywhere else it is the one enumerated)
    assert np.array_equal((codes >> 29)[sec > 0], cls.astype(np.uint32)[sec > 0])
    cls = (codes >> 29).astype(np.int64)
    assert cls.min() >= 1 and cls.max() <= 6
    bands = np.array([lib.cds_debug_band_of(p, int(c), int(s), int(m)) for c, s, m in zip(cls[::7], sec[::7], mxs[::7])])
    hf = (sec.astype(np.float32) * rcp[mxs]).astype(np.float32)
    band_all = ((cls * 5 + (hf * np.float32(5.0)).astype(np.uint32)) & 31).astype(np.uint32)
    assert np.array_equal(bands, band_all[::7])                          # numpy restatement == the library's arithmetic
    assert not np.isin(band_all, (3, 4)).any()                           # bit 3 is where pixels below the threshold are sent
    rng = np.random.default_rng(77)
    masks = np.concatenate([oracle.depth_lut()[::3], rng.integers(0, 256, size=(120, 3)),
                            np.array([[255, 51, 0], [255, 50, 0], [255, 102, 1], [250, 100, 0], [0, 204, 255], [0, 255, 254], [254, 255, 0],
                                      [255, 0, 0], [30, 0, 255], [128, 127, 3], [21, 20, 1], [255, 254, 253]])]).astype(np.uint8)
    out = np.zeros(1, dtype=np.uint32)
    for m in masks:
        iv = [int(v) for v in _intervals(rcp, m, ztol)]
        bits = 0
        for lo, ln in ((iv[0], iv[1]), (iv[2], iv[3])):
            capi.check(lib.cds_debug_band_bits(lo, ln, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
            bits |= int(out[0])
        assert bits & 0b11000 == 0
        inside = (((codes - np.uint32(iv[0])) & 0xFFFFFFFF) <= iv[1]) | (((codes - np.uint32(iv[2])) & 0xFFFFFFFF) <= iv[3])
        missed = inside & (((bits >> band_all) & 1) == 0)
        assert not missed.any(), (m.tolist(), ztol, cols[missed][:5].tolist())


def test_load_only_diagnostic_is_not_process_wide():
    """cds_debug_sparse_load_only takes a context (round 3 had a process-global switch that turned every search of the process into
    the loads-only twin): without a context there is nothing to switch, and the call says so instead of flipping a global."""
    lib = capi.load()
    assert lib.cds_debug_sparse_load_only(None, 1) == -1      # CDS_ERR_INVALID_ARGUMENT
    assert lib.cds_debug_sparse_load_only_state(None) == 0
    hdr = open(os.path.join(ROOT, "include", "cds_hip_debug.h")).read()
    assert re.search(r"int cds_debug_sparse_load_only\(cds_context \*ctx, int32_t on\);", hdr)
    src = open(os.path.join(ROOT, "colormipsearch_amd", "csrc", "cds_sparse.hip")).read()
    assert not re.search(r"\bg_sparse_load_only\b", src) and "static std::atomic<int> g_" not in src
