#!/usr/bin/env python3
"""Where the many-mask pipeline's time goes (run on the GPU box): the stages of processor._pipelined_match_table one after the other, timed per
mask set -- preparation (cds_maskset_create), scan + device match list (cds_search_sparse_match_list), list download + columns
(cds_match_list_fetch_columns), destruction -- and then the pipelined call itself.  TARGETS=1048576 MASKS=1024 by default."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch  # noqa: F401  (loads the HIP runtime first)
from colormipsearch_amd import capi, cds, processor as P

W, H = 1210, 566
n, n_m = int(os.environ.get("TARGETS", 1048576)), int(os.environ.get("MASKS", 1024))
dense = cds.Library(W, H, 8192)
sl = cds.SparseLibrary(W, H, n, 24000 * n + 4_000_000)
for f in range(0, n, 8192):
    dense.generate_synthetic(0xC0FFEE, f, 0, 8192, kind=0)
    sl.append_from(dense, 0, 8192)
mlib = cds.Library(W, H, n_m)
mlib.generate_synthetic(0xBADC0DE, 0, 0, n_m, kind=2)
masks = mlib.download(0, n_m)
provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 0.01, 2, cds.label_regions)
search = cds.ColorMIPSearch(0.01, 20, provider)
lib = capi.load()
tot = dict(prep=0.0, scan=0.0, fetch=0.0, close=0.0)
for m0 in range(0, n_m, 128):
    t0 = time.perf_counter()
    alg = search.createQueryColorDepthSearchWithDefaultThreshold(masks[m0:m0 + 128])
    t1 = time.perf_counter()
    h, cnt = ctypes.c_void_p(), ctypes.c_int64()
    capi.check(lib.cds_search_sparse_match_list(alg._h, sl._h, 0, n, 0.01, ctypes.byref(h), ctypes.byref(cnt)))
    t2 = time.perf_counter()
    rows = cnt.value
    m, t, sc = np.empty(rows, np.int32), np.empty(rows, np.int64), np.empty(rows, np.int32)
    ratio, mir = np.empty(rows, np.float32), np.empty(rows, np.uint8)
    sizes = np.ascontiguousarray(alg.getQuerySizes(), dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    capi.check(lib.cds_match_list_fetch_columns(h, 0, rows, vp(sizes), len(sizes), m0, vp(m), vp(t), vp(sc), vp(ratio), vp(mir)))
    t3 = time.perf_counter()
    lib.cds_match_list_destroy(h)
    alg.close()
    t4 = time.perf_counter()
    print(f"set {m0:4d}: prepare {1e3 * (t1 - t0):6.1f} ms  scan+list {1e3 * (t2 - t1):6.1f} ms  fetch+columns {1e3 * (t3 - t2):6.1f} ms ({rows} rows)  destroy {1e3 * (t4 - t3):5.1f} ms", flush=True)
    tot["prep"] += t1 - t0; tot["scan"] += t2 - t1; tot["fetch"] += t3 - t2; tot["close"] += t4 - t3
print("sequential totals (s):", {k: round(v, 3) for k, v in tot.items()}, flush=True)
proc = P.GpuColorMIPSearchProcessor(search, None, None, sparse_library=sl)
# timeline of the pipelined call: every preparation and every search call with its thread
import threading
events = []
_make = search.createQueryColorDepthSearchWithDefaultThreshold
def make_logged(m):
    t0 = time.perf_counter()
    a = _make(m)
    events.append(("prepare", threading.get_ident() % 1000, t0, time.perf_counter(), len(m)))
    _cols = a.searchSparseLibraryMatchColumns
    def cols_logged(*args, **kw):
        t1 = time.perf_counter()
        r = _cols(*args, **kw)
        events.append(("search+fetch", threading.get_ident() % 1000, t1, time.perf_counter(), len(m)))
        return r
    a.searchSparseLibraryMatchColumns = cols_logged
    _close = a.close
    def close_logged():
        t2 = time.perf_counter()
        _close()
        events.append(("close", threading.get_ident() % 1000, t2, time.perf_counter(), len(m)))
    a.close = close_logged
    return a
search.createQueryColorDepthSearchWithDefaultThreshold = make_logged
for rep in range(2):
    t0 = time.perf_counter()
    tab = proc.findAllColorDepthMatchesTable(masks, mask_set_size=128)
    t1 = time.perf_counter()
    print(f"pipelined: {t1 - t0:.3f} s, rows {len(tab)}, stages {proc.last_pipeline}", flush=True)
    for what, th, a, b, k in sorted(events, key=lambda e: e[2]):
        print(f"   {what:13s} thread {th:3d}  {1e3 * (a - t0):7.1f} .. {1e3 * (b - t0):7.1f} ms  ({k} masks)")
    events.clear()
