#!/usr/bin/env python3
"""Latency of the per-pair entry points (the granularity of the reference's calculateMatchingScore)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from colormipsearch_amd import cds
W, H = 1210, 566
rects = cds.label_regions(W)
lib = cds.Library(W, H, 2)
lib.generate_synthetic(0xC0FFEE, 5, 0, 2, kind=0, with_variants=True)
t = lib.download(0, 1)[0]
g, z = lib.download_variants(0, 1)
mlib = cds.Library(W, H, 1)
mlib.generate_synthetic(0xBADC0DE, 0, 0, 1, kind=2)
mask = mlib.download(0, 1)[0]
out = {}
alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
alg.calculateMatchingScore(t)
t0 = time.perf_counter()
for _ in range(100):
    s = alg.calculateMatchingScore(t)
out["pixel_match_pair_ms"] = round((time.perf_counter() - t0) / 100 * 1e3, 3)
sh = cds.Shape2DMatchColorDepthSearchAlgorithm(mask, 20, True, None, rects)
variants = {cds.ComputeFileType.GradientImage: lambda: g[0], cds.ComputeFileType.ZGapImage: lambda: z[0]} if hasattr(cds, "ComputeFileType") else {"GradientImage": lambda: g[0], "ZGapImage": lambda: z[0]}
try:
    sh.calculateMatchingScore(t, variants)
    t0 = time.perf_counter()
    for _ in range(50):
        sh.calculateMatchingScore(t, variants)
    out["shape_pair_ms"] = round((time.perf_counter() - t0) / 50 * 1e3, 3)
except Exception as e:  # the mirror's variant-supplier keys may differ
    out["shape_pair_error"] = str(e)[:100]
print(json.dumps(out))

# small batched calls (the reference's per-mask granularity): 500 pairs / 1 mask x 4096 targets per call
lib2 = cds.Library(W, H, 4096)
lib2.generate_synthetic(0xC0FFEE, 0, 0, 4096, kind=0, with_variants=True)
pm = np.zeros(500, dtype=np.int32); ps = np.arange(500, dtype=np.int64) * 8
sh.scoreLibraryPairs(lib2, pm, ps)
t0 = time.perf_counter()
for _ in range(50):
    sh.scoreLibraryPairs(lib2, pm, ps)
out2 = {"gradscore_500_pairs_call_ms": round((time.perf_counter() - t0) / 50 * 1e3, 3), "gradscore_kernel_ms": round(sh.lastKernelMs()[0], 3)}
alg.searchLibrary(lib2)
t0 = time.perf_counter()
for _ in range(50):
    alg.searchLibrary(lib2)
out2["search_4096_targets_call_ms"] = round((time.perf_counter() - t0) / 50 * 1e3, 3)
out2["search_kernel_ms"] = round(alg.lastKernelMs()[0], 3)
print(json.dumps(out2))

# many host threads on ONE algorithm object (the reference's usage: LocalColorMIPSearchProcessor.java:88-105): the mask set's
# slot pool lets their uploads / launches / read-backs overlap (ctypes releases the GIL during a call)
from concurrent.futures import ThreadPoolExecutor
targets = [lib2.download(i, 1)[0] for i in range(16)]
out3 = {}
for n_thr in (1, 4, 8, 16):
    with ThreadPoolExecutor(max_workers=n_thr) as pool:
        list(pool.map(lambda i: alg.calculateMatchingScore(targets[i % 16]), range(64)))
        t0 = time.perf_counter()
        list(pool.map(lambda i: alg.calculateMatchingScore(targets[i % 16]), range(2000)))
        out3[f"pixel_match_pairs_per_s_{n_thr}_threads"] = round(2000 / (time.perf_counter() - t0))
print(json.dumps(out3))
