#!/usr/bin/env python3
"""Per-target fixed cost vs per-band cost of the single-mask scan: time the benchmark mask cropped to its first k bands."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from colormipsearch_amd import cds
W, H = 1210, 566
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = cds.default_context()
lib = cds.Library(W, H, n)
lib.generate_synthetic(0xC0FFEE, 0, 0, n, kind=0)
mlib = cds.Library(W, H, 1)
mlib.generate_synthetic(0xBADC0DE, 0, 0, 1, kind=2)
mask = mlib.download(0, 1)[0]
rects = cds.label_regions(W)
os.environ["CDS_TIME_KERNELS"] = "1"
for rows in (8, 20, 44, 92, 170):
    m = mask.copy(); m[165 + rows:] = 0
    alg = cds.PixelMatchColorDepthSearchAlgorithm(m, 20, True, None, 0, False, 20, 0.01, 2, rects)
    best = 1e9
    for _ in range(4):
        alg.searchLibrary(lib)
        ms, nl = alg.lastKernelMs()
        best = min(best, ms)
    print(json.dumps({"mask_rows": rows, "bands": (rows + 11) // 12, "kernel_ms": round(best, 3), "targets": n}))
