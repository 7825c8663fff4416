#!/usr/bin/env python3
"""Single masks of growing size: single-mask kernel vs the batch kernel with one mask (CDS_SCAN_SINGLE_AS_BATCH)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from colormipsearch_amd import cds
W, H = 1210, 566
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
lib = cds.Library(W, H, n)
lib.generate_synthetic(0xC0FFEE, 0, 0, n, kind=0)
mlib = cds.Library(W, H, 8)
mlib.generate_synthetic(0xBADC0DE, 0, 0, 8, kind=2)
ms = mlib.download(0, 8)
rects = cds.label_regions(W)
os.environ["CDS_TIME_KERNELS"] = "1"
for k in (1, 2, 4, 8):
    m = ms[:k].max(axis=0)   # union of k arbors: ~k x 11 K pixels
    P = int((m.max(axis=2) > 20).sum())
    row = {"mask_pixels": P}
    for mode in ("0", "1"):
        os.environ["CDS_SCAN_SINGLE_AS_BATCH"] = mode
        alg = cds.PixelMatchColorDepthSearchAlgorithm(m, 20, True, None, 0, False, 20, 0.01, 2, rects)
        best = 1e9
        for _ in range(3):
            s, _m = alg.searchLibrary(lib)
            best = min(best, alg.lastKernelMs()[0])
        row["batch_kernel_ms" if mode == "1" else "single_kernel_ms"] = round(best, 3)
        row["sum" + mode] = int(s.sum())
    print(json.dumps(row))
