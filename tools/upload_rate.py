#!/usr/bin/env python3
"""Host -> resident library upload rate (what a caller with host buffers pays before the first scan)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from colormipsearch_amd import cds
W, H = 1210, 566
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
imgs = np.random.default_rng(0).integers(0, 255, size=(n, H, W, 3), dtype=np.uint8)
lib = cds.Library(W, H, n)
lib.upload(0, imgs[:64])
t0 = time.perf_counter(); lib.upload(0, imgs); dt = time.perf_counter() - t0
t0 = time.perf_counter(); back = lib.download(0, n); dt2 = time.perf_counter() - t0
print(json.dumps({"images": n, "upload_GBps": round(imgs.nbytes / dt / 1e9, 2), "upload_images_per_s": round(n / dt),
                  "download_GBps": round(imgs.nbytes / dt2 / 1e9, 2), "roundtrip_equal": bool(np.array_equal(back, imgs))}))
