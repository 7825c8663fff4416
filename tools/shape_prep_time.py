#!/usr/bin/env python3
"""Time of the shape-mask preparation (cds_shapeset_create) and of the pixel-match mask preparation (cds_maskset_create)
per mask, for `--masks` synthetic benchmark masks."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from colormipsearch_amd import cds
W, H = 1210, 566
ap = argparse.ArgumentParser()
ap.add_argument("--masks", type=int, default=128)
a = ap.parse_args()
mlib = cds.Library(W, H, a.masks); mlib.generate_synthetic(0xBADC0DE, 0, 0, a.masks, kind=2); masks = mlib.download(0, a.masks)
cds.Shape2DMatchColorDepthSearchAlgorithm(masks[:2], 20, True, None, cds.label_regions)   # warm-up (slice table, kernels)
t0 = time.perf_counter(); sh = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, 20, True, None, cds.label_regions); t_shape = time.perf_counter() - t0
qm, qh = sh.maskStats()
t0 = time.perf_counter(); pm = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, cds.label_regions); t_pix = time.perf_counter() - t0
print(json.dumps({"masks": a.masks, "shapeset_create_s": round(t_shape, 3), "ms_per_mask_shape": round(1e3 * t_shape / a.masks, 3),
                  "maskset_create_s": round(t_pix, 3), "ms_per_mask_pixmatch": round(1e3 * t_pix / a.masks, 3),
                  "qm_sum0": int(qm[0]), "qh_sum0": int(qh[0])}))
