#!/usr/bin/env python3
"""Phases of cds_maskset_create (CDS_TIME_PREP=1) for mask sets of --set masks out of --masks synthetic benchmark masks."""
import argparse, os, sys, time
os.environ["CDS_TIME_PREP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colormipsearch_amd import cds
ap = argparse.ArgumentParser()
ap.add_argument("--masks", type=int, default=256)
ap.add_argument("--set", type=int, default=128)
a = ap.parse_args()
W, H = 1210, 566
ctx = cds.Context(0)
ml = cds.Library(W, H, a.masks, context=ctx)
ml.generate_synthetic(0xBADC0DE, 0, 0, a.masks, kind=2)
masks = ml.download(0, a.masks)
rects = cds.label_regions(W)
for m0 in range(0, a.masks, a.set):
    t0 = time.perf_counter()
    alg = cds.PixelMatchColorDepthSearchAlgorithm(masks[m0:m0 + a.set], 20, True, None, 0, False, 20, 0.01, 2, rects, context=ctx)
    t1 = time.perf_counter()
    alg.close()
    print(f"set at {m0}: create {1e3 * (t1 - t0):.1f} ms, close {1e3 * (time.perf_counter() - t1):.1f} ms", file=sys.stderr)
