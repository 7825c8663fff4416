#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md (NOT the driver's bench line; timing only -- the same workloads are checked
against the oracle by tests/test_bench_shapes_gpu.py): BASELINE configs[2] (mask batches sharing
one pass over the library) and configs[4] (GradientAreaGap on the top hits per mask), on one MI355X.

    python tools/bench_extra.py --masks 128 --targets 16384         # config 3 in miniature
    python tools/bench_extra.py --gradscore --masks 32 --top 500    # config 5 in miniature
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H = 1210, 566


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--masks", type=int, default=128)
    ap.add_argument("--targets", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--gradscore", action="store_true")
    ap.add_argument("--top", type=int, default=500)
    ap.add_argument("--check", type=int, default=0, help="(ignored; parity of these workloads is tested in tests/test_bench_shapes_gpu.py)")
    args = ap.parse_args()
    from colormipsearch_amd import cds
    ctx = cds.default_context()
    rects = cds.label_regions(W)
    mlib = cds.Library(W, H, args.masks)
    mlib.generate_synthetic(0xBADC0DE, 0, 0, args.masks, kind=2)
    masks = mlib.download(0, args.masks)
    lib = cds.Library(W, H, args.targets)
    lib.generate_synthetic(0xC0FFEE, 0, 0, args.targets, kind=0, with_variants=args.gradscore)
    out = {}
    if not args.gradscore:
        t0 = time.perf_counter()
        alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects)
        prep_s = time.perf_counter() - t0
        best = None
        for _ in range(args.reps):
            t0 = time.perf_counter()
            scores, mir = alg.searchLibrary(lib)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        kms, nl = alg.lastKernelMs()
        out = {"workload": f"{args.masks} masks x {args.targets} targets (BASELINE configs[2] shape), xyShift 2, mirror",
               "comparisons_per_s_wall": args.masks * args.targets / best,
               "comparisons_per_s_kernels": args.masks * args.targets / (kms * nl * 1e-3),
               "kernel_ms_avg": kms, "launches": nl, "mask_prep_s": prep_s}
    else:
        t0 = time.perf_counter()
        alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, 20, True, None, rects)
        prep_s = time.perf_counter() - t0
        rng = np.random.default_rng(5)
        pm = np.repeat(np.arange(args.masks), args.top).astype(np.int32)
        ps = np.concatenate([rng.choice(args.targets, size=args.top, replace=False) for _ in range(args.masks)]).astype(np.int64)
        best = None
        for _ in range(args.reps):
            t0 = time.perf_counter()
            gap, high, mir = alg.scoreLibraryPairs(lib, pm, ps)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        kms, nl = alg.lastKernelMs()
        n = len(pm)
        out = {"workload": f"GradientAreaGap on top-{args.top} x {args.masks} masks = {n} pairs (BASELINE configs[4] shape)",
               "pairs_per_s_wall": n / best, "pairs_per_s_kernel": n / (kms * 1e-3), "kernel_ms": kms,
               "algorithmic_GBps": n * 2081200 / (kms * 1e-3) / 1e9, "mask_prep_s": prep_s}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
