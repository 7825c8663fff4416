#!/usr/bin/env python3
"""Where the end-to-end time of configs[2] goes (processor.findAllColorDepthMatchesTable over a synthetic sparse library):
timers around the mask preparation, the device search (scan + isMatch + compaction), the list download and the host-side
table assembly.  --targets (default 1,048,576), --masks (default 1024), --set (mask-set size, default 128)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from colormipsearch_amd import cds, processor as PR  # noqa: E402

W, H = 1210, 566


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--targets", type=int, default=1 << 20)
    ap.add_argument("--masks", type=int, default=1024)
    ap.add_argument("--set", type=int, default=128)
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    ctx = cds.Context(0)
    n = args.targets
    stage = cds.Library(W, H, 8192, context=ctx)
    sl = cds.SparseLibrary(W, H, n, 24000 * n + 4_000_000, context=ctx)
    for f in range(0, n, 8192):
        c = min(8192, n - f)
        stage.generate_synthetic(0xC0FFEE, f, 0, c, kind=0)
        sl.append_from(stage, 0, c)
    ml = cds.Library(W, H, args.masks, context=ctx)
    ml.generate_synthetic(0xBADC0DE, 0, 0, args.masks, kind=2)
    masks = ml.download(0, args.masks)
    del ml, stage
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 1.0, 2, cds.label_regions, context=ctx)
    proc = PR.GpuColorMIPSearchProcessor(cds.ColorMIPSearch(1.0, 20, provider), None, None, first=0, count=n, sparse_library=sl)
    T = {}

    def wrap(obj, name, key):
        f = getattr(obj, name)

        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            T[key] = T.get(key, 0.0) + time.perf_counter() - t0
            return r
        setattr(obj, name, g)
    wrap(PR.GpuColorMIPSearchProcessor, "_device_match_table", "device_match_table(total)")
    wrap(cds.PixelMatchColorDepthSearchAlgorithm, "searchSparseLibraryMatches", "searchSparseLibraryMatches")
    lib = cds.capi.load()
    for rep in range(args.reps):
        T.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        table = proc.findAllColorDepthMatchesTable(masks, mask_set_size=args.set)
        wall = time.perf_counter() - t0
        print(json.dumps({"rep": rep, "wall_s": round(wall, 3), "matches": len(table), **{k: round(v, 3) for k, v in T.items()},
                          "pipeline": {k: round(v, 3) for k, v in getattr(proc, "last_pipeline", {}).items()}}), flush=True)


if __name__ == "__main__":
    main()
