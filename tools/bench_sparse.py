#!/usr/bin/env python3
"""Sparse-library scan: build a synthetic sparse library of --targets images (through dense staging chunks), time the scan
with --masks benchmark masks, and check the first staging chunk against the dense scan.
Knobs (environment, read when the mask set is created / at launch): CDS_SPARSE_THREADS, CDS_SPARSE_LDS_KB,
CDS_SPARSE_WGS_PER_CU, CDS_SPARSE_GROUP."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from colormipsearch_amd import cds  # noqa: E402

W, H = 1210, 566


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--targets", type=int, default=65536)
    ap.add_argument("--stage", type=int, default=16384)
    ap.add_argument("--masks", type=int, default=1)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--kind", type=int, default=0)
    ap.add_argument("--check", type=int, default=1)
    ap.add_argument("--sweep", default="")
    args = ap.parse_args()
    ctx = cds.Context(0)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    n, st = args.targets, min(args.stage, args.targets)
    dense = cds.Library(W, H, st, context=ctx)
    per_img = 46000 if args.kind == 0 else W * H + H
    sl = cds.SparseLibrary(W, H, n, per_img * n if args.kind else 24000 * n + 4_000_000, context=ctx, density_limit=2.0 if args.kind else None)
    t0 = time.time()
    tg = ta = 0.0
    for first in range(0, n, st):
        c = min(st, n - first)
        t1 = time.time()
        dense.generate_synthetic(0xC0FFEE, first, 0, c, kind=args.kind)
        t2 = time.time()
        sl.append_from(dense, 0, c)
        t3 = time.time()
        tg += t2 - t1
        ta += t3 - t2
    build_s = time.time() - t0
    if n > st:   # leave the first chunk in the staging library for the dense comparison
        dense.generate_synthetic(0xC0FFEE, 0, 0, st, kind=args.kind)
    mlib = cds.Library(W, H, args.masks, context=ctx)
    mlib.generate_synthetic(0xBADC0DE, 0, 0, args.masks, kind=2)
    masks = mlib.download(0, args.masks)
    rects = cds.label_regions(W)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects, context=ctx)
    scores = torch.empty(args.masks * n, dtype=torch.int32, device="cuda")
    for _ in range(args.warmup):
        alg.searchSparseLibraryDevice(sl, 0, n, scores.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(args.reps):
        alg.searchSparseLibraryDevice(sl, 0, n, scores.data_ptr())
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    out = {"targets": n, "masks": args.masks, "kind": args.kind, "records": sl.pixels, "bytes_per_target": 4 * sl.pixels / n + 4 * sl.n_strips * (H + 1) + 8,
           "build_s": round(build_s, 3), "generate_s": round(tg, 3), "append_s": round(ta, 3),
           "scan_ms": ms, "comparisons_per_s": args.masks * n / (ms * 1e-3),
           "env": {k: v for k, v in os.environ.items() if k.startswith("CDS_SPARSE")}}
    if args.check:
        d = torch.empty(args.masks * st, dtype=torch.int32, device="cuda")
        alg.searchLibraryDevice(dense, 0, st, d.data_ptr())
        torch.cuda.synchronize()
        e0.record(stream)
        alg.searchLibraryDevice(dense, 0, st, d.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        out["dense_ms_per_stage_chunk"] = e0.elapsed_time(e1)
        out["dense_comparisons_per_s"] = args.masks * st / (out["dense_ms_per_stage_chunk"] * 1e-3)
        same = bool(torch.equal(d.view(args.masks, st), scores.view(args.masks, n)[:, :st]))
        out["equals_dense_scan"] = same
        out["nonzero_scores"] = int((scores.view(args.masks, n)[:, :st] & 0x7FFFFFFF).gt(0).sum().item())
    print(json.dumps(out))
    if args.check and not out["equals_dense_scan"]:
        raise SystemExit("sparse scan differs from the dense scan")
    # --sweep "THREADS:LDS_KB:GROUP:WGS,..." : re-time the scan under other layouts (the mask set is rebuilt per entry)
    for item in [x for x in args.sweep.split(",") if x]:
        th, kb, grp, wgs = (item.split(":") + ["", "", "", ""])[:4]
        for k, v in (("CDS_SPARSE_THREADS", th), ("CDS_SPARSE_LDS_KB", kb), ("CDS_SPARSE_GROUP", grp), ("CDS_SPARSE_WGS_PER_CU", wgs)):
            if v:
                os.environ[k] = v
            else:
                os.environ.pop(k, None)
        a2 = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects, context=ctx)
        s2 = torch.empty_like(scores)
        for _ in range(max(1, args.warmup // 2)):
            a2.searchSparseLibraryDevice(sl, 0, n, s2.data_ptr())
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(args.reps):
            a2.searchSparseLibraryDevice(sl, 0, n, s2.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / args.reps
        print(json.dumps({"sweep": item, "scan_ms": round(ms2, 4), "comparisons_per_s": round(args.masks * n / (ms2 * 1e-3)),
                          "same": bool(torch.equal(s2, scores))}), flush=True)


if __name__ == "__main__":
    main()
