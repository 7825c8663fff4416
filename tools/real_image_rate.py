#!/usr/bin/env python3
"""Scan rate on REAL images: the reference's own test MIPs (tests/golden/cds_fixtures.npz: 4 LM images, 3 EM masks),
each LM image replicated --copies times into a resident library, scanned with each EM mask over the sparse and the dense
layout.  Real LM MIPs carry far more signal pixels than the benchmark's synthetic targets, so the sparse scan examines
more records per target; this prints records per target and both rates (scores checked equal)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from colormipsearch_amd import cds  # noqa: E402

W, H = 1210, 566


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--mask-thr", type=int, default=20)
    ap.add_argument("--data-thr", type=int, default=20)
    ap.add_argument("--ztol", type=float, default=0.01)
    ap.add_argument("--xy", type=int, default=2)
    ap.add_argument("--no-mirror", action="store_true")
    ap.add_argument("--min-threshold", type=int, default=0, help="minimum data threshold of the sparse library (pixels at or below it are not stored)")
    ap.add_argument("--mask", default="", help="only this EM mask (e.g. em_12191)")
    ap.add_argument("--sparse-only", action="store_true", help="time the sparse scan only (A/B builds: CDS_HIP_LIBRARY=...)")
    args = ap.parse_args()
    fx = np.load(os.path.join(ROOT, "tests", "golden", "cds_fixtures.npz"))
    lms = [k for k in fx.keys() if k.startswith("lm_")]
    ems = [k for k in fx.keys() if k.startswith("em_") and (not args.mask or k == args.mask)]
    ctx = cds.Context(0)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    n = len(lms) * args.copies
    dense = cds.Library(W, H, n, context=ctx)
    for i, k in enumerate(lms):          # target j = image j % 4 (interleaved, as a mixed library would be)
        pass
    imgs = np.stack([fx[k] for k in lms])
    for c in range(args.copies):
        dense.upload(c * len(lms), imgs)
    sig = [(int((fx[k].max(axis=2) > 0).sum())) for k in lms]
    sl = cds.SparseLibrary(W, H, n, int(sum(sig) * args.copies * 1.05) + 8 * n * H, context=ctx, min_threshold=args.min_threshold)
    for first in range(0, n, 4096):
        sl.append_from(dense, first, min(4096, n - first))
    rects = cds.label_regions(W)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    print(json.dumps({"targets": n, "images": lms, "nonblack_pixels": dict(zip(lms, sig)), "records_per_target": sl.pixels / n,
                      "sparse_bytes_per_target": 4 * sl.pixels / n}))
    for mk in ems:
        alg = cds.PixelMatchColorDepthSearchAlgorithm(fx[mk], args.mask_thr, not args.no_mirror, None, 0, False, args.data_thr, args.ztol, args.xy, rects, context=ctx)
        s_sp = torch.empty(n, dtype=torch.int32, device="cuda")
        s_de = torch.empty(n, dtype=torch.int32, device="cuda")
        out = {"mask": mk, "mask_pixels": int(alg.getQuerySize()) if hasattr(alg, "getQuerySize") else None, "layout": alg.sparseLayout(0)}
        for name, fn, buf in (("sparse", lambda b: alg.searchSparseLibraryDevice(sl, 0, n, b.data_ptr()), s_sp),
                              ("dense", lambda b: alg.searchLibraryDevice(dense, 0, n, b.data_ptr()), s_de)):
            if args.sparse_only and name == "dense":
                continue
            fn(buf)
            torch.cuda.synchronize()
            e0.record(stream)
            for _ in range(args.reps):
                fn(buf)
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            out[name + "_ms"] = round(ms, 3)
            out[name + "_comparisons_per_s"] = round(n / (ms * 1e-3))
        out["equal"] = bool(torch.equal(s_sp, s_de))
        out["scores_first_images"] = [int(v) & 0x7FFFFFFF for v in s_sp[: len(lms)].tolist()]
        out["groups"] = alg.sparseGroups()
        lay = out.pop("layout")
        out["interval_records"] = lay.get("interval_records") if isinstance(lay, dict) else None
        print(json.dumps(out), flush=True)
    if args.sparse_only:
        return
    # the three masks (and shifted / flipped variants of them: 8 masks = one group) in ONE mask set: the multi-mask scan
    # against one launch per mask
    em = [fx[k] for k in ems]
    masks8 = np.stack(em + [np.roll(em[0], 6, axis=1), np.roll(em[1], -8, axis=0), np.fliplr(em[2]).copy(), np.roll(em[0], 12, axis=0), np.roll(em[1], 10, axis=1)])
    alg = cds.PixelMatchColorDepthSearchAlgorithm(masks8, 20, True, None, 0, False, 20, 0.01, 2, rects, context=ctx)
    res = {}
    for name, env in (("multi", None), ("per_mask", "0")):
        if env is None:
            os.environ.pop("CDS_SPARSE_MULTI", None)
        else:
            os.environ["CDS_SPARSE_MULTI"] = env
        buf = torch.empty(8 * n, dtype=torch.int32, device="cuda")
        alg.searchSparseLibraryDevice(sl, 0, n, buf.data_ptr())
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(2):
            alg.searchSparseLibraryDevice(sl, 0, n, buf.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 2
        res[name] = buf
        print(json.dumps({"8_masks": name, "groups": alg.sparseGroups(), "ms": round(ms, 3), "comparisons_per_s": round(8 * n / (ms * 1e-3))}), flush=True)
    os.environ.pop("CDS_SPARSE_MULTI", None)
    print(json.dumps({"multi_equals_per_mask": bool(torch.equal(res["multi"], res["per_mask"]))}))


if __name__ == "__main__":
    main()
