#!/usr/bin/env python3
"""The overlapping-synthetic leg of bench.py (aux.overlapping_synthetic) as a command of its own, for profiles: builds the library of
synthetic targets that OVERLAP the benchmark mask (bench.overlapping_leg: 38 % of the visited records reach an evaluation round) and
scans it with the benchmark mask -- the pilot's choice, then each kernel forced.  tools/profile_round.sh runs it under rocprofv3
(`overlap` command: class overlapping_synthetic = the dispatches of cds::sparse_scan_kernel, all over --targets targets).
    python tools/overlap_rate.py [--targets 262144]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from colormipsearch_amd import cds  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--targets", type=int, default=262144)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    args = argparse.Namespace(targets=4 * a.targets, no_cpu_baseline=a.no_cpu_baseline)
    dev = torch.device("cuda", 0)
    ctx = cds.Context(0)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    mlib = cds.Library(bench.W, bench.H, 1, context=ctx)
    mlib.generate_synthetic(bench.MASK_SEED, 0, 0, 1, kind=2)
    mask = mlib.download(0, 1)[0]
    failed = []
    out = bench.overlapping_leg(args, cds, ctx, dev, stream, mask, cds.label_regions(bench.W), failed, n_o=a.targets)
    out["failed"] = failed
    print(json.dumps(out))
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
