#!/usr/bin/env python3
"""bench.py -- mask x target-MIP comparisons/sec of the pixel-match scan on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): 1 mask x 1M synthetic 1210x566 RGB targets, xyShift 2, mirror on,
production thresholds (maskThreshold 20, dataThreshold 20, pixColorFluctuation 1 -> zTolerance 0.01, label
regions excluded).  1M full frames are 2.06 TB and do not fit one GPU's 288 GB, so a STEP is one pass of the
scan kernel over one HBM-resident chunk of `--chunk` targets (default 65,536 = 135 GB; 16 steps = 1,048,576
comparisons).  The chunk is generated on the device before the timed region (inputs resident in HBM); every
step re-scans it (135 GB >> 256 MiB Infinity Cache, so no step is served from cache).

Multi-GPU (--gpus N, launched by torch.distributed.run): the library is sharded by target index, every rank
scans its own resident chunk (weak scaling), and each step ends with ONE RCCL gather of the per-target score
array (int32, score | mirrored<<31) to rank 0 -- the only collective on the path.

Prints one JSON line (rank 0): the contract fields plus `roofline` (scan kernel, HBM-bound) and
`cpu_baseline` (the C oracle = restatement of the reference's Java loops, timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1210, 566
LIB_SEED, MASK_SEED = 0xC0FFEE, 0xBADC0DE
MASK_THR, DATA_THR, ZTOL, XY_SHIFT, MIRROR = 20, 20, 0.01, 2, True
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def usable_cores():
    """Host cores this process may really use: affinity mask and cgroup CPU quota, not the machine total."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunk", type=int, default=65536, help="resident targets per GPU = comparisons per step per GPU")
    ap.add_argument("--kind", type=int, default=0, help="0 sparse-lut targets (default), 1 dense-uniform")
    ap.add_argument("--cpu-sample", type=int, default=0, help="targets in the cpu_baseline sample (0 = 8000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the secondary (batched-mask) measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get("CDS_BENCH_FORCE_DIST") == "1"   # (rehearsal of the N>1 code path on one GPU)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    if not os.path.exists(os.path.join(ROOT, "colormipsearch_amd", "libcds_hip.so")) and rank == 0:
        import __graft_entry__
        __graft_entry__.build()   # (normally built beforehand; hipcc is in the image)
    if world > 1:
        dist.barrier()
    from colormipsearch_amd import cds

    ctx = cds.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    # ---- inputs, resident in HBM before the timed region ----
    chunk = args.chunk
    nbytes = cds.Library.bytes_required(W, H, chunk)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)     # torch owns the HBM (plumbing)
    buf.zero_()
    lib = cds.Library(W, H, chunk, context=ctx, device_buffer=buf)
    t0 = time.time()
    lib.generate_synthetic(LIB_SEED, rank * chunk, 0, chunk, kind=args.kind)   # this rank's shard of the library
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    mlib = cds.Library(W, H, 1, context=ctx)
    mlib.generate_synthetic(MASK_SEED, 0, 0, 1, kind=2)
    mask = mlib.download(0, 1)[0]
    rects = cds.label_regions(W)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, MASK_THR, MIRROR, None, 0, False, DATA_THR, ZTOL, XY_SHIFT, rects, context=ctx)
    P = alg.getQuerySize()
    ys = np.flatnonzero((mask.max(axis=2) > MASK_THR).any(axis=1))
    rows = min(H, int(ys.max() - ys.min() + 1) + 2 * XY_SHIFT)
    alg_bytes_per_target = rows * 3 * W          # R x 3630 B

    scores = torch.empty(chunk, dtype=torch.int32, device=dev)
    gathered = [torch.empty(chunk, dtype=torch.int32, device=dev) for _ in range(world)] if (use_dist and rank == 0) else None

    def step():
        alg.searchLibraryDevice(lib, 0, chunk, scores.data_ptr())
        if use_dist:
            dist.gather(scores, gathered, dst=0)

    for _ in range(args.warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        alg.searchLibraryDevice(lib, 0, chunk, scores.data_ptr())
        ev[k][1].record(stream)
        if use_dist:
            dist.gather(scores, gathered, dst=0)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))   # scan kernel only (one launch per step)

    host_scores = scores.cpu().numpy()
    sc, mir = host_scores & 0x7FFFFFFF, (host_scores >> 31) & 1

    out = None
    if rank == 0:
        total = world * chunk * args.steps
        achieved = alg_bytes_per_target * chunk / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "scan_traffic.json")   # PMC-derived HBM bytes per launch, if collected
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("chunk") == chunk and tj.get("kind") == args.kind:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "mask x target MIP comparisons/sec at 1210x566 RGB",
            "value": total / elapsed,
            "unit": "comparisons/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "1 mask x 1M synthetic 1210x566 RGB targets, xyShift=2, mirror on (BASELINE configs[1]); "
                            f"step = scan of one HBM-resident chunk of {chunk} targets per GPU",
                "targets_per_step_per_gpu": chunk,
                "target_kind": "sparse-lut" if args.kind == 0 else "dense-uniform",
                "mask_pixels": P,
                "mask_rows_touched": rows,
                "params": {"maskThreshold": MASK_THR, "dataThreshold": DATA_THR, "zTolerance": ZTOL, "xyShift": XY_SHIFT, "mirror": MIRROR},
                "parallelism": f"target-shard x{world}" + (", 1 RCCL gather of int32 scores per step" if world > 1 else ""),
                "library_generation_s": round(gen_s, 3),
                "nonzero_scores_in_chunk": int((sc > 0).sum()),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "cds::scan_kernel<xy2,mirror,single-mask>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_target": alg_bytes_per_target,
                "kernel_ms_per_launch": kern_ms,
                # the kernel fetches only the column window of each row, so real traffic < algorithmic bytes
                "traffic_GBps": (traffic / (kern_ms * 1e-3) / 1e9) if traffic else None,
                "note": "achieved = algorithmic bytes (mask rows x 3630 B per target, SURVEY 8d) / kernel time; the kernel "
                        "fetches only the mask's column window of those rows, so the PMC traffic is lower and frac may exceed 1",
            },
        }

    # ---- CPU baseline: the C oracle (a restatement of the reference's Java loops), bounded sample ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cds_oracle as O
        cores = usable_cores()
        # bounded sample: the first n_s targets of the chunk (16.4 GB of host memory at 8000), scanned repeatedly until
        # ~1 s of wall time = ~cores x 1 s of CPU work has been measured
        n_s = min(args.cpu_sample or 8000, chunk)   # 80 partitions of 100
        sample = lib.download(0, n_s)
        q = O.PixelMatchQuery(mask, MASK_THR, DATA_THR, ZTOL, XY_SHIFT, MIRROR, rects)
        q.scan(sample[: min(64, n_s)], n_threads=cores, partition=4)   # warm the thread pool / page in
        passes, dt = 0, 0.0
        while dt < 1.0 and passes < 64:
            t0 = time.perf_counter()
            ws, wm = q.scan(sample, n_threads=cores, partition=100)   # partitions of 100 like LocalColorMIPSearchProcessor
            dt += time.perf_counter() - t0
            passes += 1
        dt /= passes
        n1 = max(16, min(n_s, int(n_s / max(cores, 1) / 4)))      # single-thread leg on a slice of the sample
        t0 = time.perf_counter()
        q.scan(sample[:n1], n_threads=1, partition=100)
        dt1 = time.perf_counter() - t0
        try:
            cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            cpu_model = "unknown CPU"
        ok = bool(np.array_equal(ws, sc[:n_s]) and np.array_equal(wm.astype(np.int64), mir[:n_s].astype(np.int64)))
        out["cpu_baseline"] = {
            "value": n_s / dt,
            "unit": "comparisons/s",
            "cores": cores,
            "kind": "port",
            "sample": f"first {n_s} targets of the same chunk, same mask and parameters, C oracle (-O2, f64 predicate, "
                      f"18 gather passes per pair) on {cores} threads of {cpu_model}, target partitions of 100: {passes} passes of "
                      f"{dt:.2f} s wall = {passes * dt * cores:.0f} core-seconds; scores equal to the GPU's: {ok}",
            "single_thread_value": n1 / dt1,
        }
        if not ok:
            out["cpu_baseline"]["parity_failure"] = True
    # ---- secondary measurement (not `value`): BASELINE configs[2] in miniature -- 128 masks share one pass over the first
    # 16,384 resident targets (the production shape: many masks against one library)
    if rank == 0 and world == 1 and not args.no_aux and chunk >= 16384:
        n_m, n_t = 128, 16384
        mlib2 = cds.Library(W, H, n_m, context=ctx)
        mlib2.generate_synthetic(MASK_SEED, 0, 0, n_m, kind=2)
        batch = cds.PixelMatchColorDepthSearchAlgorithm(mlib2.download(0, n_m), MASK_THR, MIRROR, None, 0, False, DATA_THR, ZTOL,
                                                        XY_SHIFT, rects, context=ctx)
        bscores = torch.empty(n_m * n_t, dtype=torch.int32, device=dev)
        batch.searchLibraryDevice(lib, 0, n_t, bscores.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(2):
            batch.searchLibraryDevice(lib, 0, n_t, bscores.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        bms = e0.elapsed_time(e1) / 2
        same = bool(torch.equal(bscores[:n_t], scores[:n_t]))   # mask 0 of the batch is the benchmark mask
        out["aux"] = {"workload": f"{n_m} masks x {n_t} resident targets in one batched pass (BASELINE configs[2] shape)",
                      "comparisons_per_s": n_m * n_t / (bms * 1e-3), "ms_per_pass": bms,
                      "mask0_scores_equal_single_mask_scan": same}
    failed = []
    if rank == 0:
        if out.get("cpu_baseline", {}).get("parity_failure"):
            failed.append("GPU scores differ from the CPU oracle's on the sampled targets")
        if out.get("aux", {}).get("mask0_scores_equal_single_mask_scan") is False:
            failed.append("batched-mask scan disagrees with the single-mask scan")
        if failed:
            out["value"] = None     # a wrong result is not a measurement
            out["parity_failure"] = failed
        print(json.dumps(out))
    if use_dist:
        if rank == 0 and not torch.equal(gathered[0], scores):
            raise SystemExit("gathered scores differ from the local ones")
        dist.destroy_process_group()
    if failed:
        raise SystemExit("bench.py: parity failure: " + "; ".join(failed))


if __name__ == "__main__":
    main()
