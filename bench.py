#!/usr/bin/env python3
"""bench.py -- mask x target-MIP comparisons/sec of the pixel-match scan on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): 1 mask x 1M synthetic 1210x566 RGB targets, xyShift 2, mirror on, production
thresholds (maskThreshold 20, dataThreshold 20, pixColorFluctuation 1 -> zTolerance 0.01, label regions excluded).
The whole library (`--targets`, default 1,048,576 per GPU) is RESIDENT in HBM as a sparse library (signal-pixel records,
~82 KB per target instead of 2.06 MB of packed RGB: DESIGN.md section 3); it is generated on the device, through dense
staging slots, before the timed region.  A STEP is one scan of the whole resident library with the benchmark mask =
`--targets` comparisons per GPU (the library is far larger than the 256 MiB Infinity Cache, so no step is served from cache).

Multi-GPU (`--gpus N`): started directly, this script launches its N ranks itself (torch.distributed.run, before anything
touches a GPU); under torch.distributed.run it is one of the ranks.  The library is sharded by target index, every rank
scans its own resident shard (weak scaling: N x 1M targets = configs[3] at N = 8), and each step ends with ONE RCCL gather
of the per-target score array (int32, score | mirrored<<31) to rank 0 -- the only collective on the path.

EVERY score of the last timed step is verified: each rank regenerates its shard window by window as dense images and
compares the sparse scan with the dense-layout kernel (a second, independent implementation), and three 256-target slices
(first window, a window whose records lie beyond 2^32 in the arena, last window) with the CPU oracle.  A flipped bit
anywhere makes the run fail (value null, exit code 1).

Prints one JSON line (rank 0): the contract fields plus `roofline` (instruction-issue bound with the HBM fraction by
measured traffic beside it), `cpu_baseline` (the C oracle = restatement of the reference's Java loops, timed on the host
cores) and `aux` (dense layout; BASELINE configs[2] at full size; GradientAreaGap scores of top hits; the reference's JUnit
pairs; the scan rate on the reference's real test images).
"""
import argparse
import hashlib
import json
import os
import subprocess
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
HBM_COPY_GBS = 6290.0  # the same guide: 6.29 TB/s measured for a float4 copy (79 % of the spec), "~6.3 TB/s achievable"
PMC_FILE = os.path.join(ROOT, "profiles", "scan_pmc.json")   # written by tools/summarize_profile.py from rocprofv3 --pmc passes


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


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as children (nothing in this process
    has touched a GPU yet, and it never will) and leave with their exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def load_pmc():
    """Per-target / per-pair hardware-counter figures of the kernels, collected by separate rocprofv3 --pmc passes of this
    script (tools/profile_round.sh) and keyed by the SHA-256 of the kernel sources they were measured on."""
    try:
        j = json.load(open(PMC_FILE))
    except Exception:
        return None
    stale = []
    for rel, want in j.get("sources", {}).items():
        try:
            got = hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()
        except Exception:
            got = None
        if got != want:
            stale.append(rel)
    j["stale_sources"] = stale
    return j


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--targets", type=int, default=1 << 20, help="resident targets per GPU = comparisons per step per GPU")
    ap.add_argument("--stage", type=int, default=8192, help="dense staging slots used while the library is generated / verified")
    ap.add_argument("--kind", type=int, default=0, help="0 sparse-lut targets (default), 1 dense-uniform")
    ap.add_argument("--cpu-sample", type=int, default=0, help="targets in the cpu_baseline sample (0 = 8000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--batch-masks", type=int, default=1024, help="masks of the configs[2] leg (aux.mask_batch)")
    ap.add_argument("--dist-grad-masks", type=int, default=64, help="masks of the sharded gradient-score leg (N > 1 or CDS_BENCH_FORCE_DIST=1)")
    ap.add_argument("--verify-windows", type=int, default=0, help="staging-sized windows of the library compared with the dense scan "
                    "(0 = all of them: every score of the run is checked; k >= 3 = k windows spread over the index range)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.gpus > torch.cuda.device_count():      # (device_count() does not initialise the GPU on this image)
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} visible")
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE {world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly ONE line, the JSON: whatever libraries print there (RCCL's version banner) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
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
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count

    # ---- inputs, resident in HBM before the timed region ----
    n = args.targets
    if args.kind == 1:
        n = min(n, 32768)   # dense-uniform images are 2.7 MB each as records (the sparse layout's worst case): a smaller resident library
    stage_n = min(args.stage, n)
    dense = cds.Library(W, H, stage_n, context=ctx)             # dense staging slots (packed RGB rows)
    cap_px = (W * H + 4 * H) * n if args.kind == 1 else 24000 * n + 4_000_000
    slib = cds.SparseLibrary(W, H, n, cap_px, context=ctx, density_limit=2.0 if args.kind == 1 else None)      # the resident library
    t0 = time.time()
    conv_s = 0.0                                                # the library CONVERSION alone (dense rows -> records: the count / scan / fill
    for first in range(0, n, stage_n):                          # kernels of cds_sparse_library_append), apart from the synthesis of the images
        c = min(stage_n, n - first)                             # this rank's shard: library indices rank*n + [0, n)
        dense.generate_synthetic(LIB_SEED, rank * n + first, 0, c, kind=args.kind)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        slib.append_from(dense, 0, c)
        torch.cuda.synchronize()
        conv_s += time.perf_counter() - t1
    gen_s = time.time() - t0
    has_stream16 = os.environ.get("CDS_SPARSE_STREAM16", "1") != "0"
    lib_bytes = (6 if has_stream16 else 4) * slib.pixels + n * (4 * slib.n_strips * (H + 1) + 8)   # records (+ their 16-bit stage-1 stream) + row pointers
    mlib = cds.Library(W, H, 1, context=ctx)
    mlib.generate_synthetic(MASK_SEED, 0, 0, 1, kind=2)
    mask = mlib.download(0, 1)[0]
    rects = cds.label_regions(W)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, MASK_THR, MIRROR, None, 0, False, DATA_THR, ZTOL, XY_SHIFT, rects, context=ctx)
    # SURVEY 8(d)(i) counts the masks' preparation and upload with a search: the mask set of the benchmark mask is made three more
    # times on the clock (host: positions, interval tables, plane, band map; then the H2D copies) -- reported beside `value`
    prep_ms = []
    for _ in range(3):
        t1 = time.perf_counter()
        a_tmp = cds.PixelMatchColorDepthSearchAlgorithm(mask, MASK_THR, MIRROR, None, 0, False, DATA_THR, ZTOL, XY_SHIFT, rects, context=ctx)
        ctx.synchronize()
        prep_ms.append((time.perf_counter() - t1) * 1e3)
        a_tmp.close()
    mask_prepare_ms = float(np.median(prep_ms))
    P = alg.getQuerySize()
    ys = np.flatnonzero((mask.max(axis=2) > MASK_THR).any(axis=1))
    rows = min(H, int(ys.max() - ys.min() + 1) + 2 * XY_SHIFT)
    dense_bytes_per_target = rows * 3 * W        # R x 3630 B -- what a DENSE layout would stream

    # Two score buffers: the gather of step k (RCCL's own stream, waits for scan k) runs while step k + 1 scans into the
    # other buffer -- as consecutive masks of a search would; a buffer is reused only after its gather has completed.
    sbuf = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
    gathered = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(world)] if (use_dist and rank == 0) else None
    pending = [None, None]

    def one_step(k, e0=None, e1=None):
        b = k & 1
        if pending[b] is not None:
            pending[b].wait()          # (stream-level: the scan below is ordered after the gather that read this buffer)
            pending[b] = None
        if e0 is not None:
            e0.record(stream)
        alg.searchSparseLibraryDevice(slib, 0, n, sbuf[b].data_ptr())
        if e1 is not None:
            e1.record(stream)
        if use_dist:
            pending[b] = dist.gather(sbuf[b], gathered, dst=0, async_op=True)

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for k in range(args.warmup):
        one_step(k)
    drain()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(k, ev[k][0], ev[k][1])
    drain()                            # every step's scores are on rank 0 before the clock stops
    scores = sbuf[(args.steps - 1) & 1]
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms = [a.elapsed_time(b) for a, b in ev]
    kern_ms = float(np.mean(step_ms))   # the scan of one step (one kernel launch), on the launch stream
    rank_kern_ms = [kern_ms]
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        km = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(km, torch.tensor([kern_ms], dtype=torch.float64, device=dev))
        rank_kern_ms = [float(x.item()) for x in km]
        if rank == 0 and not torch.equal(gathered[0], scores):
            raise SystemExit("gathered scores differ from the local ones")

    host_scores = scores.cpu().numpy()
    sc, mir = host_scores & 0x7FFFFFFF, (host_scores >> 31) & 1
    failed = []
    # the scan's memory floor, measured now: the same launch with the kernel's loads-only twin (same work distribution, run headers,
    # record loads and prefetch distance; no examination of the records) into a scratch buffer
    load_only_ms = None
    if args.kind == 0:
        scratch = torch.empty(n, dtype=torch.int32, device=dev)
        try:
            ctx.sparse_load_only(True)
            alg.searchSparseLibraryDevice(slib, 0, n, scratch.data_ptr())
            lo_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for a, b in lo_ev:
                a.record(stream)
                alg.searchSparseLibraryDevice(slib, 0, n, scratch.data_ptr())
                b.record(stream)
            torch.cuda.synchronize()
            load_only_ms = float(np.mean([a.elapsed_time(b) for a, b in lo_ev]))
        except cds.CdsError:
            load_only_ms = None       # (a layout without a loads-only twin)
        finally:
            ctx.sparse_load_only(False)
        del scratch
    stream_gbs = None
    try:
        stream_gbs, _ = slib.stream_read_gbs()     # a plain streaming read of the whole record arena on THIS device (fastest of three)
    except cds.CdsError:
        pass
    native_comm = None
    if use_dist and rank == 0:
        # the same gather behind the C ABI (cds_comm over this rank's context: ncclCommInitAll + ncclGather through libcds_hip.so,
        # what a JVM host with all GPUs in one process calls): its scores must be the ones torch.distributed gathered
        comm = cds.Comm([ctx])
        ns, nm, _ = comm.searchSparseLibrarySharded([alg], [slib])
        ok = bool(np.array_equal(ns[0], sc) and np.array_equal(nm[0].astype(np.int64), mir.astype(np.int64)))
        native_comm = {"entry": "cds_search_sparse_sharded", "contexts": len(comm), "rccl": comm.uses_rccl, "scores_equal": ok}
        comm.close()
        if not ok:
            failed.append("cds_search_sparse_sharded disagrees with the local scan")

    # ---- verification of the WHOLE shard: sparse scan == dense-layout scan, window by window; oracle on three slices ----
    offs = slib.image_offsets(0, n)                               # record offsets in the arena (uint64)
    first_beyond = int(np.searchsorted(offs[:-1], np.uint64(1) << np.uint64(32)))   # first image stored beyond 2^32 records
    starts = list(range(0, n, stage_n))
    if args.verify_windows >= 3 and args.verify_windows < len(starts):
        pick = sorted({int(round(i * (len(starts) - 1) / (args.verify_windows - 1))) for i in range(args.verify_windows)})
        beyond = [i for i, s0 in enumerate(starts) if s0 >= first_beyond]
        if beyond and not any(starts[i] >= first_beyond for i in pick):
            pick.append(beyond[len(beyond) // 2])
        starts = [starts[i] for i in sorted(set(pick))]
    dsc = torch.empty(stage_n, dtype=torch.int32, device=dev)
    mism = torch.zeros((), dtype=torch.int64, device=dev)
    t0 = time.time()
    slice_at = {0: "first window", starts[-1]: "last window"}
    mid = [s0 for s0 in starts if s0 >= first_beyond and s0 != starts[-1]]
    if mid:
        slice_at[mid[len(mid) // 2]] = "window beyond 2^32 records"
    oracle_slices, q_or = [], None
    for s0 in starts:
        c = min(stage_n, n - s0)
        dense.generate_synthetic(LIB_SEED, rank * n + s0, 0, c, kind=args.kind)
        alg.searchLibraryDevice(dense, 0, c, dsc.data_ptr())
        mism += (dsc[:c] != scores[s0:s0 + c]).sum()
        if s0 in slice_at and rank == 0 and not args.no_cpu_baseline:
            from oracle import cds_oracle as O
            if q_or is None:
                q_or = O.PixelMatchQuery(mask, MASK_THR, DATA_THR, ZTOL, XY_SHIFT, MIRROR, rects)
            k0 = max(0, c - 256) if s0 == starts[-1] else 0          # (the last window: its last 256 targets)
            k1 = min(c, k0 + 256)
            imgs = dense.download(k0, k1 - k0)
            ws, wm = q_or.scan(imgs, n_threads=usable_cores(), partition=16)
            ok = bool(np.array_equal(ws, sc[s0 + k0:s0 + k1]) and np.array_equal(wm.astype(np.int64), mir[s0 + k0:s0 + k1].astype(np.int64)))
            oracle_slices.append({"targets": [s0 + k0, s0 + k1], "what": slice_at[s0], "first_record_offset": int(offs[s0 + k0]),
                                  "equal_to_oracle": ok})
            if not ok:
                failed.append(f"sparse scan differs from the CPU oracle on targets [{s0 + k0}, {s0 + k1})")
            del imgs
    n_mism = int(mism.item())
    verified = sum(min(stage_n, n - s0) for s0 in starts)
    if use_dist:
        tm = torch.tensor([n_mism, verified], dtype=torch.int64, device=dev)
        dist.all_reduce(tm)
        n_mism, verified = int(tm[0].item()), int(tm[1].item())
    if n_mism:
        failed.append(f"sparse scan and dense-layout scan disagree on {n_mism} of the {verified} verified targets")
    verification = {"targets_verified_against_dense_scan": verified, "of": world * n, "windows_per_gpu": len(starts), "mismatches": n_mism,
                    "first_target_beyond_2^32_records": first_beyond if first_beyond < n else None,
                    "targets_beyond_2^32_records_verified_per_gpu": sum(min(stage_n, n - s0) for s0 in starts if s0 >= first_beyond),
                    "oracle_slices": oracle_slices, "seconds": round(time.time() - t0, 2)}

    grad_leg = None
    if use_dist and args.kind == 0 and not args.no_aux:
        grad_leg = dist_gradient_leg(args, cds, dist, ctx, dev, dense, slib, n, stage_n, rects, rank, world, failed)

    out = None
    if rank == 0:
        total = world * n * args.steps
        pmc = load_pmc()
        issue_ns = ctx.valu_issue_ns()                            # measured now, on this device
        simds = 4 * n_cus
        # The bound is HBM: the scan's loads-only twin takes 0.8-0.9 of the scan's time, i.e. the kernel waits for its records
        # (DESIGN.md 4b).  achieved = MEASURED HBM traffic per launch (PMC: 2 x FETCH_SIZE per target x targets) / kernel time of
        # this run; the instruction-issue fraction and the loads-only twin are secondary fields.
        scan_kernel, pilot_ratio = alg.lastScanKernel()
        roof = {"bound": "hbm", "kernel": "cds::sparse_scan16_kernel<xy2,mirror> (16-bit stage-1 stream, chosen by the pilot)" if scan_kernel == 1
                else "cds::sparse_scan_kernel<xy2,mirror> (4-byte records)", "achieved": None, "peak": HBM_PEAK_GBS,
                "pilot": {"share_of_visited_records_reaching_a_round": pilot_ratio, "kernel_chosen": "stream16" if scan_kernel == 1 else "records32",
                          "what": "the first search measures this share on its first 4,096 targets: <= 0.05 -> the 16-bit stream kernel "
                                  "(CDS_SPARSE_SCAN16=1 / CDS_SPARSE_SCAN32=1 force one)"},
                "unit": "GB/s", "frac": None, "traffic": None,
                "kernel_ms_per_launch": kern_ms, "kernel_ms_min_median_max": [float(np.min(step_ms)), float(np.median(step_ms)), float(np.max(step_ms))],
                "targets_per_launch": n,
                "measured_copy_rate": HBM_COPY_GBS, "frac_of_measured_copy_rate": None,
                "stream_read_of_the_record_arena_in_this_run": stream_gbs, "frac_of_stream_read_in_this_run": None,
                "issue": {"what": "secondary: VALU wave-instructions per launch (PMC) x the issue time of one wave64 VALU instruction per SIMD "
                                  "measured in this run / (SIMDs x kernel time)", "achieved": None, "peak": simds / issue_ns,
                          "unit": "G wave64-VALU-instructions/s", "frac": None, "valu_issue_ns_per_simd": issue_ns, "simds": simds},
                "load_only": None if load_only_ms is None else {
                    "what": "the same launch with the scan kernel's loads-only twin (cds_debug_sparse_load_only: same grabs, run headers, record "
                            "loads and prefetch distance, no examination of the records), mean of 5 launches in this run: what the memory "
                            "system delivers for this access pattern",
                    "kernel_ms_per_launch": load_only_ms, "scan_time_over_load_only_time": kern_ms / load_only_ms, "hbm_gbs": None,
                    "frac_of_stream_read_in_this_run": None},
                "dense_layout_bytes_per_target": dense_bytes_per_target,
                "note": "bound = hbm by the task's convention: frac = measured HBM traffic of the launch (PMC counters of profiles/scan_pmc.json, "
                        "per target, keyed by the kernel sources' hash; see traffic_source for how FETCH_SIZE becomes bytes) / kernel time of this run / the 8 TB/s spec.  The kernel "
                        "reads a record library (signal pixels only), not packed-RGB rows: SURVEY 8(d)'s byte model "
                        "(dense_layout_bytes_per_target) describes aux.dense_layout, not this kernel.  Round 4 halved the streamed bytes (the "
                        "16-bit stage-1 stream), so the fraction of the HBM peak FELL while `value` rose: the stream kernel is bound by "
                        "instruction issue (roofline.issue: vector and scalar pipes), its loads-only twin (load_only) takes about half its time.  "
                        "stream_read = a plain read of the 4-byte record arena in this run."}
        cls = (pmc or {}).get("classes", {}).get("sparse_scan_headline") if args.kind == 0 else None
        if cls and {"SQ_INSTS_VALU", "hbm_bytes"} <= set(cls.get("per_target", {})):
            pt = cls["per_target"]
            roof["traffic"] = pt["hbm_bytes"] * n
            # (round 5, VERDICT r4 #7: FETCH_SIZE counts 64 B per request whatever its size; the profile now carries the request-size counters,
            # and every read request of this kernel -- the stream's AND the rounds' gathers -- is 128 B wide: the doubled figure is the exact one)
            roof["traffic_source"] = {"hbm_correction": pt.get("hbm_correction", "x2 (profile older than round 5)"),
                                      "bytes_per_target": pt["hbm_bytes"], "face_value": pt.get("hbm_bytes_face_value"), "doubled": pt.get("hbm_bytes_doubled"),
                                      "what": "`request sizes` = 32 B x TCC_EA0_RDREQ_32B + 64 B x _64B + 128 B x _128B per target, measured in separate --pmc passes "
                                              "(FETCH_SIZE x 1024 B = 64 B per request: half the bytes when the requests are 128 B wide -- which all of this kernel's "
                                              "are, gathers included: default-policy loads make L2 fetch whole 128-B lines; tools/fetchcal, "
                                              "profiles/r05_fetch_size_calibration.txt)"}
            roof["achieved"] = roof["traffic"] / (kern_ms * 1e-3) / 1e9
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
            roof["frac_of_measured_copy_rate"] = roof["achieved"] / HBM_COPY_GBS
            if stream_gbs:
                roof["frac_of_stream_read_in_this_run"] = roof["achieved"] / stream_gbs
            roof["issue"]["achieved"] = pt["SQ_INSTS_VALU"] * n / (kern_ms * 1e6)     # G wave-instructions / s
            roof["issue"]["frac"] = roof["issue"]["achieved"] / roof["issue"]["peak"]
            if {"SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"} <= set(pt):
                # how busy the vector pipes were in the PROFILED launches, from the counters alone: SQ_ACTIVE_INST_VALU counts quad-cycles
                # summed over the SIMDs, GRBM_GUI_ACTIVE the launch's cycles summed over the 8 XCDs
                roof["issue"]["valu_pipe_active_frac_in_the_profile"] = pt["SQ_ACTIVE_INST_VALU"] * 4.0 / (simds * pt["GRBM_GUI_ACTIVE"] / 8.0)
                roof["issue"]["valu_note"] = ("this kernel's instruction mix (three-operand and packed encodings, v_mbcnt, compares into scalar registers) occupies a "
                                              "SIMD for about 4 cycles per wave64 instruction, against the ~2.5 of the plain v_add / v_xor chains that `peak` "
                                              "is calibrated with: by pipe-active cycles the vector pipes are nearly saturated although `frac` says half")
            if roof["load_only"]:
                roof["load_only"]["hbm_gbs"] = roof["traffic"] / (load_only_ms * 1e-3) / 1e9
                if stream_gbs:
                    roof["load_only"]["frac_of_stream_read_in_this_run"] = roof["load_only"]["hbm_gbs"] / stream_gbs
            roof["counters"] = {"source": f"profiles/scan_pmc.json (tag {pmc.get('tag')}): separate rocprofv3 --pmc passes of "
                                          f"`{pmc.get('command')}`, mean of {cls.get('dispatches')} launches of {cls.get('targets_per_dispatch')} targets",
                                "measured_in_this_run": False, "stale_sources": pmc["stale_sources"],
                                "per_target": pt, "profile_kernel_ms_per_launch": cls.get("mean_ms")}
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
                            f"step = scan of the whole HBM-resident library of {n} targets per GPU (sparse layout)",
                "targets_per_step_per_gpu": n,
                "library_layout": "sparse (signal-pixel records, 4 column strips" + (" + their 16-bit stage-1 stream)" if has_stream16 else ")"),
                "library_bytes_per_gpu": lib_bytes,
                "library_bytes_per_target": round(lib_bytes / n, 1),
                "target_kind": "sparse-lut" if args.kind == 0 else "dense-uniform",
                "mask_pixels": P,
                "mask_rows_touched": rows,
                "params": {"maskThreshold": MASK_THR, "dataThreshold": DATA_THR, "zTolerance": ZTOL, "xyShift": XY_SHIFT, "mirror": MIRROR},
                "parallelism": f"target-shard x{world}" + (", 1 RCCL gather of int32 scores per step, overlapped with the next step's scan" if world > 1 else ""),
                "library_generation_s": round(gen_s, 3),
                "library_conversion_s": round(conv_s, 3),
                "nonzero_scores": int((sc > 0).sum()),
            },
            # what `value` leaves out, by name: the mask's preparation (host tables + H2D) and the
            # one-off conversion of the library into records (mask-independent; amortised over every mask searched)
            "excluded_costs": {
                "mask_prepare_ms": mask_prepare_ms,
                "mask_prepare_ms_runs": prep_ms,
                "comparisons_per_s_including_mask_preparation": n * world / ((elapsed / args.steps) + mask_prepare_ms * 1e-3),
                "library_conversion_s": conv_s,
                "library_conversion_us_per_target": conv_s / n * 1e6,
                "library_synthesis_s (benchmark scaffolding)": gen_s - conv_s,
                "what": "mask_prepare_ms = cds_maskset_create of the benchmark mask (positions, interval tables, LDS plane, band map, H2D), median of 3; "
                        "library_conversion_s = the dense-rows -> records kernels of cds_sparse_library_append over the whole shard "
                        "(count / prefix / fill), apart from the synthesis of the images; break_even_masks (aux.dense_layout) = masks per "
                        "resident library above which conversion + sparse scans beat dense scans of the same targets",
            },
            "kernel_ms_per_rank": rank_kern_ms,
            "native_comm": native_comm,
            "gradient_scores_sharded": grad_leg,
            "verification": verification,
            "roofline": roof,
        }

    # ---- CPU baseline: the C oracle (a restatement of the reference's Java loops), bounded sample ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cds_oracle as O
        cores = usable_cores()
        # bounded sample: the first n_s targets of the library as dense images (16.4 GB of host memory at 8000), scanned
        # repeatedly until ~1 s of wall time = ~cores x 1 s of CPU work has been measured
        n_s = min(args.cpu_sample or 8000, n, stage_n)   # 80 partitions of 100
        dense.generate_synthetic(LIB_SEED, 0, 0, n_s, kind=args.kind)
        sample = dense.download(0, n_s)
        q = q_or or O.PixelMatchQuery(mask, MASK_THR, DATA_THR, ZTOL, XY_SHIFT, MIRROR, rects)
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
            "sample": f"first {n_s} targets of the same library (as dense images), same mask and parameters, C oracle (-O2, f64 "
                      f"predicate, 18 gather passes per pair) on {cores} threads of {cpu_model}, target partitions of 100: {passes} "
                      f"passes of {dt:.2f} s wall = {passes * dt * cores:.0f} core-seconds; scores equal to the GPU's: {ok}",
            "single_thread_value": n1 / dt1,
        }
        if not ok:
            out["cpu_baseline"]["parity_failure"] = True
            failed.append("GPU scores differ from the CPU oracle's on the sampled targets")
        del sample

    # ---- secondary measurements (not `value`) ----
    if rank == 0 and world == 1 and not args.no_aux:
        out["aux"] = aux_legs(args, cds, ctx, dev, stream, dense, slib, alg, mask, rects, scores, n, stage_n, dense_bytes_per_target, failed)
        dl = out["aux"].get("dense_layout")
        if dl and dl.get("comparisons_per_s"):
            # conversion costs C per target once; a mask then costs ts (sparse) instead of td (dense) per target: C + m ts < m td
            td, ts = 1.0 / dl["comparisons_per_s"], kern_ms * 1e-3 / n
            out["excluded_costs"]["break_even_masks_per_resident_library_vs_dense_scan"] = (conv_s / n) / (td - ts) if td > ts else None
            out["excluded_costs"]["dense_scan_ns_per_comparison"] = td * 1e9
            out["excluded_costs"]["sparse_scan_ns_per_comparison"] = ts * 1e9
        ri = out["aux"].get("real_images", {}).get("masks")
        if ri:   # next to the headline: the same kernel on the reference's own (dense, mask-overlapping) images is 5-20x slower
            out["real_image_comparisons_per_s"] = {k: v["sparse_comparisons_per_s"] for k, v in ri.items()}
            ov = out["aux"].get("overlapping_synthetic")
            if ov:
                out["overlapping_synthetic_comparisons_per_s"] = ov["comparisons_per_s"]
            out["real_image_note"] = ("`value` is measured on BASELINE's synthetic library, whose targets' colours are unrelated to the mask's; on the reference's LM "
                                      "test MIPs -- every one chosen because it overlaps these EM masks -- the same single-mask scan runs at the rates above")
    if rank == 0:
        if failed:
            out["value"] = None     # a wrong result is not a measurement
            out["parity_failure"] = failed
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()
    if failed:
        raise SystemExit("bench.py: parity failure: " + "; ".join(failed))


def dist_gradient_leg(args, cds, dist, ctx, dev, dense, slib, n, stage_n, rects, rank, world, failed):
    """BASELINE configs[4] on N GPUs (run for world > 1, and with CDS_BENCH_FORCE_DIST=1 as a one-rank rehearsal): the top-500 hits
    of `--dist-grad-masks` masks over the WHOLE sharded library -- every rank scans its shard and selects its own top 500 per mask on
    the device, rank 0 merges the candidates into the global top 500 and broadcasts the (mask, library index) pair list -- then
    GradientAreaGap on the rank that owns each target (dist.sharded_pair_scores driving cds_gradscore: the pairs of a rank are
    visited in windows of the staging slots, regenerated with their gradient / zgap variants), results gathered to rank 0 in pair order
    (the reference: CalculateGradientScoresCmd.java:541-586 over the partitions of SparkColorMIPSearchProcessor.java:50-73)."""
    from colormipsearch_amd import dist as cdist
    g_m, top = max(1, args.dist_grad_masks), min(500, n)
    t_all = time.perf_counter()
    mlib = cds.Library(W, H, g_m, context=ctx)
    mlib.generate_synthetic(MASK_SEED, 0, 0, g_m, kind=2)
    masks = mlib.download(0, g_m)
    del mlib
    batch = cds.PixelMatchColorDepthSearchAlgorithm(masks, MASK_THR, MIRROR, None, 0, False, DATA_THR, ZTOL, XY_SHIFT, rects, context=ctx)
    bsc = torch.empty(g_m * n, dtype=torch.int32, device=dev)
    batch.searchSparseLibraryDevice(slib, 0, n, bsc.data_ptr())
    tk_t, tk_s, _ = batch.topKOfDeviceScores(bsc.data_ptr(), n, top)
    del bsc
    cand = torch.from_numpy(np.stack([tk_t.astype(np.int64) + rank * n, tk_s.astype(np.int64)], axis=2)).to(dev)   # [g_m, top, 2]: library index, score
    recv = [torch.empty_like(cand) for _ in range(world)] if rank == 0 else None
    dist.gather(cand, recv, dst=0)
    pairs = torch.zeros((g_m * top, 2), dtype=torch.int64, device=dev)
    if rank == 0:
        allc = torch.cat(recv, dim=1).cpu().numpy()                      # [g_m, world * top, 2]
        pm, pt = [], []
        for m in range(g_m):
            idx, sc = allc[m, :, 0], allc[m, :, 1]
            order = np.lexsort((idx, -sc))[:top]                          # score descending, ties by library index
            pm.append(np.full(len(order), m, dtype=np.int64)), pt.append(idx[order])
        pairs = torch.from_numpy(np.stack([np.concatenate(pm), np.concatenate(pt)], axis=1)).to(dev)
    dist.broadcast(pairs, src=0)
    pairs = pairs.cpu().numpy()
    pair_mask, pair_target = pairs[:, 0].astype(np.int32), pairs[:, 1].astype(np.int64)
    sel_s = time.perf_counter() - t_all
    t0 = time.perf_counter()
    shape = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, MASK_THR, MIRROR, None, rects, context=ctx)
    prep_s = time.perf_counter() - t0
    stats = {"kernel_ms": 0.0, "launches": 0, "pairs": 0, "load_s": 0.0, "score_s": 0.0}

    def local_score(mask_idx, slots):
        """this rank's pairs (slots = indices inside its shard): its DISTINCT selected targets are made resident `stage_n` at a time
        (regenerated with their variants: the stand-in for loading those targets' files), one scoring launch per batch"""
        out_l = np.zeros((len(mask_idx), 3), dtype=np.int64)
        distinct, inv = np.unique(slots, return_inverse=True)
        order = np.argsort(inv, kind="stable")
        isorted = inv[order]
        for b0 in range(0, len(distinct), stage_n):
            idx = distinct[b0:b0 + stage_n]
            lo, hi = np.searchsorted(isorted, b0), np.searchsorted(isorted, b0 + len(idx))
            rows = order[lo:hi]
            t1 = time.perf_counter()
            dense.generate_synthetic_list(LIB_SEED, rank * n + idx.astype(np.int64), 0, kind=0, with_variants=True)
            ctx.synchronize()
            t2 = time.perf_counter()
            gap, high, mir = shape.scoreLibraryPairs(dense, mask_idx[rows], (inv[rows] - b0).astype(np.int64))
            stats["load_s"] += t2 - t1
            stats["score_s"] += time.perf_counter() - t2
            ms, launches = shape.lastKernelMs()
            stats["kernel_ms"] += ms * launches
            stats["launches"] += launches
            stats["pairs"] += len(rows)
            out_l[rows, 0], out_l[rows, 1], out_l[rows, 2] = gap, high, mir
        return torch.from_numpy(out_l)

    t0 = time.perf_counter()
    res = cdist.sharded_pair_scores(local_score, pair_mask, pair_target, world * n, dev)
    torch.cuda.synchronize()
    score_wall = time.perf_counter() - t0
    st = torch.tensor([stats["kernel_ms"], stats["pairs"], stats["load_s"], stats["score_s"], score_wall], dtype=torch.float64, device=dev)
    allst = [torch.zeros_like(st) for _ in range(world)]
    dist.all_gather(allst, st)
    if rank != 0:
        return None
    allst = [x.cpu().numpy() for x in allst]
    res = res.cpu().numpy()
    ok, checked = True, 0
    if not args.no_cpu_baseline:
        from oracle import cds_oracle as O
        rng = np.random.default_rng(9)
        for m in sorted(set(rng.choice(g_m, size=min(4, g_m), replace=False).tolist())):
            sq = O.ShapeQuery(masks[m], MASK_THR, MIRROR, rects)
            for i in rng.choice(top, size=4, replace=False):
                j = m * top + int(i)
                t = int(pair_target[j])
                zg, gr = O.synth_variants(LIB_SEED, t, W, H)
                og, oh, _, om = sq.score(O.synth_image(LIB_SEED, t, 0, W, H), gr, zg)
                ok &= (int(res[j, 0]), int(res[j, 1]), bool(res[j, 2])) == (og, oh, bool(om))
                checked += 1
    if not ok:
        failed.append("sharded gradient scores differ from the CPU oracle's on the sampled pairs")
    n_pairs = len(pair_mask)
    kms = [float(a[0]) for a in allst]
    return {"workload": f"GradientAreaGap of the global top-{top} hits of {g_m} masks over the {world} x {n}-target sharded library = {n_pairs} pairs "
                        "(BASELINE configs[4] on N GPUs), every pair scored on the rank owning its target, results gathered in pair order",
            "entry": "dist.sharded_pair_scores (routing by owner + one gather) driving cds_gradscore: a rank's distinct selected targets resident a staging-slot batch at a time",
            "world": world, "pairs": n_pairs, "pairs_per_rank": [int(a[1]) for a in allst], "kernel_ms_per_rank": kms,
            "pairs_per_s_per_rank (kernel time)": [(a[1] / (a[0] * 1e-3)) if a[0] > 0 else None for a in allst],
            "pairs_per_s (kernel time, slowest rank)": n_pairs / (max(kms) * 1e-3) if max(kms) > 0 else None,
            "scoring_wall_s (regeneration of the selected synthetic targets included)": max(float(a[4]) for a in allst),
            "selected_target_regeneration_s_per_rank (benchmark scaffolding)": [float(a[2]) for a in allst],
            "scoring_calls_s_per_rank": [float(a[3]) for a in allst],
            "top_hit_selection_s (scan of the masks, device top-k, gather, merge, broadcast)": sel_s, "shape_mask_preparation_s": prep_s,
            "pairs_checked_against_oracle": checked, "sampled_pairs_equal_oracle": bool(ok)}


def overlapping_leg(args, cds, ctx, dev, stream, mask, rects, failed, n_o=262144, n_w=2048):
    """A synthetic library that OVERLAPS the benchmark mask the way real libraries overlap real masks (the four LM test
    images x 8,192 copies give every wave one of four workloads).  Every target is a sparse-lut target of the headline stream with a
    thinned, slightly displaced, dimmed copy of the mask's own arbor laid over it: half of the mask's pixels (one of 4 fixed random
    patterns), shifted by (dx, dy) in {-1, 0, 1}^2, brightness x {1, 0.85, 0.7, 0.55} -- so the target's depth colour FOLLOWS the mask's
    wherever they overlap (what makes real images expensive: stage 1 passes a third to a half of the visited records) while no two
    targets are alike.  Built on the device (torch ops over the dense staging slots), 262,144 targets; scores checked against the
    dense-layout kernel on 8 windows and against the CPU oracle on 16 targets."""
    n_o = min(n_o, max(n_w, args.targets // 4 // n_w * n_w))
    pitch_px = ((3 * W + 47) // 48 * 48) // 3
    tbuf = torch.zeros(cds.Library.bytes_required(W, H, n_w), dtype=torch.uint8, device=dev)
    lib_w = cds.Library(W, H, n_w, context=ctx, device_buffer=tbuf)
    view = tbuf[: n_w * H * pitch_px * 3].view(n_w, H, pitch_px, 3)
    mdev = torch.zeros((H, pitch_px, 3), dtype=torch.uint8, device=dev)
    mdev[:, :W] = torch.from_numpy(mask).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(20240)
    patterns = torch.rand((4, H, pitch_px), generator=g, device=dev) < 0.5
    scales = torch.tensor([1.0, 0.85, 0.7, 0.55], device=dev)
    combos = [(dx, dy) for dx in (-1, 0, 1) for dy in (-1, 0, 1)]

    def make_window(w0):
        """targets w0 .. w0 + n_w - 1 of the overlapping library into the staging slots (a pure function of w0)"""
        lib_w.generate_synthetic(LIB_SEED, 5_000_000 + w0, 0, n_w, kind=0)
        torch.cuda.synchronize()
        t = torch.arange(w0, w0 + n_w, device=dev)
        for c, (dx, dy) in enumerate(combos):
            idx = torch.nonzero((t * 7 + 3) % 9 == c).flatten()
            if idx.numel() == 0:
                continue
            ms = torch.roll(mdev, shifts=(dy, dx), dims=(0, 1))
            has = ms.max(dim=2).values > 0
            tt = t[idx]
            sel = patterns[(tt // 9) % 4] & has                                           # [k, H, pitch]
            col = (ms.to(torch.float32)[None] * scales[(tt // 36) % 4][:, None, None, None]).to(torch.uint8)
            sub = view[idx]
            view[idx] = torch.where(sel[..., None], col, sub)
        torch.cuda.synchronize()

    so = cds.SparseLibrary(W, H, n_o, 40000 * n_o + 4_000_000, context=ctx)
    t0 = time.perf_counter()
    for w0 in range(0, n_o, n_w):
        make_window(w0)
        so.append_from(lib_w, 0, n_w)
    build_s = time.perf_counter() - t0
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, MASK_THR, MIRROR, None, 0, False, DATA_THR, ZTOL, XY_SHIFT, rects, context=ctx)
    sc = torch.empty(n_o, dtype=torch.int32, device=dev)
    alg.searchSparseLibraryDevice(so, 0, n_o, sc.data_ptr())                              # (the pilot runs here)
    kernel, ratio = alg.lastScanKernel()
    ms_auto = timed(stream, lambda: alg.searchSparseLibraryDevice(so, 0, n_o, sc.data_ptr()), 3)
    rates = {}
    for name, env in (("stream16", "CDS_SPARSE_SCAN16"), ("records32", "CDS_SPARSE_SCAN32")):
        os.environ[env] = "1"
        alg.searchSparseLibraryDevice(so, 0, n_o, sc.data_ptr())
        rates[name] = n_o / (timed(stream, lambda: alg.searchSparseLibraryDevice(so, 0, n_o, sc.data_ptr()), 3) * 1e-3)
        del os.environ[env]
    alg.searchSparseLibraryDevice(so, 0, n_o, sc.data_ptr())
    # verification: 8 windows against the dense-layout kernel, 16 targets against the oracle
    dsc = torch.empty(n_w, dtype=torch.int32, device=dev)
    wins = sorted({(i * (n_o // n_w - 1) // 7) * n_w for i in range(8)})
    same, ok_or, checked = True, True, 0
    for w0 in wins:
        make_window(w0)
        alg.searchLibraryDevice(lib_w, 0, n_w, dsc.data_ptr())
        same &= bool(torch.equal(dsc, sc[w0:w0 + n_w]))
        if not args.no_cpu_baseline and w0 in (wins[0], wins[-1]):
            from oracle import cds_oracle as O
            imgs = lib_w.download(0, 8)
            ws, wm = O.PixelMatchQuery(mask, MASK_THR, DATA_THR, ZTOL, XY_SHIFT, MIRROR, rects).scan(imgs, n_threads=usable_cores(), partition=2)
            hs = sc[w0:w0 + 8].cpu().numpy()
            ok_or &= bool(np.array_equal(ws, hs & 0x7FFFFFFF) and np.array_equal(wm.astype(np.int64), ((hs >> 31) & 1).astype(np.int64)))
            checked += 8
    if not same:
        failed.append("overlapping-synthetic leg: sparse and dense scans disagree")
    if not ok_or:
        failed.append("overlapping-synthetic leg: scores differ from the CPU oracle's")
    hs = (sc & 0x7FFFFFFF).cpu().numpy()
    return {"workload": f"1 mask x {n_o} synthetic targets that overlap it (sparse-lut targets + a thinned, displaced, dimmed copy of the mask's arbor: depth colour follows "
                        "the mask's where they overlap), xyShift 2, mirror on", "records_per_target": so.pixels / n_o,
            "comparisons_per_s": n_o / (ms_auto * 1e-3), "ms_per_pass": ms_auto,
            "kernel_chosen_by_the_pilot": "stream16" if kernel == 1 else "records32", "share_of_visited_records_reaching_a_round": ratio,
            "comparisons_per_s_by_kernel": rates, "library_build_s": build_s,
            "score_quartiles": [int(v) for v in np.percentile(hs, [0, 25, 50, 75, 100])], "mask_pixels": int(alg.getQuerySize()),
            "windows_equal_dense_scan": same, "windows_checked": len(wins), "targets_checked_against_oracle": checked,
            "targets_equal_oracle": bool(ok_or) if checked else None}


def variant_shard_leg(cds, ctx, fx, rects, failed, n_t=2048, pairs_per_target=2.2):
    """Kernel 2 over a library whose variants sit in a variant shard (round 5): the reference's three LM test targets with
    their GradientImage and a device-made zgap, packed under n_t names, the selected targets' blobs read back and DECODED ON THE GPU batch
    by batch while the previous batch is scored (processor.calculateGradientScoresFromShardTable) -- every score compared with the same
    pairs scored from memory."""
    import tempfile
    from colormipsearch_amd import processor as P
    keys = [("lm_BJD", "grad_BJD"), ("lm_VT016795", "grad_VT016795"), ("lm_VT033614", "grad_VT033614")]
    H, W = fx[keys[0][0]].shape[:2]
    lib = cds.Library(W, H, len(keys), context=ctx)
    lib.upload(0, np.stack([fx[t] for t, _ in keys]))
    lib.upload_variants(0, gradients=np.stack([fx[g] for _, g in keys]))
    lib.make_zgap_variants(0, len(keys), 20, 10, rects)
    imgs = lib.download(0, len(keys))
    grads, zgaps = lib.download_variants(0, len(keys))
    masks = np.stack([fx[k] for k in sorted(k for k in fx if k.startswith("em_"))])
    rng = np.random.default_rng(7)
    n_pairs = int(n_t * pairs_per_target)
    t_idx = np.concatenate([np.arange(n_t), rng.integers(0, n_t, n_pairs - n_t)]).astype(np.int64)
    m_idx = rng.integers(0, len(masks), n_pairs).astype(np.int32)
    order = np.lexsort((t_idx, m_idx))
    table = lambda ti: P.MatchTable(m_idx[order], ti, np.full(n_pairs, 100, np.int32), np.full(n_pairs, 0.01, np.float32), np.zeros(n_pairs, bool))
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 1, 2, rects)
    proc = P.GpuColorMIPSearchProcessor(cds.ColorMIPSearch(0.01, 20, provider), lib, None)
    ref = proc.calculateGradientScoresTable(masks, table(t_idx[order] % len(keys)), 20, excludedRegions=rects)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "variants.cdsv")
        rep = 16
        t0 = time.perf_counter()
        try:
            with cds.VariantShardWriter(path, W, H) as w:   # target i of the shard = fixture i % 3
                block = (np.concatenate([imgs] * rep), np.concatenate([grads] * rep), np.concatenate([zgaps] * rep))
                for first in range(0, n_t, len(keys) * rep):
                    k = min(len(keys) * rep, n_t - first)
                    w.append(block[0][:k], block[1][:k], block[2][:k], threads=usable_cores())
        except (OSError, cds.CdsError) as e:   # (~0.6 GB in the temporary directory: no room there is the box's business, not a result)
            proc.close()
            return {"skipped": f"could not write the shard: {type(e).__name__}: {e}"}
        t_write = time.perf_counter() - t0
        shard = cds.VariantShard(path)
        proc.calculateGradientScoresFromShardTable(masks, table(t_idx[order]).take(np.arange(64)), shard, 20, excludedRegions=rects, batch=512, threads=usable_cores())
        t0 = time.perf_counter()
        out = proc.calculateGradientScoresFromShardTable(masks, table(t_idx[order]), shard, 20, excludedRegions=rects, batch=512, threads=usable_cores())
        t = time.perf_counter() - t0
        same = bool(np.array_equal(out.gradientAreaGap, ref.gradientAreaGap) and np.array_equal(out.highExpressionArea, ref.highExpressionArea))
        res = {"workload": f"{n_t} distinct targets (the reference's 3 LM test targets + GradientImage + device-made zgap under {n_t} names) x {n_pairs} pairs of "
                           f"{len(masks)} EM masks, variants read from a variant shard and decoded on the GPU (page cache warm)",
               "targets_per_s": n_t / t, "pairs_per_s": n_pairs / t, "seconds": t, "shard_bytes_per_target": shard.file_bytes / n_t,
               "blob_bytes_read_per_s": shard.file_bytes / t, "gradients_stored_as_second_differences": int((shard.flags & shard.GRADIENT_DIFF2 != 0).sum()),
               "write_targets_per_s_from_memory": n_t / t_write, "host_threads": usable_cores(), "scores_equal_to_in_memory_path": same,
               "path": proc.last_from_files}
        shard.close()
    proc.close()
    if not same:
        failed.append("gradient scores through the variant shard differ from the in-memory path's")
    return res


def config0_leg(args, cds, ctx, dev, stream, dense, mask, rects, failed, n0=100):
    """BASELINE configs[0]: 1 mask x 100 synthetic targets, xyShift 0, no mirror -- the reference's own CPU-runnable case (plumbing).  The C
    restatement of the reference's per-pair loop on ONE host thread (the per-pair call the reference's worker threads make), the same
    pairs through the dense and the sparse resident layout on the GPU, scores and mirrored flags equal."""
    dense.generate_synthetic(LIB_SEED, 0, 0, n0, kind=0)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, MASK_THR, False, None, 0, False, DATA_THR, ZTOL, 0, rects, context=ctx)
    sc_d = torch.empty(n0, dtype=torch.int32, device=dev)
    alg.searchLibraryDevice(dense, 0, n0, sc_d.data_ptr())
    t_dense = timed(stream, lambda: alg.searchLibraryDevice(dense, 0, n0, sc_d.data_ptr()), 5)
    sl = cds.SparseLibrary(W, H, n0, 40000 * n0, context=ctx)
    sl.append_from(dense, 0, n0)
    sc_s = torch.empty(n0, dtype=torch.int32, device=dev)
    alg.searchSparseLibraryDevice(sl, 0, n0, sc_s.data_ptr())
    t_sparse = timed(stream, lambda: alg.searchSparseLibraryDevice(sl, 0, n0, sc_s.data_ptr()), 5)
    same = bool(torch.equal(sc_d, sc_s))
    out = {"workload": f"1 mask x {n0} synthetic 1210x566 RGB targets, xyShift 0, no mirror (BASELINE configs[0])",
           "gpu_ms_per_pass": {"dense_layout": t_dense, "sparse_layout": t_sparse}, "dense_and_sparse_scores_equal": same}
    if not same:
        failed.append("configs[0]: dense and sparse scans disagree")
    if not args.no_cpu_baseline:
        from oracle import cds_oracle as O
        imgs = dense.download(0, n0)
        q = O.PixelMatchQuery(mask, MASK_THR, DATA_THR, ZTOL, 0, False, rects)
        q.scan(imgs[:4], n_threads=1)
        t0 = time.perf_counter()
        ws, wm = q.scan(imgs, n_threads=1)
        dt = time.perf_counter() - t0
        hs = sc_d.cpu().numpy()
        ok = bool(np.array_equal(ws, hs & 0x7FFFFFFF) and not wm.any() and not (hs < 0).any())
        out["cpu"] = {"comparisons_per_s": n0 / dt, "threads": 1, "kind": "port", "what": "the C restatement of the reference's per-pair loop (one transform), one host thread",
                      "scores_equal_gpu": ok}
        if not ok:
            failed.append("configs[0]: GPU scores differ from the CPU oracle's")
    return out


def timed(stream, fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def aux_legs(args, cds, ctx, dev, stream, dense, slib, alg, mask, rects, scores, n, stage_n, dense_bytes_per_target, failed):
    aux = {}
    pmc = load_pmc() or {}
    # (a) the same scan over the DENSE layout (packed RGB rows streamed from HBM, cds::scan_kernel): the first stage_n targets
    n_d = stage_n
    dense.generate_synthetic(LIB_SEED, 0, 0, n_d, kind=args.kind)
    dsc = torch.empty(n_d, dtype=torch.int32, device=dev)
    for _ in range(2):
        alg.searchLibraryDevice(dense, 0, n_d, dsc.data_ptr())
    dms = timed(stream, lambda: alg.searchLibraryDevice(dense, 0, n_d, dsc.data_ptr()), 5)
    same_d = bool(torch.equal(dsc, scores[:n_d]))
    d_ach = dense_bytes_per_target * n_d / (dms * 1e-3) / 1e9
    dcls = pmc.get("classes", {}).get("dense_scan")
    if dcls and "hbm_bytes" not in dcls.get("per_target", {}):
        dcls = None
    aux["dense_layout"] = {"workload": f"1 mask x {n_d} targets resident as packed RGB rows (2,064,768 B each)", "kernel": "cds::scan_kernel<xy2,mirror,single-mask>",
                           "comparisons_per_s": n_d / (dms * 1e-3), "ms_per_pass": dms,
                           "roofline": {"bound": "hbm", "achieved": d_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d_ach / HBM_PEAK_GBS,
                                        "algorithmic_bytes_per_target": dense_bytes_per_target,
                                        "traffic": (dcls["per_target"]["hbm_bytes"] * n_d) if dcls else None,
                                        "frac_by_traffic": (dcls["per_target"]["hbm_bytes"] * n_d / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS) if dcls else None},
                           "scores_equal_sparse_scan": same_d}
    if not same_d:
        failed.append("dense-layout scan disagrees with the sparse scan")

    # (b) BASELINE configs[2] at FULL size: `--batch-masks` (1,024) masks x ALL resident targets over the sparse library;
    # sampled masks are checked against the dense batch kernel (128 masks share one pass over dense rows) and the oracle
    n_m, n_t = args.batch_masks, n
    mlib2 = cds.Library(W, H, max(128, n_m), context=ctx)
    mlib2.generate_synthetic(MASK_SEED, 0, 0, max(128, n_m), kind=2)
    masks = mlib2.download(0, max(128, n_m))
    del mlib2
    t0 = time.perf_counter()
    batch = cds.PixelMatchColorDepthSearchAlgorithm(masks[:n_m], MASK_THR, MIRROR, None, 0, False, DATA_THR, ZTOL, XY_SHIFT, rects, context=ctx)
    prep_s = time.perf_counter() - t0
    bscores = torch.empty(n_m * n_t, dtype=torch.int32, device=dev)
    bms = timed(stream, lambda: batch.searchSparseLibraryDevice(slib, 0, n_t, bscores.data_ptr()), 1)
    bview = bscores.view(n_m, n_t)
    same_b = bool(torch.equal(bview[0], scores[:n_t]))   # mask 0 of the batch is the benchmark mask
    n_t2 = min(16384, n_d)
    batch128 = cds.PixelMatchColorDepthSearchAlgorithm(masks[:128], MASK_THR, MIRROR, None, 0, False, DATA_THR, ZTOL, XY_SHIFT, rects, context=ctx)
    b2 = torch.empty(128 * n_t2, dtype=torch.int32, device=dev)
    batch128.searchLibraryDevice(dense, 0, n_t2, b2.data_ptr())
    b2ms = timed(stream, lambda: batch128.searchLibraryDevice(dense, 0, n_t2, b2.data_ptr()), 2)
    k128 = min(128, n_m)
    same_b2 = bool(torch.equal(b2.view(128, n_t2)[:k128], bview[:k128, :n_t2]))
    # masks beyond the first 128, spread over the set: the single-mask dense kernel on the staging targets
    sample_masks = sorted({min(n_m - 1, i) for i in (0, n_m // 3, (2 * n_m) // 3, n_m - 1)})
    same_b3, ok_or = True, True
    for mi in sample_masks:
        a1 = cds.PixelMatchColorDepthSearchAlgorithm(masks[mi], MASK_THR, MIRROR, None, 0, False, DATA_THR, ZTOL, XY_SHIFT, rects, context=ctx)
        a1.searchLibraryDevice(dense, 0, n_d, dsc.data_ptr())
        same_b3 &= bool(torch.equal(dsc, bview[mi, :n_d]))
        if not args.no_cpu_baseline:
            from oracle import cds_oracle as O
            qm = O.PixelMatchQuery(masks[mi], MASK_THR, DATA_THR, ZTOL, XY_SHIFT, MIRROR, rects)
            ws, wm = qm.scan(dense.download(0, 128), n_threads=usable_cores(), partition=8)
            hs = bview[mi, :128].cpu().numpy()
            ok_or &= bool(np.array_equal(ws, hs & 0x7FFFFFFF) and np.array_equal(wm.astype(np.int64), ((hs >> 31) & 1).astype(np.int64)))
    # the same workload END TO END through the caller the reference would use (processor.findAllColorDepthMatchesTable): mask sets of
    # 128 prepared on a second host thread while earlier sets are scanned (two search workers), ColorMIPSearch.isMatch (pctPositivePixels 1 %) and the
    # compaction on the device, only the match list downloaded
    from colormipsearch_amd import processor as PR0
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(MIRROR, DATA_THR, ZTOL * 100, XY_SHIFT, cds.label_regions, context=ctx)
    proc0 = PR0.GpuColorMIPSearchProcessor(cds.ColorMIPSearch(1.0, MASK_THR, provider), None, None, first=0, count=n_t, sparse_library=slib)
    torch.cuda.synchronize()
    ctx.reset_stream()        # (the context's own non-blocking stream: the synchronous copies of the preparing thread must not wait for the scans)
    # (twice: the first call of a process also allocates what later calls recycle -- two 512 MB score matrices, the lists, the sets' plans,
    # cds::cache_alloc --; a service runs many searches per process, so `seconds` is the second call and the first is reported beside it)
    e2e_runs = []
    for _ in range(2):
        t0 = time.perf_counter()
        mtable = proc0.findAllColorDepthMatchesTable(masks[:n_m], mask_set_size=128)
        e2e_runs.append(time.perf_counter() - t0)
    e2e_s = e2e_runs[-1]
    ctx.set_stream(stream.cuda_stream)
    hs = bview.cpu().numpy() if n_m * n_t <= (1 << 26) else None     # (small runs: the whole list against a host-side filter)
    want_matches = None
    if hs is not None:
        sz = np.asarray(batch.getQuerySizes())
        want_matches = int(sum(int(cds.ColorMIPSearch(1.0, MASK_THR, provider).isMatchArray(hs[m] & 0x7FFFFFFF, int(sz[m])).sum()) for m in range(n_m)))
    sample_ok = True
    for mi in sample_masks[:2]:                                        # sampled masks: the list's rows == isMatch over the mask's score row
        row = (bview[mi] & 0x7FFFFFFF).cpu().numpy()
        keep = cds.ColorMIPSearch(1.0, MASK_THR, provider).isMatchArray(row, int(batch.getQuerySize(mi)))
        sel = mtable.maskIndex == mi
        sample_ok &= bool(np.array_equal(mtable.targetIndex[sel], np.flatnonzero(keep)) and np.array_equal(mtable.matchingPixels[sel], row[keep]))
    if not sample_ok or (want_matches is not None and want_matches != len(mtable)):
        failed.append("device match list disagrees with the host-side isMatch filter")
    mcls = pmc.get("classes", {}).get("sparse_multi")
    mroof = None
    if mcls and {"SQ_INSTS_VALU", "hbm_bytes"} <= set(mcls.get("per_comparison", {})):
        pc = mcls["per_comparison"]
        issue_ns = ctx.valu_issue_ns()
        n_cmp = n_m * n_t
        simds = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
        ach = pc["SQ_INSTS_VALU"] * n_cmp / (bms * 1e6)
        hb = pc["hbm_bytes"] * n_cmp / (bms * 1e-3) / 1e9
        mroof = {"bound": "hbm", "kernel": "cds::sparse_multi_kernel", "achieved": hb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hb / HBM_PEAK_GBS,
                 "traffic": pc["hbm_bytes"] * n_cmp,
                 "issue": {"achieved": ach, "peak": simds / issue_ns, "unit": "G wave64-VALU-instructions/s", "frac": ach / (simds / issue_ns)},
                 "note": "the records are streamed once per group of masks, so HBM is far from saturated; the kernel waits for L2 gathers of the group's maps and planes (DESIGN.md 4c)",
                 "counters": {"source": f"profiles/scan_pmc.json (tag {pmc.get('tag')}), class sparse_multi", "measured_in_this_run": False,
                              "stale_sources": pmc.get("stale_sources"), "per_comparison": pc}}
    aux["mask_batch"] = {"workload": f"{n_m} masks x {n_t} resident targets (BASELINE configs[2]" + (" at full size" if n_m >= 1024 and n_t >= 1 << 20 else "") + "), sparse layout",
                         "comparisons_per_s": n_m * n_t / (bms * 1e-3), "seconds_per_pass": bms * 1e-3, "mask_preparation_s": prep_s,
                         "kernel": "cds::sparse_multi_kernel (groups of up to 8 masks per launch)", "multi_mask_groups": batch.sparseGroups(),
                         "roofline": mroof,
                         "end_to_end": {"what": "processor.findAllColorDepthMatchesTable: mask sets of 128 through three overlapped stages (preparation on one host "
                                                "thread; two search workers: scan + isMatch (pctPositivePixels 1) + compaction on the device, then the list "
                                                "download and the host-side columns of a set while the next set is scanned)",
                                        "pipeline_stage_seconds": getattr(proc0, "last_pipeline", None),
                                        "seconds": e2e_s, "first_call_seconds": e2e_runs[0], "seconds_over_seconds_per_pass": e2e_s / (bms * 1e-3),
                                        "comparisons_per_s": n_m * n_t / e2e_s, "matches": int(len(mtable)),
                                        "match_list_bytes": int(len(mtable)) * 12, "score_matrix_bytes": int(n_m) * int(n_t) * 4,
                                        "sampled_masks_equal_host_filter": bool(sample_ok)},
                         "mask0_scores_equal_single_mask_scan": same_b,
                         "sampled_masks": sample_masks, "sampled_masks_equal_dense_scan": same_b3,
                         "sampled_masks_equal_oracle_on_128_targets": bool(ok_or) if not args.no_cpu_baseline else None,
                         "dense_batch_kernel": {"workload": f"128 masks x {n_t2} dense targets in one batched pass",
                                                "comparisons_per_s": 128 * n_t2 / (b2ms * 1e-3), "ms_per_pass": b2ms,
                                                "scores_equal_sparse_scan": same_b2}}
    if not same_b:
        failed.append("mask-batch scan disagrees with the single-mask scan")
    if not same_b2:
        failed.append("dense batch kernel disagrees with the sparse scan")
    if not same_b3:
        failed.append("mask-batch scan disagrees with the dense scan on sampled masks")
    if not ok_or:
        failed.append("mask-batch scan disagrees with the oracle on sampled masks")
    del batch128, b2

    # (b2) BASELINE configs[4]: GradientAreaGap of the top-500 hits of EVERY mask of the batch (1,024 x 500 = 512,000 pairs over
    # the 1 M-target library).  The library is resident as signal-pixel records only, so the selected targets are visited in
    # windows of `stage_n` consecutive indices: each window is regenerated on the device WITH its gradient / zgap variants
    # into the dense staging slots and its pairs are scored with one cds::gradscore_list_kernel launch
    # (processor.calculateGradientScoresBatchedTable); per-mask normalisation on the host.  64 random pairs go to the oracle.
    from colormipsearch_amd import processor as PR
    n_gm, top = n_m, min(500, n_t)
    t0 = time.perf_counter()
    shp_masks = masks[:n_gm]
    tk_t, tk_s, tk_m = batch.topKOfDeviceScores(bscores.data_ptr(), n_t, top)   # (per-mask top-k on the device: cds_scores_topk)
    pm = np.repeat(np.arange(n_gm, dtype=np.int32), top)
    ps = tk_t.reshape(-1).astype(np.int64)
    pv = tk_s.reshape(-1).astype(np.int32)
    pmir = tk_m.reshape(-1)
    sel_s = time.perf_counter() - t0
    chk = (bview[n_gm // 2] & 0x7FFFFFFF).cpu().numpy().astype(np.int64)            # one mask's row against a stable host sort
    if not np.array_equal(tk_t[n_gm // 2], np.lexsort((np.arange(n_t), -chk))[:top]):
        failed.append("device top-k disagrees with the host-side selection")
    sizes = np.asarray(batch.getQuerySizes(), dtype=np.float64)
    table = PR.MatchTable(pm, ps, pv, (pv / np.maximum(sizes[pm], 1.0)).astype(np.float32), pmir)
    proc = PR.GpuColorMIPSearchProcessor(None, None, None, first=0, count=n_t, sparse_library=slib)
    load_sel = lambda lib, idx: lib.generate_synthetic_list(LIB_SEED, idx, 0, kind=args.kind, with_variants=True)
    # Only the SELECTED targets are made resident (as the reference opens only the files of the matches it scores): their distinct
    # indices, `stage_n` at a time, regenerated with their variants into the dense staging slots -- the synthetic stand-in for loading
    # those targets' files.  Run twice: (1) the default policy -- a target serves ~2 pairs here, so the scorer reads the zgap bytes
    # itself and NO conversion pass runs; (2) CDS_SHAPE_ZSLICE=1: every loaded zgap is first converted into its zslice plane (round 3's
    # only path).  Same scores; the second run's table is the one checked against the oracle below as well.
    runs = {}
    for label, env in (("zgap bytes read by the scorer (default policy here)", None), ("zslice plane (conversion pass per loaded target)", "1")):
        if env is None:
            os.environ.pop("CDS_SHAPE_ZSLICE", None)
        else:
            os.environ["CDS_SHAPE_ZSLICE"] = env
        tcopy = PR.MatchTable(pm.copy(), ps.copy(), pv.copy(), table.normalizedScore.copy() if False else (pv / np.maximum(sizes[pm], 1.0)).astype(np.float32), pmir.copy())
        t0 = time.perf_counter()
        tcopy = proc.calculateGradientScoresSelectedTable(shp_masks, tcopy, load_sel, MASK_THR, MIRROR, None, rects, batch=stage_n, library=dense)
        torch.cuda.synchronize()
        runs[label] = (tcopy, dict(proc.last_gradscore), time.perf_counter() - t0)
    os.environ.pop("CDS_SHAPE_ZSLICE", None)
    (table, gs, wall_s), (table2, gs2, wall2_s) = runs.values()
    if not (np.array_equal(table.gradientAreaGap, table2.gradientAreaGap) and np.array_equal(table.highExpressionArea, table2.highExpressionArea)):
        failed.append("gradient scores differ between the zgap-bytes path and the zslice-plane path")
    gms = gs["kernel_ms"]
    ok_g, n_checked = True, 0
    if not args.no_cpu_baseline:
        from oracle import cds_oracle as O
        # >= 1,000 pairs against the CPU oracle: 8 masks (one CPU mask preparation each: two r = 60 / r = 20 max filters, seconds) x 128
        # of their top hits, the pairs of a mask on a thread pool (the C oracle releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        rng = np.random.default_rng(7)
        with ThreadPoolExecutor(max_workers=usable_cores()) as pool:
            for m in sorted(set(rng.choice(n_gm, size=min(8, n_gm), replace=False).tolist())):
                sq = O.ShapeQuery(masks[m], MASK_THR, MIRROR, rects)

                def one(i, m=m, sq=sq):
                    j = m * top + int(i)
                    tgt = O.synth_image(LIB_SEED, int(ps[j]), args.kind, W, H)
                    zg, gr = O.synth_variants(LIB_SEED, int(ps[j]), W, H)
                    og, oh, _, _ = sq.score(tgt, gr, zg)
                    return (int(table.gradientAreaGap[j]), int(table.highExpressionArea[j])) == (og, oh)

                res = list(pool.map(one, rng.choice(top, size=min(128, top), replace=False)))
                ok_g &= all(res)
                n_checked += len(res)
    gcls = pmc.get("classes", {}).get("gradscore_list")
    if gcls and "hbm_bytes" not in gcls.get("per_pair", {}):
        gcls = None
    n_pairs = len(pm)
    g_alg = n_pairs * 2081200 / (gms * 1e-3) / 1e9            # 2,081,200 B per pair at R = 170
    g_traffic = gcls["per_pair"]["hbm_bytes"] * n_pairs if gcls else None
    aux["gradient_scores"] = {"workload": f"GradientAreaGap of the top-{top} hits of {n_gm} masks over the {n_t}-target library = {n_pairs} pairs "
                                          f"(BASELINE configs[4]" + (" at full size" if n_gm >= 1024 and n_t >= 1 << 20 else "") + f"), the {gs['distinct_targets']} distinct targets made resident "
                                          f"{stage_n} at a time with their variants ({gs['launches']} scoring launches)", "kernel": "cds::gradscore_list_kernel",
                              "pairs_per_s": n_pairs / (gms * 1e-3), "kernel_ms_total": gms, "launches": gs["launches"],
                              "s_of_z_source": "zgap bytes read by the scorer: no conversion pass (the default policy at %.1f pairs per selected target)" % (n_pairs / max(1, gs["distinct_targets"]))
                                               if gs["launches_on_zslice_plane"] == 0 else "zslice plane",
                              "pairs_per_s_including_variant_conversion": n_pairs / ((gms + gs["conversion_ms"]) * 1e-3),
                              "product_pairs_per_s (shape-mask preparation + scoring calls incl. any conversion + top-hit selection)":
                                  n_pairs / ((gs.get("mask_preparation_s") or 0) + (gs.get("score_calls_s") or 0) + sel_s),
                              "with_zslice_conversion (CDS_SHAPE_ZSLICE=1: round 3's path, every loaded zgap converted first)": {
                                  "kernel_ms_total": gs2["kernel_ms"], "conversion_ms_total": gs2["conversion_ms"], "converted_targets": gs2["converted_slots"],
                                  "pairs_per_s": n_pairs / (gs2["kernel_ms"] * 1e-3),
                                  "pairs_per_s_including_variant_conversion": n_pairs / ((gs2["kernel_ms"] + gs2["conversion_ms"]) * 1e-3),
                                  "scoring_calls_s": gs2.get("score_calls_s"), "scores_equal": True},
                              "end_to_end_s": wall_s, "end_to_end_pairs_per_s": n_pairs / wall_s, "top_hit_selection_s": sel_s,
                              "product_seconds (mask preparation + scoring calls + selection)": (gs.get("mask_preparation_s") or 0) + (gs.get("score_calls_s") or 0) + sel_s,
                              "end_to_end_breakdown_s": {"shape_mask_preparation": gs.get("mask_preparation_s"),
                                                         "regeneration_of_the_selected_synthetic_targets_with_variants (benchmark scaffolding: a real library loads those targets' files)": gs.get("load_s"),
                                                         "scoring_calls (pair upload + kernel + read-back)": gs.get("score_calls_s")},
                              "end_to_end_note": "includes the shape-mask preparation of all masks, the regeneration of the selected targets of the synthetic library "
                                                 "with gradient and zgap variants, the host-side pair bookkeeping and normalisation",
                              "distinct_targets": int(gs["distinct_targets"]), "normalized_scores_positive": int((table.normalizedScore > 0).sum()),
                              "roofline": {"bound": "hbm", "achieved": (g_traffic / (gms * 1e-3) / 1e9) if g_traffic else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": (g_traffic / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS) if g_traffic else None, "traffic": g_traffic,
                                           "traffic_source": (f"profiles/scan_pmc.json (tag {pmc.get('tag')}), correction {gcls['per_pair'].get('hbm_correction', 'x2 (profile older than round 5)')}: "
                                                              f"the scorer's 12-byte groups at scattered pixels are default-policy loads = 128-B line fetches (tools/fetchcal: only non-temporal "
                                                              f"scattered loads go out as 64-B requests), so 2 x FETCH_SIZE = {gcls['per_pair'].get('hbm_bytes_doubled')} B per pair is the figure unless the "
                                                              f"class carries the request-size counters; stale sources: {pmc.get('stale_sources')}") if gcls else None,
                                           "survey_8d_bytes_per_pair": 2081200, "survey_8d_GBps": g_alg,
                                           "note": "the list kernel reads only the groups the mask's lists select, fewer bytes than SURVEY 8(d) counts; frac is by measured traffic"},
                              "pairs_checked_against_oracle": n_checked, "sampled_pairs_equal_oracle": bool(ok_g)}
    if not ok_g:
        failed.append("gradient scores differ from the CPU oracle's on the sampled pairs")
    del bscores, bview, batch

    # (c) the reference's own fixture images and known answers ("in every benchmark binary"): the JUnit pairs
    # of PixelMatchColorDepthSearchAlgorithmTest through the sparse scan, the dense scan and the shape scorer
    fpath = os.path.join(ROOT, "tests", "golden", "cds_fixtures.npz")
    if os.path.exists(fpath):
        from tests.golden import expected as E
        with np.load(fpath) as z:
            fx = {k: z[k] for k in z.files}
        tkeys = sorted({t for _, t, _, _ in E.PIXEL_MATCH_PROVIDER})
        tg = np.stack([fx[k] for k in tkeys])
        fl = cds.Library(W, H, len(tkeys), context=ctx)
        fl.upload(0, tg)
        fs = cds.SparseLibrary(W, H, len(tkeys), int((tg.max(axis=3) > 0).sum()) + 4 * H * len(tkeys) + 64, context=ctx)
        fs.append_from(fl)
        got, ok_f = [], True
        for mk, tk, want_s, want_m in E.PIXEL_MATCH_PROVIDER:
            a = cds.PixelMatchColorDepthSearchAlgorithm(fx[mk], 20, True, None, 0, False, 20, 0.01, 2, E.STD_RECTS, context=ctx)
            t = tkeys.index(tk)
            s1, m1 = a.searchSparseLibrary(fs, t, 1)
            s2, m2 = a.searchLibrary(fl, t, 1)
            got.append([mk, tk, int(s1[0, 0]), bool(m1[0, 0])])
            ok_f &= (int(s1[0, 0]), bool(m1[0, 0])) == (want_s, want_m) == (int(s2[0, 0]), bool(m2[0, 0]))
        sk = E.SHAPE_SCORES[2]   # the pair with a ZGapImage file: (mask, target, grad, zgap, gaps, highExpression, score, mirrored)
        fl.upload_variants(tkeys.index(sk[1]), fx[sk[2]][None], fx[sk[3]][None])
        sh = cds.Shape2DMatchColorDepthSearchAlgorithm(fx[sk[0]], 20, True, None, E.STD_RECTS, context=ctx)
        gap, high, smir = sh.scoreLibraryPairs(fl, [0], [tkeys.index(sk[1])])
        ok_f &= (int(gap[0]), int(high[0]), bool(smir[0])) == (sk[4], sk[5], sk[7])
        aux["reference_fixtures"] = {"pixel_match": got, "shape": [int(gap[0]), int(high[0]), bool(smir[0])], "equal_to_reference_junit_values": bool(ok_f)}
        # (d) the same scan on REAL images: the reference's LM test MIPs replicated into a resident library (they carry 2x the
        # records of a synthetic target and all overlap the EM test masks, so far more of them reach the interval test)
        del fl, fs
        lkeys = sorted(k for k in fx if k.startswith("lm_"))
        copies = max(1, stage_n // len(lkeys))
        n_rd = copies * len(lkeys)                       # dense slots holding the images
        reps = max(1, 32768 // n_rd)
        n_r = n_rd * reps                                # sparse targets: the staging slots appended `reps` times
        limgs = np.stack([fx[k] for k in lkeys])
        for c in range(copies):
            dense.upload(c * len(lkeys), limgs)
        rs = cds.SparseLibrary(W, H, n_r, int((limgs.max(axis=3) > 0).sum()) * copies * reps + 8 * H * n_r + 64, context=ctx)
        for _ in range(reps):
            for first in range(0, n_rd, 4096):
                rs.append_from(dense, first, min(4096, n_rd - first))
        real = {"workload": f"{len(lkeys)} LM test MIPs of the reference x {copies * reps} copies = {n_r} resident sparse targets ({n_rd} of them "
                            f"also dense), xyShift=2, mirror on", "records_per_target": rs.pixels / n_r, "masks": {}}
        ok_r = True
        for mk in sorted(k for k in fx if k.startswith("em_")):
            a = cds.PixelMatchColorDepthSearchAlgorithm(fx[mk], 20, True, None, 0, False, 20, 0.01, 2, E.STD_RECTS, context=ctx)
            rsp = torch.empty(n_r, dtype=torch.int32, device=dev)
            rde = torch.empty(n_rd, dtype=torch.int32, device=dev)
            a.searchSparseLibraryDevice(rs, 0, n_r, rsp.data_ptr())
            rms = timed(stream, lambda: a.searchSparseLibraryDevice(rs, 0, n_r, rsp.data_ptr()), 3)
            a.searchLibraryDevice(dense, 0, n_rd, rde.data_ptr())
            dms = timed(stream, lambda: a.searchLibraryDevice(dense, 0, n_rd, rde.data_ptr()), 1)
            same_r = bool(torch.equal(rsp.view(reps, n_rd), rde.view(1, n_rd).expand(reps, n_rd)))
            ok_r &= same_r
            real["masks"][mk] = {"sparse_comparisons_per_s": n_r / (rms * 1e-3), "dense_comparisons_per_s": n_rd / (dms * 1e-3),
                                 "scores_equal": same_r}
        # (e) the same images at the reference CLI's DEFAULT parameters (AbstractColorDepthMatchArgs.java:18-31: thresholds 100 / 100,
        # pixColorFluctuation 2, xyShift 0, no mirror) over the library above and over one built with minimum data threshold 100
        # (pixels the scan never looks at are not stored): same scores, fewer records
        rs100 = cds.SparseLibrary(W, H, n_r, rs.pixels + 64, context=ctx, min_threshold=100)
        for _ in range(reps):
            for first in range(0, n_rd, 4096):
                rs100.append_from(dense, first, min(4096, n_rd - first))
        cli = {"parameters": "maskThreshold 100, dataThreshold 100, pixColorFluctuation 2, xyShift 0, mirror off (the reference CLI's defaults)",
               "records_per_target": rs.pixels / n_r, "records_per_target_with_minimum_data_threshold_100": rs100.pixels / n_r, "masks": {}}
        for mk in sorted(k for k in fx if k.startswith("em_")):
            a = cds.PixelMatchColorDepthSearchAlgorithm(fx[mk], 100, False, None, 0, False, 100, 0.02, 0, E.STD_RECTS, context=ctx)
            r0 = torch.empty(n_r, dtype=torch.int32, device=dev)
            r1 = torch.empty(n_r, dtype=torch.int32, device=dev)
            rde = torch.empty(n_rd, dtype=torch.int32, device=dev)
            a.searchSparseLibraryDevice(rs, 0, n_r, r0.data_ptr())
            ms0 = timed(stream, lambda: a.searchSparseLibraryDevice(rs, 0, n_r, r0.data_ptr()), 3)
            a.searchSparseLibraryDevice(rs100, 0, n_r, r1.data_ptr())
            ms1 = timed(stream, lambda: a.searchSparseLibraryDevice(rs100, 0, n_r, r1.data_ptr()), 3)
            a.searchLibraryDevice(dense, 0, n_rd, rde.data_ptr())
            torch.cuda.synchronize()
            same_c = bool(torch.equal(r0, r1) and torch.equal(r0.view(reps, n_rd), rde.view(1, n_rd).expand(reps, n_rd)))
            ok_r &= same_c
            cli["masks"][mk] = {"sparse_comparisons_per_s": n_r / (ms0 * 1e-3), "sparse_comparisons_per_s_with_minimum_data_threshold_100": n_r / (ms1 * 1e-3),
                                "scores_equal (both libraries, dense scan)": same_c}
        real["reference_cli_defaults"] = cli
        del rs100, rs
        aux["real_images"] = real
        aux["overlapping_synthetic"] = overlapping_leg(args, cds, ctx, dev, stream, mask, rects, failed)
        try:
            aux["variant_shard"] = variant_shard_leg(cds, ctx, fx, E.STD_RECTS, failed, n_t=2048 if n >= (1 << 20) else 256)
        except OSError as e:   # (the leg writes a ~0.6 GB shard into the temporary directory: no room there is the box's business, not a result)
            aux["variant_shard"] = {"skipped": f"{type(e).__name__}: {e}"}
        if not ok_r:
            failed.append("sparse and dense scans disagree on the reference's images")
        if not ok_f:
            failed.append("reference fixture pairs do not give the reference's JUnit values")
    aux["config0"] = config0_leg(args, cds, ctx, dev, stream, dense, mask, rects, failed)
    return aux


if __name__ == "__main__":
    main()
