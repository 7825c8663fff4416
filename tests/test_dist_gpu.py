"""The sharded search on a real GPU through RCCL (world size 1 on the one-GPU box: the gather degenerates, the
plumbing -- torch memory, torch's stream, device score buffers, nccl process group -- is the multi-GPU one)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 1210, 566


def test_sharded_library_search_single_rank(oracle):
    import torch
    import torch.distributed as dist
    from colormipsearch_amd import cds
    from colormipsearch_amd import dist as cdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rects = cds.label_regions(W)
        n = 24
        shard = cdist.ShardedLibrary(W, H, n)
        assert (shard.begin, shard.end) == (0, n)
        shard.generate_synthetic(0xC0FFEE, kind=1)
        masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(2)])
        alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects, context=shard.ctx)
        packed = shard.search(alg)
        torch.cuda.synchronize()
        scores, mirrored = cdist.unpack_scores(packed.cpu().numpy())
        targets = shard.library.download(0, n)
        for m in range(2):
            ws, wm = oracle.PixelMatchQuery(masks[m], 20, 20, 0.01, 2, True, rects).scan(targets, n_threads=8)
            assert np.array_equal(scores[m], ws) and np.array_equal(mirrored[m], wm.astype(bool))
        assert scores.max() > 0
        # the same shard as a sparse resident library, generated through 10 staging slots (3 batches): identical gather
        sshard = cdist.ShardedLibrary(W, H, n, context=shard.ctx, sparse=True, records_per_target=W * H, stage=10)
        sshard.generate_synthetic(0xC0FFEE, kind=1)
        assert len(sshard.sparse_library) == n
        packed2 = sshard.search(alg)
        torch.cuda.synchronize()
        assert torch.equal(packed2, packed)
        # mask sets gather match lists instead of the score matrix: isMatch + compaction on the GPU, two small collectives
        lists = sshard.search_matches(alg, 1.0).cpu().numpy()
        sizes = alg.getQuerySizes()
        want = []
        for m in range(2):
            keep = cds.ColorMIPSearch(1.0, 20, None).isMatchArray(scores[m], int(sizes[m]))
            want += [(m, int(t), int(packed[m, t].item())) for t in np.flatnonzero(keep)]
        assert [tuple(int(v) for v in r) for r in lists] == want and len(want) > 0
    finally:
        shard.ctx.reset_stream()
        dist.destroy_process_group()


def test_native_sharded_search_through_the_c_abi(oracle):
    """cds_comm / cds_search_sparse_sharded (one process, one context per shard): (a) one context = the RCCL communicator
    (ncclCommInitAll + ncclGather with one rank), (b) three contexts on the one GPU = uneven shards through the
    device-to-device rehearsal path; both equal the single search over the whole library and the oracle."""
    from colormipsearch_amd import cds
    rects = cds.label_regions(W)
    n = 41
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(3)])
    targets = np.stack([oracle.synth_image(0xC0FFEE, 900 + i, 0) for i in range(n - 2)] + [masks[1], np.fliplr(masks[2]).copy()])

    def shard_of(ctx, imgs):
        sl = cds.SparseLibrary(W, H, max(1, len(imgs)), int((imgs.max(axis=3) > 0).sum()) + 4 * H * len(imgs) + 64, context=ctx)
        if len(imgs):
            sl.append(imgs)
        return sl

    want_s, want_m = [], []
    for m in range(3):
        ws, wm = oracle.PixelMatchQuery(masks[m], 20, 20, 0.01, 2, True, rects).scan(targets, n_threads=8)
        want_s.append(ws), want_m.append(wm.astype(bool))
    want_s, want_m = np.stack(want_s), np.stack(want_m)
    sizes = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects).getQuerySizes()
    cms = lambda pct: cds.ColorMIPSearch(pct, 20, None)
    for bounds in ([0, n], [0, 17, 17, n], [0, 5, 30, n]):      # 1 shard (RCCL); 3 shards, one of them empty; 3 uneven shards
        k = len(bounds) - 1
        ctxs = [cds.Context(0) for _ in range(k)]
        comm = cds.Comm(ctxs)
        assert len(comm) == k and comm.uses_rccl == (k == 1)
        algs = [cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects, context=c) for c in ctxs]
        libs = [shard_of(ctxs[r], targets[bounds[r]:bounds[r + 1]]) for r in range(k)]
        s, mir, offs = comm.searchSparseLibrarySharded(algs, libs)
        assert offs.tolist() == bounds
        assert np.array_equal(s, want_s) and np.array_equal(mir, want_m), bounds
        s2, _, _ = comm.searchSparseLibrarySharded(algs, libs)   # (the buffers are reused)
        assert np.array_equal(s2, want_s)
        # the match lists of the shards (isMatch + compaction on every GPU, one host thread per shard) == the host-side filter over
        # the whole score matrix, ordered by mask, then library index
        for pct in (0.0, 1.0):
            mm, tt, sc, mr, offs2 = comm.searchSparseLibraryShardedMatches(algs, libs, pct)
            assert offs2.tolist() == bounds
            rows = []
            for m in range(3):
                keep = cms(pct).isMatchArray(want_s[m], int(sizes[m]))
                rows += [(m, int(t), int(want_s[m, t]), bool(want_m[m, t])) for t in np.flatnonzero(keep)]
            assert list(zip(mm.tolist(), tt.tolist(), sc.tolist(), mr.tolist())) == rows, (bounds, pct)
        # top-k over the shards == a stable host selection over the whole score matrix (score descending, ties by library index)
        for k in (1, 5, n + 3):
            tk, sk, mk, _ = comm.searchSparseLibraryShardedTopK(algs, libs, k)
            for m in range(3):
                order = np.lexsort((np.arange(n), -want_s[m].astype(np.int64)))[:k]
                kk = len(order)
                assert np.array_equal(tk[m, :kk], order) and np.array_equal(sk[m, :kk], want_s[m][order]) and np.array_equal(mk[m, :kk], want_m[m][order]), (bounds, k, m)
                assert (tk[m, kk:] == -1).all()
        comm.close()
    assert want_s.max() > 0 and len(rows) > 0


def _build_rccl_stub():
    """tests/rccl_stub/rccl_stub.cpp -> librccl_stub.so (host code over the HIP runtime API; g++ is enough)."""
    import shutil
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_stub")
    src, out = os.path.join(here, "rccl_stub.cpp"), os.path.join(here, "librccl_stub.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        cxx = shutil.which("g++") or os.path.join(rocm, "bin", "amdclang++")
        subprocess.check_call([cxx, "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), src, "-o", out,
                               "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath," + os.path.join(rocm, "lib")],
                              stderr=subprocess.DEVNULL)
    return out


def test_the_rccl_branch_of_the_sharded_search_over_three_ranks_on_one_gpu(oracle, monkeypatch):
    """the code that will run on 8 GPUs had never executed with more than one rank.  With CDS_RCCL_LIBRARY pointing at
    the stand-in of tests/rccl_stub (ncclCommInitAll / GroupStart / Gather / GroupEnd / CommDestroy over hipMemcpyAsync on the callers'
    streams; it accepts duplicate devices and REJECTS what a real gather would hang or fault on) and CDS_COMM_FORCE_RCCL=1,
    cds_search_sparse_sharded takes its n > 1 RCCL branch over three contexts of the one GPU: communicator init over three ranks, one
    grouped gather with NULL receive buffers on the non-roots, the root's read-back behind the root's stream, the peers' streams
    synchronised afterwards, buffers reused by a second call and regrown by a larger one.  Scores equal the oracle's."""
    import ctypes
    from colormipsearch_amd import cds
    stub = _build_rccl_stub()
    monkeypatch.setenv("CDS_RCCL_LIBRARY", stub)
    monkeypatch.setenv("CDS_COMM_FORCE_RCCL", "1")
    probe = ctypes.CDLL(stub)
    probe.rccl_stub_gathers.restype = probe.rccl_stub_inits.restype = ctypes.c_ulong
    rects = cds.label_regions(W)
    n = 37
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(2)])
    targets = np.stack([oracle.synth_image(0xC0FFEE, 1200 + i, 0) for i in range(n - 2)] + [masks[0], np.fliplr(masks[1]).copy()])
    want_s, want_m = [], []
    for m in range(2):
        ws, wm = oracle.PixelMatchQuery(masks[m], 20, 20, 0.01, 2, True, rects).scan(targets, n_threads=8)
        want_s.append(ws), want_m.append(wm.astype(bool))
    want_s, want_m = np.stack(want_s), np.stack(want_m)

    def shard_of(ctx, imgs):
        sl = cds.SparseLibrary(W, H, max(1, len(imgs)), int((imgs.max(axis=3) > 0).sum()) + 4 * H * len(imgs) + 64, context=ctx)
        if len(imgs):
            sl.append(imgs)
        return sl

    g0, i0 = probe.rccl_stub_gathers(), probe.rccl_stub_inits()
    calls = 0
    for bounds in ([0, 9, 20, n], [0, 30, 30, n], [0, 1, 2, n]):      # uneven shards; an empty one in the middle; the largest shard last
        ctxs = [cds.Context(0) for _ in range(3)]
        comm = cds.Comm(ctxs)
        assert len(comm) == 3 and comm.uses_rccl
        algs = [cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects, context=c) for c in ctxs]
        libs = [shard_of(ctxs[r], targets[bounds[r]:bounds[r + 1]]) for r in range(3)]
        for _ in range(2):                                             # (the second call reuses the communicator's buffers)
            s, mir, offs = comm.searchSparseLibrarySharded(algs, libs)
            calls += 1
            assert offs.tolist() == bounds
            assert np.array_equal(s, want_s) and np.array_equal(mir, want_m), bounds
        # a larger search through the same communicator: send / receive buffers grow
        big = [shard_of(ctxs[r], np.concatenate([targets[bounds[r]:bounds[r + 1]]] * 2)) for r in range(3)]
        s, mir, offs = comm.searchSparseLibrarySharded(algs, big)
        calls += 1
        for r in range(3):
            a, b = bounds[r], bounds[r + 1]
            lo = int(offs[r])
            for rep in range(2):
                assert np.array_equal(s[:, lo + rep * (b - a): lo + (rep + 1) * (b - a)], want_s[:, a:b]), (bounds, r, rep)
        comm.close()
    assert probe.rccl_stub_gathers() - g0 == calls and probe.rccl_stub_inits() - i0 == 3    # the stand-in really carried every gather
    assert want_s.max() > 0


def _variant_shard(cds, ctx, seed, first, count):
    lib = cds.Library(W, H, max(1, count), context=ctx)
    if count:
        lib.generate_synthetic(seed, first, 0, count, kind=0, with_variants=True)
    return lib


def test_native_sharded_gradient_scores_through_the_c_abi(oracle):
    """cds_gradscore_sharded (kernel 2 over dense shards, one process): every (mask, library index) pair is scored on the context
    owning its target.  (a) one context = the communicator RCCL initialises, (b) three contexts on the one GPU with uneven and
    empty shards; results in pair order equal cds_gradscore over ONE library holding every target, and the CPU oracle on a sample.
    A slot without variants gives (-1, -1) on whichever shard it lives."""
    from colormipsearch_amd import cds
    rects = cds.label_regions(W)
    n, seed, base = 13, 0xC0FFEE, 500
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(3)])
    ctx0 = cds.default_context()
    whole = _variant_shard(cds, ctx0, seed, base, n)
    sh0 = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, 20, True, None, rects, context=ctx0)
    rng = np.random.default_rng(11)
    pm = rng.integers(0, 3, size=40).astype(np.int32)
    pt = rng.integers(0, n, size=40).astype(np.int64)
    want = sh0.scoreLibraryPairs(whole, pm, pt)
    assert (want[0] > 0).any() and want[2].any() and (~want[2]).any()       # (both orientations win somewhere)
    # the CPU oracle on a few of the pairs
    for i in (0, 7, 23):
        sq = oracle.ShapeQuery(masks[pm[i]], 20, True, rects)
        zg, gr = oracle.synth_variants(seed, base + int(pt[i]))
        og, oh, _, om = sq.score(oracle.synth_image(seed, base + int(pt[i]), 0), gr, zg)
        assert (int(want[0][i]), int(want[1][i]), bool(want[2][i])) == (og, oh, bool(om))
    for bounds in ([0, n], [0, 6, 6, n], [0, 2, 9, n]):
        k = len(bounds) - 1
        ctxs = [cds.Context(0) for _ in range(k)]
        comm = cds.Comm(ctxs)
        assert comm.uses_rccl == (k == 1)
        shapes = [cds.Shape2DMatchColorDepthSearchAlgorithm(masks, 20, True, None, rects, context=c) for c in ctxs]
        libs = [_variant_shard(cds, ctxs[r], seed, base + bounds[r], bounds[r + 1] - bounds[r]) for r in range(k)]
        sizes = [bounds[r + 1] - bounds[r] for r in range(k)]
        gap, high, mir, offs = comm.scoreLibraryPairsSharded(shapes, libs, pm, pt, shard_sizes=sizes)
        assert offs.tolist() == bounds
        assert np.array_equal(gap, want[0]) and np.array_equal(high, want[1]) and np.array_equal(mir, want[2]), bounds
        # no pairs at all, and pairs that all live on the last shard
        g0, _, _, _ = comm.scoreLibraryPairsSharded(shapes, libs, np.zeros(0, np.int32), np.zeros(0, np.int64), shard_sizes=sizes)
        assert len(g0) == 0
        last = np.flatnonzero(pt >= bounds[-2])
        g1, h1, m1, _ = comm.scoreLibraryPairsSharded(shapes, libs, pm[last], pt[last], shard_sizes=sizes)
        assert np.array_equal(g1, want[0][last]) and np.array_equal(h1, want[1][last]) and np.array_equal(m1, want[2][last])
        # errors: a target outside the library, a mask outside the set, a shard larger than its library
        for bad_m, bad_t, bad_sizes in (([0], [n], sizes), ([3], [0], sizes), ([0], [0], [s + 64 for s in sizes])):
            with pytest.raises(cds.CdsInvalidArgument):
                comm.scoreLibraryPairsSharded(shapes, libs, np.array(bad_m, np.int32), np.array(bad_t, np.int64), shard_sizes=bad_sizes)
        comm.close()
    # a shard whose slots have no variants: (-1, -1, False) for its pairs, the other shard's pairs unchanged
    ctxs = [cds.Context(0) for _ in range(2)]
    comm = cds.Comm(ctxs)
    shapes = [cds.Shape2DMatchColorDepthSearchAlgorithm(masks, 20, True, None, rects, context=c) for c in ctxs]
    bare = cds.Library(W, H, 6, context=ctxs[0])
    bare.generate_synthetic(seed, base, 0, 6, kind=0)
    libs = [bare, _variant_shard(cds, ctxs[1], seed, base + 6, n - 6)]
    gap, high, mir, _ = comm.scoreLibraryPairsSharded(shapes, libs, pm, pt, shard_sizes=[6, n - 6])
    lo = pt < 6
    assert (gap[lo] == -1).all() and (high[lo] == -1).all() and not mir[lo].any()
    assert np.array_equal(gap[~lo], want[0][~lo]) and np.array_equal(high[~lo], want[1][~lo]) and np.array_equal(mir[~lo], want[2][~lo])
    comm.close()


def test_sharded_pair_scores_with_the_real_kernel(oracle):
    """dist.sharded_pair_scores (one process per GPU, torch.distributed) driven by cds_gradscore: nccl with one rank on the one-GPU
    box -- the routing, padding, gather and reordering are the multi-rank code; the scorer is the HIP kernel, not a stand-in."""
    import torch
    import torch.distributed as dist
    from colormipsearch_amd import cds
    from colormipsearch_amd import dist as cdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29613"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rects = cds.label_regions(W)
        n, seed, base = 9, 0xC0FFEE, 700
        ctx = cds.default_context()
        masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(2)])
        lib = _variant_shard(cds, ctx, seed, base, n)
        shape = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, 20, True, None, rects, context=ctx)
        pm = np.array([0, 1, 1, 0, 0, 1, 0], dtype=np.int32)
        pt = np.array([8, 0, 5, 3, 7, 8, 0], dtype=np.int64)
        scorer = cdist.gradscore_scorer(shape, lib)
        got = cdist.sharded_pair_scores(scorer, pm, pt, n, torch.device("cuda", 0)).cpu().numpy()
        want = shape.scoreLibraryPairs(lib, pm, pt)
        assert np.array_equal(got[:, 0], want[0]) and np.array_equal(got[:, 1], want[1]) and np.array_equal(got[:, 2].astype(bool), want[2])
        sq = oracle.ShapeQuery(masks[1], 20, True, rects)
        zg, gr = oracle.synth_variants(seed, base + 5)
        og, oh, _, om = sq.score(oracle.synth_image(seed, base + 5, 0), gr, zg)
        assert (int(got[2, 0]), int(got[2, 1]), bool(got[2, 2])) == (og, oh, bool(om))
    finally:
        dist.destroy_process_group()


def test_two_distinct_gpus_through_rccl(oracle):
    """The branch no one-GPU box reaches: cds_comm over contexts of DIFFERENT devices (ncclCommInitAll over two GPUs, the grouped
    ncclGather with a NULL receive buffer on the non-root rank, per-device streams and host threads).  Skipped where fewer than two
    GPUs are visible -- the round's evidence for that branch is this test on the driver's multi-GPU node, nothing else."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (the one-GPU box rehearses the sharding logic through same-device contexts only)")
    from colormipsearch_amd import cds
    rects = cds.label_regions(W)
    n = 23
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(2)])
    targets = np.stack([oracle.synth_image(0xC0FFEE, 300 + i, 0) for i in range(n - 1)] + [masks[0]])
    want_s, want_m = [], []
    for m in range(2):
        ws, wm = oracle.PixelMatchQuery(masks[m], 20, 20, 0.01, 2, True, rects).scan(targets, n_threads=8)
        want_s.append(ws), want_m.append(wm.astype(bool))
    ctxs = [cds.Context(0), cds.Context(1)]
    comm = cds.Comm(ctxs)
    assert comm.uses_rccl and len(comm) == 2
    bounds = [0, 9, n]
    algs = [cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects, context=c) for c in ctxs]
    libs = []
    for r in range(2):
        imgs = targets[bounds[r]:bounds[r + 1]]
        sl = cds.SparseLibrary(W, H, len(imgs), int((imgs.max(axis=3) > 0).sum()) + 4 * H * len(imgs) + 64, context=ctxs[r])
        sl.append(imgs)
        libs.append(sl)
    s, mir, offs = comm.searchSparseLibrarySharded(algs, libs)
    assert offs.tolist() == bounds and np.array_equal(s, np.stack(want_s)) and np.array_equal(mir, np.stack(want_m))
    mm, tt, sc, mr, _ = comm.searchSparseLibraryShardedMatches(algs, libs, 0.0)
    assert len(mm) == int((np.stack(want_s) > 0).sum())
    # kernel 2 over the two GPUs
    shapes = [cds.Shape2DMatchColorDepthSearchAlgorithm(masks, 20, True, None, rects, context=c) for c in ctxs]
    dl = [_variant_shard(cds, ctxs[0], 0xC0FFEE, 500, 5), _variant_shard(cds, ctxs[1], 0xC0FFEE, 505, 4)]
    whole = _variant_shard(cds, ctxs[0], 0xC0FFEE, 500, 9)
    pm = np.array([0, 1, 1, 0, 1, 0], dtype=np.int32)
    pt = np.array([8, 0, 5, 4, 3, 6], dtype=np.int64)
    want = shapes[0].scoreLibraryPairs(whole, pm, pt)
    gap, high, gm, goffs = comm.scoreLibraryPairsSharded(shapes, dl, pm, pt, shard_sizes=[5, 4])
    assert goffs.tolist() == [0, 5, 9] and np.array_equal(gap, want[0]) and np.array_equal(high, want[1]) and np.array_equal(gm, want[2])
    comm.close()
