"""The N>1 path on CPU: world_size-2 (and 3, uneven shards) gloo process groups exercise the sharding and the
single gather of colormipsearch_amd.dist.  The local scan is the CPU oracle here (a test stand-in for the HIP
kernel, which the -m gpu tests cover); the gathered result must equal a one-process scan of the whole library."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

W, H = 1210, 566
N_TOTAL = 11          # not divisible by 2 or 3: uneven shards


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from colormipsearch_amd import dist as cdist
        from oracle import cds_oracle as O
        rects = O.standard_label_rects(W)
        masks = [O.synth_image(0xBADC0DE, i, 2) for i in range(2)]
        masks[1] = np.fliplr(masks[1]).copy()
        b, e = cdist.shard_range(N_TOTAL, rank, world)
        # this rank's shard of the global library stream; the last global target is mask 0 itself
        shard = np.stack([O.synth_image(0xC0FFEE, i, 1) if i < N_TOTAL - 1 else masks[0] for i in range(b, e)]) if e > b else None
        queries = [O.PixelMatchQuery(m, 20, 20, 0.01, 2, True, rects) for m in masks]

        def local_scan(out):
            for m, q in enumerate(queries):
                if e > b:
                    s, mir = q.scan(shard, n_threads=2)
                    packed = (s.astype(np.int64) | (mir.astype(np.int64) << 31)).astype(np.uint32).view(np.int32)
                    out[m] = torch.from_numpy(packed.copy())

        got = cdist.sharded_search(local_scan, len(masks), N_TOTAL, torch.device("cpu"))
        # pair list scored by the owner of each target, gathered in pair order
        pm = np.array([0, 1, 1, 0, 0, 1], dtype=np.int32)
        pt = np.array([10, 0, 5, 3, 7, 10], dtype=np.int64)

        def local_score(mask_idx, slots):
            return torch.tensor([[int(mi) * 1000 + int(sl) + b, int(sl) + b, int(mi) & 1] for mi, sl in zip(mask_idx, slots)], dtype=torch.int64)

        pairs = cdist.sharded_pair_scores(local_score, pm, pt, N_TOTAL, torch.device("cpu"))
        # match lists: every rank filters ITS scores (score > 0 stands in for isMatch) into (mask, local target, packed) triples
        # ordered by mask, then target -- what the device compaction leaves -- and the lists are gathered instead of the matrix
        trip = []
        for m, q in enumerate(queries):
            if e > b:
                s, mir = q.scan(shard, n_threads=2)
                packed = (s.astype(np.int64) | (mir.astype(np.int64) << 31)).astype(np.uint32).view(np.int32)
                trip += [(m, int(t), int(packed[t])) for t in np.flatnonzero(s > 0)]
        local = torch.tensor(trip, dtype=torch.int32).reshape(-1, 3)
        lists = cdist.gather_match_lists(local, b)
        if rank == 0:
            ret["scores"] = got.numpy().copy()
            ret["pairs"] = pairs.numpy().copy()
            ret["matches"] = lists.numpy().copy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_equals_single_process(world, oracle):
    from colormipsearch_amd import dist as cdist
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + world + (os.getpid() % 200)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    rects = oracle.standard_label_rects(W)
    masks = [oracle.synth_image(0xBADC0DE, i, 2) for i in range(2)]
    masks[1] = np.fliplr(masks[1]).copy()
    library = np.stack([oracle.synth_image(0xC0FFEE, i, 1) for i in range(N_TOTAL - 1)] + [masks[0]])
    scores, mirrored = cdist.unpack_scores(ret["scores"])
    for m, mask in enumerate(masks):
        q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, 2, True, rects)
        ws, wm = q.scan(library, n_threads=4)
        assert np.array_equal(scores[m], ws)
        assert np.array_equal(mirrored[m], wm.astype(bool))
    assert scores[0, N_TOTAL - 1] > 0 and not mirrored[0, N_TOTAL - 1]
    assert scores[1, N_TOTAL - 1] > 0 and mirrored[1, N_TOTAL - 1]      # the flipped mask finds mask 0 mirrored
    pm = np.array([0, 1, 1, 0, 0, 1])
    pt = np.array([10, 0, 5, 3, 7, 10])
    assert np.array_equal(ret["pairs"], np.stack([pm * 1000 + pt, pt, pm & 1], axis=1))
    # the gathered match lists == the filter over the gathered score matrix, by mask, then library index
    want = [(m, int(t), int(ret["scores"][m, t])) for m in range(len(masks)) for t in np.flatnonzero(scores[m] > 0)]
    assert [tuple(int(v) for v in row) for row in ret["matches"]] == want and len(want) >= 2


def test_shard_ranges_partition_the_library():
    from colormipsearch_amd import dist as cdist
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            r = [cdist.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            if n:
                idx = np.unique(np.concatenate([np.arange(0, min(n, 50)), np.arange(max(0, n - 50), n), np.array([b for b, _ in r if b < n])]))
                own = cdist.owner_of(idx, n, world)
                for i, o in zip(idx, own):
                    assert r[o][0] <= i < r[o][1]
