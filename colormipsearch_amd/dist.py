"""Target-sharded multi-GPU search: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference's only parallel strategy is data parallelism over target partitions
(tools cmd/cdsprocess/LocalColorMIPSearchProcessor.java:93-97, SparkColorMIPSearchProcessor.java:50-73: partitions of
targets, results `collect()`-ed).  Here every (mask, target) score depends on that pair alone, so the library is cut
into `world` contiguous index ranges, each resident in one GPU's HBM; a search is a local scan per rank followed by
ONE gather of the per-target score arrays (int32, score | mirrored << 31) to rank 0 -- the only collective on the path
(4 MB per GPU per million targets: latency-, not bandwidth-bound on xGMI; no reduction, no ring).  For mask sets the
match lists are gathered instead of the score matrix (`gather_match_lists`: isMatch + compaction on every GPU).  The mask set is tiny
and replicated by the host.

torch is plumbing here (process group + device buffers); the scan itself is the HIP kernel behind the C ABI.
The functions take the local scan as a callable so that the sharding / gather logic is testable on CPU with gloo.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous shard [begin, end) of rank: GPU g owns [g*n/world, (g+1)*n/world)."""
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


def owner_of(index, n_items, world):
    """Rank owning library index `index` under shard_range (vectorised)."""
    index = np.asarray(index, dtype=np.int64)
    ends = np.array([(n_items * (r + 1)) // world for r in range(world)], dtype=np.int64)
    return np.searchsorted(ends, index, side="right")


def unpack_scores(packed):
    """int32 score | mirrored<<31  ->  (scores int32, mirrored bool)."""
    packed = np.asarray(packed).astype(np.int64) & 0xFFFFFFFF
    return (packed & 0x7FFFFFFF).astype(np.int32), ((packed >> 31) & 1).astype(bool)


def gather_scores(local_packed, n_total, group=None, dst=0):
    """local_packed: int32 tensor [n_masks, n_local] of THIS rank's shard (on the group's device type).
    Returns on rank `dst` an int32 tensor [n_masks, n_total] in library order, None elsewhere.
    One collective: a gather of equal-sized (padded) buffers."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_masks = local_packed.shape[0]
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    pad = max(sizes)
    assert local_packed.shape[1] == sizes[rank], (local_packed.shape, sizes[rank])
    send = local_packed
    if sizes[rank] != pad:
        send = torch.zeros((n_masks, pad), dtype=local_packed.dtype, device=local_packed.device)
        send[:, : sizes[rank]] = local_packed
    send = send.contiguous()
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([recv[r][:, : sizes[r]] for r in range(world)], dim=1)


def sharded_search(local_scan, n_masks, n_total, device, group=None, dst=0):
    """local_scan(out): fills the int32 tensor out[n_masks, n_local] (on `device`) with this rank's packed scores,
    asynchronously on the current stream.  Returns the gathered [n_masks, n_total] tensor on rank dst."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b, e = shard_range(n_total, rank, world)
    out = torch.empty((n_masks, e - b), dtype=torch.int32, device=device)
    local_scan(out)
    return gather_scores(out, n_total, group=group, dst=dst)


def gather_match_lists(local_matches, begin, group=None, dst=0):
    """local_matches: int32 tensor [k, 3] of THIS rank's match list -- (mask, target relative to the shard, score | mirrored << 31),
    ordered by mask, then target, as the device compaction leaves it -- on the group's device type; `begin` = the shard's first
    library index.  Returns on rank `dst` the int32 tensor [K, 3] of all shards' matches with library indices, ordered by mask,
    then target (the list one library holding every shard would give); None elsewhere.  Two collectives: the lengths (they are
    known only to their ranks), then one gather of the padded lists -- for a mask set this replaces the gather of the
    n_masks x n_total score matrix."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local_matches.device
    k = torch.tensor([local_matches.shape[0]], dtype=torch.int64, device=dev)
    ks = [torch.zeros_like(k) for _ in range(world)]
    dist.all_gather(ks, k, group=group)
    ks = [int(x.item()) for x in ks]
    pad = max(1, max(ks))
    send = torch.zeros((pad, 3), dtype=torch.int32, device=dev)
    if ks[rank]:
        send[: ks[rank]] = local_matches
        send[: ks[rank], 1] += int(begin)
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    parts = [recv[r][: ks[r]] for r in range(world)]
    cat = torch.cat(parts, dim=0)
    if cat.shape[0] == 0:
        return cat
    # every part is ordered by mask and the parts come in shard (= library) order: a stable sort by mask alone interleaves them
    order = torch.sort(cat[:, 0], stable=True).indices
    return cat[order]


class ShardedLibrary:
    """This rank's shard of an n_total-target library, resident in its GPU's HBM: dense slots (cds.Library over torch
    memory) or, with sparse=True, a cds.SparseLibrary (signal-pixel records; pixel-match scan only) filled through
    `stage` dense staging slots."""

    def __init__(self, width, height, n_total, context=None, group=None, sparse=False, records_per_target=24000, stage=1024, density_limit=None,
                 min_threshold=None):
        from . import cds
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.n_total = int(n_total)
        self.begin, self.end = shard_range(self.n_total, self.rank, self.world)
        self.ctx = context or cds.default_context()
        self.device = torch.device("cuda", self.ctx.device)
        self.sparse = bool(sparse)
        n_local = max(1, self.end - self.begin)
        n_dense = min(n_local, int(stage)) if self.sparse else n_local
        self._buf = torch.zeros(cds.Library.bytes_required(width, height, n_dense), dtype=torch.uint8, device=self.device)
        self.library = cds.Library(width, height, n_dense, context=self.ctx, device_buffer=self._buf)   # dense slots / staging
        self.sparse_library = None
        if self.sparse:
            self.sparse_library = cds.SparseLibrary(width, height, n_local, int(records_per_target) * n_local + 4 * height * n_local,
                                                    context=self.ctx, density_limit=density_limit, min_threshold=min_threshold)

    @property
    def n_local(self):
        return self.end - self.begin

    def generate_synthetic(self, seed, kind=0, with_variants=False):
        """Every rank regenerates its own index range of the same global stream."""
        if not self.n_local:
            return
        if not self.sparse:
            self.library.generate_synthetic(seed, self.begin, 0, self.n_local, kind=kind, with_variants=with_variants)
            return
        self.sparse_library.clear()
        for first in range(0, self.n_local, self.library.capacity):
            c = min(self.library.capacity, self.n_local - first)
            self.library.generate_synthetic(seed, self.begin + first, 0, c, kind=kind)
            self.sparse_library.append_from(self.library, 0, c)

    def upload(self, images):
        """This rank's shard from host images uint8[n_local, H, W, 3]."""
        if self.sparse:
            self.sparse_library.clear()
            self.sparse_library.append(images)
        else:
            self.library.upload(0, images)

    def search(self, algorithm, dst=0):
        """All masks of `algorithm` x the whole sharded library: gathered packed scores on rank dst."""
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

        def local_scan(out):
            if not self.n_local:
                return
            if self.sparse:
                algorithm.searchSparseLibraryDevice(self.sparse_library, 0, self.n_local, out.data_ptr())
            else:
                algorithm.searchLibraryDevice(self.library, 0, self.n_local, out.data_ptr())

        return sharded_search(local_scan, algorithm.n_masks, self.n_total, self.device, group=self.group, dst=dst)

    def search_matches(self, algorithm, pctPositivePixels, dst=0):
        """All masks x the whole sharded sparse library with ColorMIPSearch.isMatch and the compaction on every GPU: the gathered
        match list on rank dst (int32 [K, 3]: mask, library index, score | mirrored << 31; by mask, then index), None elsewhere."""
        if not self.sparse:
            raise ValueError("search_matches needs a sparse shard (the device match list is built over signal-pixel records)")
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        if self.n_local:
            m, t, sc, mir = algorithm.searchSparseLibraryMatches(self.sparse_library, pctPositivePixels, 0, self.n_local)
            packed = (sc.astype(np.int64) | (mir.astype(np.int64) << 31)).astype(np.uint32).view(np.int32)
            local = torch.from_numpy(np.stack([m, t, packed], axis=1).astype(np.int32)).to(self.device)
        else:
            local = torch.zeros((0, 3), dtype=torch.int32, device=self.device)
        return gather_match_lists(local, self.begin, group=self.group, dst=dst)


def sharded_pair_scores(local_score, pair_mask, pair_target, n_total, device, group=None, dst=0):
    """GradientAreaGap over a (mask, global target index) pair list: each rank scores the pairs whose target it owns
    (local_score(mask_idx, local_slots) -> int64 tensor [k, 3] = gap, high, mirrored), results gathered to rank dst in
    the original pair order as an int64 tensor [n_pairs, 3]."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    pair_mask = np.asarray(pair_mask, dtype=np.int32)
    pair_target = np.asarray(pair_target, dtype=np.int64)
    own = owner_of(pair_target, n_total, world)
    counts = [int((own == r).sum()) for r in range(world)]
    pad = max(1, max(counts))
    mine = np.flatnonzero(own == rank)
    b, _ = shard_range(n_total, rank, world)
    send = torch.zeros((pad, 3), dtype=torch.int64, device=device)
    if len(mine):
        send[: len(mine)] = local_score(pair_mask[mine], pair_target[mine] - b).to(device)
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    out = torch.zeros((len(pair_mask), 3), dtype=torch.int64, device=device)
    for r in range(world):
        idx = np.flatnonzero(own == r)
        if len(idx):
            out[torch.as_tensor(idx, device=device)] = recv[r][: len(idx)]
    return out


def gradscore_scorer(shape_algorithm, library):
    """The `local_score` of sharded_pair_scores backed by the HIP kernel: (mask indices, slots of THIS rank's dense shard) ->
    int64 tensor [k, 3] = gradientAreaGap, highExpressionArea, mirrored (cds_gradscore: one launch for the rank's pairs)."""

    def local_score(mask_idx, slots):
        gap, high, mir = shape_algorithm.scoreLibraryPairs(library, mask_idx, slots)
        return torch.from_numpy(np.stack([gap, high, mir.astype(np.int64)], axis=1))

    return local_score
