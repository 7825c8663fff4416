"""GpuColorMIPSearchProcessor -- the caller side of the two kernels.

Counterpart of the reference's
  * `ColorMIPSearchProcessor.findAllColorDepthMatches` (tools cmd/cdsprocess/ColorMIPSearchProcessor.java:8-12,
    LocalColorMIPSearchProcessor.java:55-116, AbstractColorMIPSearchProcessor.findPixelMatch :60-85): every mask against
    every target, keep the pairs for which `ColorMIPSearch.isMatch` holds;
  * `ColorMIPProcessUtils.selectBestMatches` (cmd/cdsprocess/ColorMIPProcessUtils.java:12-34) on top of
    `ItemsHandling.selectTopRankedElements` (API/results/ItemsHandling.java:80-109): top lines (publishedName) ->
    top samples per line (neuronId) -> top matches per sample;
  * `CalculateGradientScoresCmd`: shape scores of the selected pairs (:541-586) and the per-mask normalisation
    (:616-645, GradientAreaGapUtils.calculateNormalizedScore).

The per-pair Java loop becomes: one `cds_search` per mask batch over the resident library, the match filter (on the
device together with the compaction when the library is sparse: `cds_search_sparse_matches`; vectorised on the host for
dense slots), the (host) top-N selection over the few survivors, one `cds_gradscore` over the selected pairs.  Entities are plain
records here; `colormipsearch_amd/results.py` turns them into the reference's JSON result files (Mongo stays out of scope).
"""
from dataclasses import dataclass, field
import os
import threading
from typing import List, Optional, Sequence, Tuple

import time

import numpy as np

from . import cds


@dataclass
class NeuronInfo:
    """The three attributes of AbstractNeuronEntity the selection logic reads (API/model/AbstractNeuronEntity.java)."""
    mipId: str
    publishedName: Optional[str] = None
    neuronId: Optional[str] = None


@dataclass
class CDMatch:
    """The score fields of CDMatchEntity (API/model/CDMatchEntity.java:12-169)."""
    maskIndex: int
    targetIndex: int
    matchingPixels: int
    matchingPixelsRatio: float
    mirrored: bool
    gradientAreaGap: Optional[int] = None
    highExpressionArea: Optional[int] = None
    normalizedScore: Optional[float] = None
    errors: Optional[str] = None

    def hasGradScore(self):
        return (self.gradientAreaGap is not None and self.gradientAreaGap >= 0 and
                self.highExpressionArea is not None and self.highExpressionArea >= 0)

    def getGradScore(self):  # CDMatchEntity.java:75-84
        return cds.calculate2DShapeScore(self.gradientAreaGap, self.highExpressionArea) if self.hasGradScore() else -1


def java_string_hash(s: str) -> int:
    """java.lang.String.hashCode: s[0]*31^(n-1) + ... + s[n-1] over UTF-16 code units, as a signed 32-bit int."""
    h = 0
    b = s.encode("utf-16-be")
    for i in range(0, len(b), 2):
        h = (31 * h + ((b[i] << 8) | b[i + 1])) & 0xFFFFFFFF
    return h


def java_hashmap_order(keys: Sequence[str]) -> List[int]:
    """Iteration order of a java.util.HashMap<String, ?> into which the DISTINCT `keys` were put in this order: the
    permutation of range(len(keys)) that entrySet() yields.  Collectors.groupingBy collects into a HashMap
    (ItemsHandling.java:86-96), and the stable sort that follows (:97-103) leaves groups of equal score in this order --
    so it decides which of several tied lines / samples survive a topResults cut.
    Model: hash = h ^ (h >>> 16), bucket = hash & (capacity - 1); capacity 16, doubled while size > 0.75 * capacity;
    a bucket keeps insertion order (resize splits preserve it).  Not modelled: a bucket that Java turns into a tree
    (>= 8 colliding keys at capacity >= 64) iterates with the tree root first."""
    n = len(keys)
    cap = 16
    while n > (cap * 3) // 4:
        cap *= 2
    def bucket(k):
        h = java_string_hash(k)
        return (h ^ (h >> 16)) & (cap - 1)
    return sorted(range(n), key=lambda i: (bucket(keys[i]), i))


def selectTopRankedElements(items: Sequence, groupingCriteria, scoreExtractor, topResults: int, limitSubResults: int):
    """ItemsHandling.selectTopRankedElements (API/results/ItemsHandling.java:80-109): group, sort each group by
    descending score (stable), optionally cut each group, rank the groups by their best score (stable over the
    HashMap's iteration order, see java_hashmap_order), cut."""
    groups = {}
    for it in items:
        key = groupingCriteria(it)
        if key is None or str(key).strip() == "":
            key = "UNKNOWN"
        groups.setdefault(str(key), []).append(it)
    names = list(groups)
    groups = {names[i]: groups[names[i]] for i in java_hashmap_order(names)}
    ranked = []
    for key, lst in groups.items():
        lst = sorted(lst, key=lambda v: -float(scoreExtractor(v)))
        if 0 < limitSubResults < len(lst):
            lst = lst[:limitSubResults]
        ranked.append((key, max(float(scoreExtractor(v)) for v in lst), lst))
    ranked.sort(key=lambda e: -e[1])
    if 0 < topResults < len(ranked):
        ranked = ranked[:topResults]
    return ranked


@dataclass
class MatchTable:
    """Columnar form of a list of CDMatch records (numpy arrays of equal length): what the pipeline uses at scale -- a
    1024-mask x 1M-target search yields millions of matches, far too many for one Python object each."""
    maskIndex: np.ndarray
    targetIndex: np.ndarray
    matchingPixels: np.ndarray
    matchingPixelsRatio: np.ndarray            # float32, as PixelMatchScore.getNormalizedScore
    mirrored: np.ndarray
    gradientAreaGap: Optional[np.ndarray] = None     # int64, -1 = not computed
    highExpressionArea: Optional[np.ndarray] = None
    normalizedScore: Optional[np.ndarray] = None     # float32

    def __len__(self):
        return len(self.maskIndex)

    def take(self, idx):
        idx = np.asarray(idx, dtype=np.int64)
        opt = lambda a: None if a is None else a[idx]
        return MatchTable(self.maskIndex[idx], self.targetIndex[idx], self.matchingPixels[idx], self.matchingPixelsRatio[idx],
                          self.mirrored[idx], opt(self.gradientAreaGap), opt(self.highExpressionArea), opt(self.normalizedScore))

    def gradScores(self):
        """CDMatchEntity.getGradScore per row (API/model/CDMatchEntity.java:75-84): -1 where a score is missing."""
        g, h = self.gradientAreaGap, self.highExpressionArea
        ok = (g >= 0) & (h >= 0)
        return np.where(ok, g + h // 3, -1).astype(np.int64)

    def to_matches(self) -> List["CDMatch"]:
        out = []
        for i in range(len(self)):
            mt = CDMatch(int(self.maskIndex[i]), int(self.targetIndex[i]), int(self.matchingPixels[i]),
                         float(self.matchingPixelsRatio[i]), bool(self.mirrored[i]))
            if self.gradientAreaGap is not None and self.gradientAreaGap[i] != -1:
                mt.gradientAreaGap = int(self.gradientAreaGap[i])
            if self.highExpressionArea is not None and self.highExpressionArea[i] != -1:
                mt.highExpressionArea = int(self.highExpressionArea[i])
            if self.normalizedScore is not None:
                mt.normalizedScore = float(self.normalizedScore[i])
            out.append(mt)
        return out


def _group_codes(names):
    """(integer code per item, key string per code) for a grouping attribute; blank / missing -> "UNKNOWN"
    (ItemsHandling.java:84-88)."""
    keys = np.array(["UNKNOWN" if (n is None or str(n).strip() == "") else str(n) for n in names], dtype=object)
    uniq, codes = np.unique(keys, return_inverse=True)
    return codes.astype(np.int64), [str(k) for k in uniq]


def _key_hashes(group_keys):
    """Spread hash (h ^ h >>> 16) of every group key, once per key list (int64 array)."""
    h = np.array([java_string_hash(k) for k in group_keys], dtype=np.int64)
    return h ^ (h >> 16)


def _top_ranked(idx, group, score, top_results, limit_sub_results, group_keys, key_hashes=None):
    """Vectorised selectTopRankedElements over the items `idx` (their order = the input order): returns the kept items, groups
    ranked by their best score (descending, ties in the reference's HashMap iteration order: java_hashmap_order), items
    inside a group by descending score (ties in input order) -- exactly the order the list version produces."""
    if len(idx) == 0:
        return idx
    g, sc = group[idx], score[idx].astype(np.float64)
    first = np.full(g.max() + 1, len(idx), dtype=np.int64)
    np.minimum.at(first, g, np.arange(len(idx)))                      # first occurrence of each group = insertion into the map
    present = np.flatnonzero(first < len(idx))
    present = present[np.argsort(first[present], kind="stable")]      # the distinct groups in insertion order
    if key_hashes is None:
        key_hashes = _key_hashes(group_keys)
    cap = 16
    while len(present) > (cap * 3) // 4:
        cap *= 2
    hm = np.lexsort((np.arange(len(present)), key_hashes[present] & (cap - 1)))   # java_hashmap_order, vectorised
    first[present[hm]] = np.arange(len(present))                       # first[] now = position in the map's iteration
    order = np.lexsort((np.arange(len(idx)), -sc, first[g]))         # by group (map order), then score desc, stable
    g_sorted = g[order]
    starts = np.flatnonzero(np.r_[True, g_sorted[1:] != g_sorted[:-1]])
    rank_in_group = np.arange(len(idx)) - np.repeat(starts, np.diff(np.r_[starts, len(idx)]))
    keep = np.ones(len(idx), dtype=bool) if limit_sub_results <= 0 else rank_in_group < limit_sub_results
    best = sc[order][starts]                                          # best score of each group (groups in insertion order)
    grank = np.argsort(-best, kind="stable")                          # ranked groups
    if 0 < top_results < len(grank):
        grank = grank[:top_results]
    pos_of_group = np.full(len(starts), -1, dtype=np.int64)
    pos_of_group[grank] = np.arange(len(grank))
    gi = np.repeat(np.arange(len(starts)), np.diff(np.r_[starts, len(idx)]))
    sel = keep & (pos_of_group[gi] >= 0)
    final = np.lexsort((np.arange(len(idx))[sel], pos_of_group[gi][sel]))
    return idx[order][sel][final]


def _rank_groups(outer, code, key_hashes, score, seq, n_outer):
    """One level of selectTopRankedElements for MANY lists at once.  Items carry `outer` (which list they belong to, 0..n_outer-1),
    `code` (their group key), `score` and `seq` (their position in their list).  Returns per item (rank of its group inside
    its list, rank of the item inside its group): groups by best score descending, ties in the HashMap iteration order of that
    list's distinct keys (java_hashmap_order: bucket of the key's hash at the capacity the list's key count leads to, then
    insertion order = first appearance in the list); items by score descending, ties by position."""
    n = len(outer)
    pair, grp = np.unique(outer.astype(np.int64) * (int(code.max()) + 1 if n else 1) + code, return_inverse=True)
    ng = len(pair)
    g_outer = np.zeros(ng, dtype=np.int64); g_outer[grp] = outer
    g_code = np.zeros(ng, dtype=np.int64); g_code[grp] = code
    sc = score.astype(np.float64)
    best = np.full(ng, -np.inf); np.maximum.at(best, grp, sc)
    first = np.full(ng, np.iinfo(np.int64).max, dtype=np.int64); np.minimum.at(first, grp, seq)
    ndist = np.bincount(g_outer, minlength=n_outer)
    cap = np.full(n_outer, 16, dtype=np.int64)
    while True:
        grow = ndist > (cap * 3) // 4
        if not grow.any():
            break
        cap[grow] *= 2
    bucket = key_hashes[g_code] & (cap[g_outer] - 1)
    order = np.lexsort((first, bucket, -best, g_outer))
    starts = np.r_[0, np.cumsum(ndist)][:-1]
    grank = np.empty(ng, dtype=np.int64); grank[order] = np.arange(ng) - starts[g_outer[order]]
    io = np.lexsort((seq, -sc, grp))
    gsize = np.bincount(grp, minlength=ng)
    gstart = np.r_[0, np.cumsum(gsize)][:-1]
    irank = np.empty(n, dtype=np.int64); irank[io] = np.arange(n) - gstart[grp[io]]
    return grank[grp], irank


def _runs(flags):
    """[(start, end)) of the runs of True in a boolean array"""
    f = np.concatenate([[False], np.asarray(flags, dtype=bool), [False]])
    d = np.flatnonzero(f[1:] != f[:-1])
    return list(zip(d[0::2].tolist(), d[1::2].tolist()))


def normalizedScoresArray(pixelMatchScore, shapeScore, maxPixelMatch, maxShapeScore):
    """GradientAreaGapUtils.calculateNormalizedScore (API/cds/GradientAreaGapUtils.java:219-235) over arrays: the same double
    operations in the same order as cds_normalized_score (element-wise IEEE division / multiplication / min / max)."""
    pix = np.asarray(pixelMatchScore, dtype=np.float64)
    shape = np.asarray(shapeScore, dtype=np.float64)
    mp = np.broadcast_to(np.asarray(maxPixelMatch, dtype=np.float64), pix.shape)
    msh = np.broadcast_to(np.asarray(maxShapeScore, dtype=np.float64), pix.shape)
    plain = (pix == 0) | (mp == 0) | (shape < 0) | (msh <= 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        bounded = np.minimum(np.maximum((shape / msh) * 2.5, 0.002), 1.0)
        out = (pix / mp) / bounded * 100
    return np.where(plain, pix, out)


def normalizeTable(table: "MatchTable"):
    """CalculateGradientScoresCmd.normalizeScores (:616-645) for all masks of a table at once (normalizedScore as float32)."""
    grad = table.gradScores()
    masks, code = np.unique(table.maskIndex, return_inverse=True)
    max_pix = np.full(len(masks), -1, dtype=np.int64)
    max_grad = np.full(len(masks), -1, dtype=np.int64)
    np.maximum.at(max_pix, code, table.matchingPixels.astype(np.int64))
    np.maximum.at(max_grad, code, grad)
    table.normalizedScore = normalizedScoresArray(table.matchingPixels, grad, max_pix[code], max_grad[code]).astype(np.float32)
    return table


def resident_library_for(images, context=None, density_limit=0.25, min_threshold=0):
    """Layout choice for a library of host images uint8[n,H,W,3]: a `cds.SparseLibrary` (signal-pixel records) unless the
    images average more than `density_limit` of the frame in signal pixels -- the library itself refuses those
    (cds_sparse_library_set_density_limit) because the dense layout's scan is the faster one there -- in which case a dense
    `cds.Library` is returned.  Pass the result as `sparse_library=` or `library=` of GpuColorMIPSearchProcessor.
    `min_threshold`: the smallest data threshold the library's searches will use (pixels at or below it are not kept)."""
    imgs = np.ascontiguousarray(images, dtype=np.uint8)
    n, H, W, _ = imgs.shape
    signal = int((imgs.max(axis=3) > 0).sum())
    sl = None
    try:
        sl = cds.SparseLibrary(W, H, n, signal + 16 * H * n + 64, context=context, density_limit=density_limit, min_threshold=min_threshold)
        sl.append(imgs)
        return sl
    except cds.CdsError as e:
        if sl is not None:
            sl.close()          # (the refused library's record arena and staging slots go before the dense slots are allocated:
        if e.code != cds.capi.CDS_ERR_UNSUPPORTED:   # dense images are exactly where GPU memory is tightest)
            raise
    lib = cds.Library(W, H, n, context=context)
    lib.upload(0, imgs)
    return lib


class ShardedGpuColorMIPSearchProcessor:
    """colorDepthSearch over a target-sharded library in ONE process with several GPUs (`cds.Comm`): `shards[r]` is the
    `cds.SparseLibrary` of context r of the communicator (library index = shard offset + index inside the shard), the mask set is
    replicated, every GPU scans, filters (`ColorMIPSearch.isMatch`) and compacts its own shard, and the match lists come back merged
    by mask and library index (`cds_search_sparse_sharded_matches`) -- the reference's partitions of targets with their results
    collect()-ed (SparkColorMIPSearchProcessor.java:50-73).  `colorMIPSearch`'s provider must be a pixel-match provider of this
    package (it is re-bound to every context)."""

    def __init__(self, colorMIPSearch: cds.ColorMIPSearch, comm: cds.Comm, shards):
        assert len(shards) == len(comm.contexts)
        self.colorMIPSearch, self.comm, self.shards = colorMIPSearch, comm, list(shards)
        self.count = sum(len(s) for s in self.shards)

    def findAllColorDepthMatchesTable(self, masks, mask_set_size=128) -> MatchTable:
        """The rows of GpuColorMIPSearchProcessor.findAllColorDepthMatchesTable over one library holding all shards, same order."""
        masks = cds.capi.rgb_images(masks)
        thr = 0 if self.colorMIPSearch.defaultQueryThreshold is None else self.colorMIPSearch.defaultQueryThreshold
        providers = [self.colorMIPSearch.cdsAlgorithmProvider.forContext(c) for c in self.comm.contexts]
        parts = []
        for m0 in range(0, masks.shape[0], int(mask_set_size)):
            algs = [p.createColorDepthQuerySearchAlgorithmWithDefaultParams(masks[m0:m0 + int(mask_set_size)], thr, 0) for p in providers]
            try:
                m, t, sc, mir, _ = self.comm.searchSparseLibraryShardedMatches(algs, self.shards, self.colorMIPSearch.pctPositivePixels)
                sizes = np.asarray(algs[0].getQuerySizes(), dtype=np.float64)
            finally:
                for a in algs:
                    a.close()
            ratio = (sc.astype(np.float64) / np.maximum(sizes[m], 1.0)).astype(np.float32)
            parts.append(MatchTable(m.astype(np.int32) + np.int32(m0), t.astype(np.int64), sc.astype(np.int32), ratio, mir.astype(bool)))
        cat = lambda f: np.concatenate([getattr(p, f) for p in parts]) if parts else np.zeros(0)
        return MatchTable(cat("maskIndex"), cat("targetIndex"), cat("matchingPixels"), cat("matchingPixelsRatio"), cat("mirrored"))


class GpuColorMIPSearchProcessor:
    """masks: uint8[n_masks,H,W,3]; the targets live in a `cds.Library` (+ `targets` metadata, one NeuronInfo per slot).

    With `sparse_library` (a `cds.SparseLibrary` holding targets first..first+count-1 as signal-pixel records) the
    pixel-match scan runs over it -- the layout a production-size library is resident in -- and the dense `library` only
    has to hold the targets the shape scorer is asked about (with their gradient / zgap variants): `dense_slots[t]` = the
    dense slot of target t, default first + t.  This is the reference's own split: every target is scanned
    (colorDepthSearch), a few hundred per mask are loaded again with their variants (CalculateGradientScoresCmd)."""

    def __init__(self, colorMIPSearch: cds.ColorMIPSearch, library: cds.Library, targets: Optional[List[NeuronInfo]] = None,
                 first=0, count=None, sparse_library=None, dense_slots=None):
        self.colorMIPSearch = colorMIPSearch
        self.library = library
        self.sparse_library = sparse_library
        self.first = int(first)
        if count is None:
            count = (len(sparse_library) if sparse_library is not None else library.capacity) - self.first
        self.count = int(count)
        self.dense_slots = None if dense_slots is None else np.asarray(dense_slots, dtype=np.int64)
        self._targets = targets            # (default metadata -- one line and one sample per target -- is made on first use:
        assert targets is None or len(targets) == self.count   # a million-target library does not need a million records to be scanned)
        # The reference catches Throwable per (mask, target) pair, keeps the message in CDMatchEntity.errors and drops that pair only
        # (AbstractColorMIPSearchProcessor.java:68-83, LocalColorMIPSearchProcessor.java:102-107).  Here an image of the wrong size
        # spoils all pairs of its mask / its target and nothing else: one entry ("mask" | "target", index in the caller's list, message).
        self.errors: List[Tuple[str, int, str]] = []
        self._pools = None
        self._timing_lock = threading.Lock()

    # ---- lifecycle: the pipeline's worker threads live as long as the processor ------------------------
    def close(self):
        """Shut the pipeline's worker pools down (waits for running tasks).  The processor is a context manager."""
        pools, self._pools = self._pools, None
        if pools:
            for p in pools:
                p.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _library_size(self):
        lib = self.sparse_library if self.sparse_library is not None else self.library
        return int(lib.height), int(lib.width)

    def _usable_masks(self, masks):
        """(uint8[k,H,W,3] of the masks that have the library's image size, their indices in the caller's list).  A mask of another
        size is the reference's 'Invalid image size' error for every one of its pairs (PixelMatchColorDepthSearchAlgorithm.java:171-175):
        recorded in `errors`, dropped, and the other masks are searched."""
        if isinstance(masks, np.ndarray) and masks.ndim in (3, 4):
            m = cds.capi.rgb_images(masks)
            if tuple(m.shape[1:3]) == self._library_size():
                return m, np.arange(m.shape[0], dtype=np.int32)
            masks = list(m)
        H, W = self._library_size()
        keep, idx = [], []
        for i, m in enumerate(masks):
            a = np.asarray(m)
            if a.ndim == 3 and a.shape == (H, W, 3):
                keep.append(np.ascontiguousarray(a, dtype=np.uint8)), idx.append(i)
            else:
                shape = tuple(a.shape[:2][::-1]) if a.ndim >= 2 else tuple(a.shape)
                self.errors.append(("mask", i, f"Invalid image size - target's image size ({W}, {H}) must match query's image size: {shape}"))
        stacked = np.stack(keep) if keep else np.zeros((0, H, W, 3), dtype=np.uint8)
        return stacked, np.asarray(idx, dtype=np.int32)

    @property
    def targets(self):
        if self._targets is None:
            self._targets = [NeuronInfo(str(i), f"line{i}", f"sample{i}") for i in range(self.count)]
        return self._targets

    @targets.setter
    def targets(self, value):
        self._targets = value

    def _scan(self, alg):
        if self.sparse_library is not None:
            return alg.searchSparseLibrary(self.sparse_library, self.first, self.count)
        return alg.searchLibrary(self.library, self.first, self.count)

    def _slots(self, target_index):
        t = np.asarray(target_index, dtype=np.int64)
        return (self.first + t) if self.dense_slots is None else self.dense_slots[t]

    # ---- colorDepthSearch -------------------------------------------------------------------------
    def _device_match_table(self, alg, mask_offset=0) -> MatchTable:
        """Sparse library: scan, ColorMIPSearch.isMatch and the compaction run on the device (cds_search_sparse_match_list); only the
        match list comes back (rows ordered by mask, then target -- the order of the host-side filter below), already split into the
        table's columns by the library (cds_match_list_fetch_columns)."""
        m, t, sc, ratio, mir = alg.searchSparseLibraryMatchColumns(self.sparse_library, self.colorMIPSearch.pctPositivePixels, self.first, self.count,
                                                                   mask_offset=mask_offset)
        return MatchTable(m, t, sc, ratio, mir)

    def _pipelined_match_table(self, masks, mask_set_size, first_set_size=None) -> MatchTable:
        """Many masks against a sparse library: mask sets of `mask_set_size` (a multiple of 8, the multi-mask kernel's group size)
        go through three overlapped stages -- the host-side preparation of a set (acceptance intervals, planes, maps: about as
        long as its scan) on one thread, and two search workers taking the prepared sets in turn, so that the list download and the
        host-side columns of set i are made while set i + 1 is scanned (ctypes releases the GIL; the mask sets are independent
        objects and the context's stream orders their launches).  At most three prepared sets exist at a time."""
        from concurrent.futures import ThreadPoolExecutor, wait
        masks = cds.capi.rgb_images(masks)
        n = masks.shape[0]
        make = self.colorMIPSearch.createQueryColorDepthSearchWithDefaultThreshold
        if n <= mask_set_size:
            return self._device_match_table(make(masks))
        ahead = threading.Semaphore(3)
        self.last_pipeline = {"prepare_s": 0.0, "search_s": 0.0}

        def prepare(m0, m1):
            ahead.acquire()
            t0 = time.perf_counter()
            try:
                return make(masks[m0:m1])
            except BaseException:
                ahead.release()
                raise
            finally:
                with self._timing_lock:
                    self.last_pipeline["prepare_s"] += time.perf_counter() - t0

        def search(m0, prepared):
            alg = prepared.result()
            t0 = time.perf_counter()
            try:
                t = self._device_match_table(alg, mask_offset=m0)
            finally:
                alg.close()
                ahead.release()
            with self._timing_lock:     # (two search workers)
                self.last_pipeline["search_s"] += time.perf_counter() - t0
            return t

        # (the device idles while the FIRST set is prepared: that one is a quarter of the size -- a multiple of the kernel's group of 8; a
        # preparation has ~15 ms of fixed cost, smaller or more steps were measured and gain nothing)
        sizes = [mask_set_size]
        if first_set_size:
            sizes = [min(mask_set_size, max(8, int(first_set_size) // 8 * 8))]
        elif n >= 3 * mask_set_size and mask_set_size >= 32:
            sizes = [mask_set_size // 4 // 8 * 8]
        starts, at = [], 0
        for k in sizes:
            if at < n:
                starts.append(at)
                at += k
        starts += list(range(at, n, mask_set_size))
        ends = starts[1:] + [n]
        # (the pools live as long as the processor: a host thread that has used the GPU runtime is slow to exit -- ~100 ms for the three
        # of them after every call, measured --, and a processor runs many searches)
        if self._pools is None:
            self._pools = (ThreadPoolExecutor(max_workers=1, thread_name_prefix="cds-prepare"), ThreadPoolExecutor(max_workers=2, thread_name_prefix="cds-search"))
        prep_pool, search_pool = self._pools
        prepared = [prep_pool.submit(prepare, m0, m1) for m0, m1 in zip(starts, ends)]
        searches = [search_pool.submit(search, m0, pf) for m0, pf in zip(starts, prepared)]
        try:
            parts = [f.result() for f in searches]
        except BaseException:
            # nothing of this call may still touch the library once the error is handed to the caller (who may close it next):
            # what has not started is cancelled, what runs is waited for, prepared sets nobody will search are closed
            for f in searches + prepared:
                f.cancel()
            closed = set()
            while True:
                for i, (sf, pf) in enumerate(zip(searches, prepared)):   # (their permits let a preparation that is already running finish)
                    if i not in closed and sf.cancelled() and pf.done() and not pf.cancelled() and pf.exception() is None:
                        closed.add(i)
                        pf.result().close()
                        ahead.release()
                running = [f for f in searches + prepared if not f.done()]
                if not running:
                    break
                wait(running, timeout=0.05)
            raise
        t0 = time.perf_counter()
        rows = sum(len(t) for t in parts)
        out = MatchTable(np.empty(rows, np.int32), np.empty(rows, np.int64), np.empty(rows, np.int32), np.empty(rows, np.float32), np.empty(rows, bool))
        def fill(f):   # (one thread per column: the copies release the GIL; tens of millions of rows for 1,024 masks x 1 M targets)
            at, col = 0, getattr(out, f)
            for t in parts:
                col[at:at + len(t)] = getattr(t, f)
                at += len(t)
        with ThreadPoolExecutor(max_workers=5) as pool:
            list(pool.map(fill, ("maskIndex", "targetIndex", "matchingPixels", "matchingPixelsRatio", "mirrored")))
        self.last_pipeline["assemble_s"] = time.perf_counter() - t0
        return out

    def findAllColorDepthMatches(self, masks) -> List[CDMatch]:
        masks, orig = self._usable_masks(masks)
        if masks.shape[0] == 0:
            return []
        alg = self.colorMIPSearch.createQueryColorDepthSearchWithDefaultThreshold(masks)
        if self.sparse_library is not None:
            out = self._device_match_table(alg).to_matches()
            for r in out:
                r.maskIndex = int(orig[r.maskIndex])
            return out
        scores, mirrored = self._scan(alg)
        sizes = alg.getQuerySizes()
        out = []
        for m in range(alg.n_masks):
            if sizes[m] == 0:  # "No computation created ... because it is empty" (LocalColorMIPSearchProcessor.java:88-92)
                continue
            keep = self.colorMIPSearch.isMatchArray(scores[m], int(sizes[m]))
            ratio = (scores[m].astype(np.float64) / float(sizes[m])).astype(np.float32)
            for t in np.flatnonzero(keep):
                out.append(CDMatch(int(orig[m]), int(t), int(scores[m, t]), float(ratio[t]), bool(mirrored[m, t])))
        return out

    def findAllColorDepthMatchesTable(self, masks, mask_set_size=128, first_set_size=None) -> MatchTable:
        """findAllColorDepthMatches in columnar form (same rows, same order).  Over a sparse library the masks go in sets of
        `mask_set_size` whose preparation overlaps the previous set's scan; the first set is smaller (`first_set_size`, default a
        quarter of a set when there are at least three sets) so that the device starts early."""
        masks, orig = self._usable_masks(masks)
        if masks.shape[0] == 0:
            return MatchTable(np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32), np.zeros(0, bool))
        renumber = not np.array_equal(orig, np.arange(len(orig)))
        if self.sparse_library is not None:
            t = self._pipelined_match_table(masks, max(8, int(mask_set_size) // 8 * 8), first_set_size)
            if renumber:
                t.maskIndex = orig[t.maskIndex]
            return t
        alg = self.colorMIPSearch.createQueryColorDepthSearchWithDefaultThreshold(masks)
        scores, mirrored = self._scan(alg)
        sizes = np.asarray(alg.getQuerySizes())
        mi, ti, sc, ra, mr = [], [], [], [], []
        for m in range(alg.n_masks):
            if sizes[m] == 0:
                continue
            t = np.flatnonzero(self.colorMIPSearch.isMatchArray(scores[m], int(sizes[m])))
            mi.append(np.full(len(t), orig[m], dtype=np.int32)); ti.append(t.astype(np.int64)); sc.append(scores[m, t].astype(np.int32))
            ra.append((scores[m, t].astype(np.float64) / float(sizes[m])).astype(np.float32)); mr.append(mirrored[m, t].astype(bool))
        cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dtype=dt)
        return MatchTable(cat(mi, np.int32), cat(ti, np.int64), cat(sc, np.int32), cat(ra, np.float32), cat(mr, bool))

    # ---- a library larger than the resident capacity: pass after pass over resident chunks ---------------
    def _append_usable_targets(self, images, first_index):
        """Append the images of a chunk that have the library's size to the sparse library; the others (another size, None: a file that
        could not be read) are their targets' errors -- the reference's per-pair catch (AbstractColorMIPSearchProcessor.java:68-83) --
        and are left out.  Returns the caller-side index of every appended image."""
        H, W = self._library_size()
        if isinstance(images, np.ndarray) and images.ndim == 4 and tuple(images.shape[1:]) == (H, W, 3):
            self.sparse_library.append(images)
            return first_index + np.arange(images.shape[0], dtype=np.int64)
        keep, idx = [], []
        for i, img in enumerate(images):
            a = None if img is None else np.asarray(img)
            if a is not None and a.shape == (H, W, 3):
                keep.append(np.ascontiguousarray(a, dtype=np.uint8)), idx.append(first_index + i)
            elif a is None:
                self.errors.append(("target", first_index + i, "no image"))
            else:
                shape = tuple(a.shape[:2][::-1]) if a.ndim >= 2 else tuple(a.shape)
                self.errors.append(("target", first_index + i, f"Invalid image size - target's image size {shape} must match query's image size: ({W}, {H})"))
        if keep:
            self.sparse_library.append(np.stack(keep))
        return np.asarray(idx, dtype=np.int64)

    def findAllColorDepthMatchesOverChunksTable(self, masks, n_targets, load_chunk, chunk_targets, mask_set_size=128) -> MatchTable:
        """colorDepthSearch over a library that does not fit the resident capacity.  The reference walks a library of any size in
        partitions through its MIP cache (LocalColorMIPSearchProcessor.java:93-97, CachedMIPsUtils.java:60-100); here the sparse
        library is the resident window: chunk k = targets [k * chunk_targets, ...) is made resident (`load_chunk(first, count)` ->
        uint8[count,H,W,3], or a list of `count` images in which an entry of another size / None is that target's error), ALL mask
        sets scan it (the pipelined search above), the next chunk replaces it, and the match lists are merged into the order one
        resident library would give: by mask, then target index.  A real LM MIP costs ~150 KB as records (~220 KB with the 16-bit
        stream), so one MI355X holds about 1.2-1.8 M of them; beyond that, or for a deliberately small window, this is the loop."""
        if self.sparse_library is None:
            raise ValueError("the chunked search keeps its resident window in a sparse library")
        n_targets, chunk_targets = int(n_targets), max(1, int(chunk_targets))
        masks, orig = self._usable_masks(masks)      # (once: a mask of another size is one error, not one per chunk)
        parts, first0, count0 = [], self.first, self.count
        self.last_chunks = []
        try:
            for c0 in range(0, n_targets, chunk_targets):
                cn = min(chunk_targets, n_targets - c0)
                t0 = time.perf_counter()
                self.sparse_library.clear()
                ids = self._append_usable_targets(load_chunk(c0, cn), c0)
                t1 = time.perf_counter()
                self.first, self.count = 0, len(ids)
                if len(ids) and masks.shape[0]:
                    t = self.findAllColorDepthMatchesTable(masks, mask_set_size=mask_set_size)
                    t.targetIndex = ids[t.targetIndex]
                    t.maskIndex = orig[t.maskIndex]
                    parts.append(t)
                self.last_chunks.append({"first": c0, "targets": cn, "resident": int(len(ids)), "load_s": t1 - t0, "search_s": time.perf_counter() - t1})
        finally:
            self.first, self.count = first0, count0
        cat = lambda f, dt: np.concatenate([getattr(p, f) for p in parts]) if parts else np.zeros(0, dtype=dt)
        out = MatchTable(cat("maskIndex", np.int32), cat("targetIndex", np.int64), cat("matchingPixels", np.int32),
                         cat("matchingPixelsRatio", np.float32), cat("mirrored", bool))
        # every chunk's rows are ordered by mask, then target, and the chunks ascend: a STABLE sort by mask gives (mask, target)
        return out.take(np.argsort(out.maskIndex, kind="stable"))

    def selectBestMatchesTable(self, table: MatchTable, topLineMatches, topSamplesPerLine, topMatchesPerSample) -> MatchTable:
        """selectBestMatches in columnar form, all masks at once: per mask top lines -> top samples per line -> top matches per
        sample (two levels of ItemsHandling.selectTopRankedElements; same rows in the same order as the list version)."""
        n = len(table)
        if n == 0:
            return table.take(np.zeros(0, dtype=np.int64))
        lcode, lkeys = _group_codes([t.publishedName for t in self.targets])
        scode, skeys = _group_codes([t.neuronId for t in self.targets])
        lines, samples = lcode[table.targetIndex], scode[table.targetIndex]
        score = table.matchingPixels
        masks, mcode = np.unique(table.maskIndex, return_inverse=True)          # lists of level 1: one per mask, ascending
        mstart = np.r_[0, np.cumsum(np.bincount(mcode))][:-1]
        order0 = np.lexsort((np.arange(n), mcode))                               # items of a mask in table order
        seq0 = np.empty(n, dtype=np.int64); seq0[order0] = np.arange(n) - mstart[mcode[order0]]
        lrank, lirank = _rank_groups(mcode, lines, _key_hashes(lkeys), score, seq0, len(masks))
        keep = np.ones(n, dtype=bool) if topLineMatches <= 0 else lrank < topLineMatches
        idx = np.flatnonzero(keep)
        # level 2: one list per kept (mask, line), its items in the order level 1 left them in (score descending, stable)
        pair, pcode = np.unique(mcode[idx] * (int(lrank.max()) + 1) + lrank[idx], return_inverse=True)
        srank, sirank = _rank_groups(pcode, samples[idx], _key_hashes(skeys), score[idx], lirank[idx], len(pair))
        keep2 = np.ones(len(idx), dtype=bool)
        if topSamplesPerLine > 0:
            keep2 &= srank < topSamplesPerLine
        if topMatchesPerSample > 0:
            keep2 &= sirank < topMatchesPerSample
        sel = np.flatnonzero(keep2)
        final = np.lexsort((sirank[sel], srank[sel], pcode[sel]))                 # mask, line rank, sample rank, item rank
        return table.take(idx[sel][final])

    def calculateGradientScoresTable(self, masks, table: MatchTable, queryThreshold, mirrorMask=True, roiMask=None,
                                     excludedRegions=cds.label_regions) -> MatchTable:
        """Shape scores for the table's pairs (one launch) and the per-mask normalisation, in columnar form."""
        n = len(table)
        table.gradientAreaGap = np.full(n, -1, dtype=np.int64)
        table.highExpressionArea = np.full(n, -1, dtype=np.int64)
        table.normalizedScore = np.zeros(n, dtype=np.float32)
        if n == 0:
            return table
        alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, queryThreshold, mirrorMask, roiMask, excludedRegions,
                                                        context=self.library.ctx)
        gap, high, _ = alg.scoreLibraryPairs(self.library, table.maskIndex.astype(np.int32), self._slots(table.targetIndex))
        table.gradientAreaGap[:] = gap
        table.highExpressionArea[:] = high
        return normalizeTable(table)                             # CalculateGradientScoresCmd.normalizeScores (:616-645)

    def calculateGradientScoresBatchedTable(self, masks, table: MatchTable, load_range, queryThreshold, mirrorMask=True, roiMask=None,
                                            excludedRegions=cds.label_regions, batch=8192, library=None) -> MatchTable:
        """calculateGradientScoresTable for a library far larger than the dense slots the shape scorer reads (BASELINE configs[4]:
        1,024 masks x top-500 of a 1 M-target library resident as signal-pixel records): the table's pairs are visited in windows
        of `batch` consecutive target indices (the table's targetIndex is relative to `self.first`, like every table this class
        makes); `load_range(lib, first_target, count)` makes LIBRARY targets first_target..+count-1 (absolute: self.first is
        added here) resident WITH their gradient / zgap variants in slots 0..count-1 of `lib` (regenerated on the device for a synthetic
        library: `lambda lib, f, c: lib.generate_synthetic(seed, f, 0, c, kind=0, with_variants=True)`; ingested from a shard
        or from files otherwise), one `cds_gradscore` launch per window, per-mask normalisation on the host
        (CalculateGradientScoresCmd.java:541-586, 616-645).  The kernel time of the launches is left in
        `self.last_gradscore = {kernel_ms, launches, pairs, mask_preparation_s, load_range_s, score_calls_s}` (host seconds of the
        shape-mask preparation, of the `load_range` callbacks and of the scoring calls)."""
        n = len(table)
        table.gradientAreaGap = np.full(n, -1, dtype=np.int64)
        table.highExpressionArea = np.full(n, -1, dtype=np.int64)
        table.normalizedScore = np.zeros(n, dtype=np.float32)
        self.last_gradscore = {"kernel_ms": 0.0, "launches": 0, "pairs": 0, "mask_preparation_s": 0.0, "load_range_s": 0.0, "score_calls_s": 0.0}
        if n == 0:
            return table
        ctx = library.ctx if library is not None else (self.library.ctx if self.library is not None else self.sparse_library.ctx)
        t0 = time.perf_counter()
        alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, queryThreshold, mirrorMask, roiMask, excludedRegions, context=ctx)
        self.last_gradscore["mask_preparation_s"] = time.perf_counter() - t0
        lib = library if library is not None else cds.Library(alg.width, alg.height, int(batch), context=ctx)
        batch = min(int(batch), lib.capacity)
        order = np.argsort(table.targetIndex, kind="stable")
        tsorted = table.targetIndex[order]
        limit = self.count                      # (relative, like targetIndex and the windows below)
        w0 = (int(tsorted[0]) // batch) * batch
        while w0 <= int(tsorted[-1]):
            lo, hi = np.searchsorted(tsorted, w0), np.searchsorted(tsorted, w0 + batch)
            if hi > lo:
                rows = order[lo:hi]
                t0 = time.perf_counter()
                load_range(lib, self.first + w0, min(batch, limit - w0))
                ctx.synchronize()
                t1 = time.perf_counter()
                gap, high, _ = alg.scoreLibraryPairs(lib, table.maskIndex[rows].astype(np.int32), table.targetIndex[rows] - w0)
                self.last_gradscore["load_range_s"] += t1 - t0
                self.last_gradscore["score_calls_s"] += time.perf_counter() - t1
                table.gradientAreaGap[rows] = gap
                table.highExpressionArea[rows] = high
                ms, launches = alg.lastKernelMs()
                self.last_gradscore["kernel_ms"] += ms * launches
                self.last_gradscore["launches"] += launches
                self.last_gradscore["pairs"] += len(rows)
            w0 += batch
        return normalizeTable(table)

    def calculateGradientScoresSelectedTable(self, masks, table: MatchTable, load_targets, queryThreshold, mirrorMask=True, roiMask=None,
                                             excludedRegions=cds.label_regions, batch=8192, library=None) -> MatchTable:
        """The gradient-score step when only the SELECTED targets are made resident (what the reference does: it opens the variant files
        of the matches it scores, CalculateGradientScoresCmd.java:541-586): the distinct targets of the table's pairs are visited
        `batch` at a time; `load_targets(lib, library_indices)` (int64 array, absolute: self.first added) fills slots 0..k-1 of `lib` with
        those targets AND their gradient / zgap variants; one `cds_gradscore` launch per batch; per-mask normalisation on the host.
        Against calculateGradientScoresBatchedTable (whole index windows): for BASELINE configs[4] 230 K distinct targets are loaded
        instead of 1 M.  `self.last_gradscore` = {kernel_ms, launches, pairs, distinct_targets, mask_preparation_s, load_s, score_calls_s,
        conversion_ms, converted_slots, launches_on_zslice_plane}."""
        n = len(table)
        table.gradientAreaGap = np.full(n, -1, dtype=np.int64)
        table.highExpressionArea = np.full(n, -1, dtype=np.int64)
        table.normalizedScore = np.zeros(n, dtype=np.float32)
        st = {"kernel_ms": 0.0, "launches": 0, "pairs": 0, "distinct_targets": 0, "mask_preparation_s": 0.0, "load_s": 0.0, "score_calls_s": 0.0,
              "conversion_ms": 0.0, "converted_slots": 0, "launches_on_zslice_plane": 0}
        self.last_gradscore = st
        if n == 0:
            return table
        ctx = library.ctx if library is not None else (self.library.ctx if self.library is not None else self.sparse_library.ctx)
        t0 = time.perf_counter()
        alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, queryThreshold, mirrorMask, roiMask, excludedRegions, context=ctx)
        st["mask_preparation_s"] = time.perf_counter() - t0
        lib = library if library is not None else cds.Library(alg.width, alg.height, int(batch), context=ctx)
        batch = min(int(batch), lib.capacity)
        distinct, inv = np.unique(table.targetIndex, return_inverse=True)
        st["distinct_targets"] = int(len(distinct))
        order = np.argsort(inv, kind="stable")
        isorted = inv[order]
        for b0 in range(0, len(distinct), batch):
            idx = distinct[b0:b0 + batch]
            lo, hi = np.searchsorted(isorted, b0), np.searchsorted(isorted, b0 + len(idx))
            rows = order[lo:hi]
            t0 = time.perf_counter()
            load_targets(lib, self.first + idx.astype(np.int64))
            ctx.synchronize()
            t1 = time.perf_counter()
            gap, high, _ = alg.scoreLibraryPairs(lib, table.maskIndex[rows].astype(np.int32), (inv[rows] - b0).astype(np.int64))
            st["load_s"] += t1 - t0
            st["score_calls_s"] += time.perf_counter() - t1
            table.gradientAreaGap[rows] = gap
            table.highExpressionArea[rows] = high
            ms, launches = alg.lastKernelMs()
            plane, conv, cms = alg.lastCallInfo()
            st["kernel_ms"] += ms * launches
            st["launches"] += launches
            st["pairs"] += len(rows)
            st["conversion_ms"] += cms
            st["converted_slots"] += conv
            st["launches_on_zslice_plane"] += int(plane)
        return normalizeTable(table)

    def calculateGradientScoresFromFilesTable(self, masks, table: MatchTable, target_files, gradient_files, zgap_files, queryThreshold,
                                              mirrorMask=True, roiMask=None, excludedRegions=cds.label_regions, batch=256, threads=8,
                                              zgap_radius=10) -> MatchTable:
        """calculateGradientScoresTable for a library that is resident only as a sparse library: the targets the table names are
        read from files the way CalculateGradientScoresCmd does it through its variant suppliers (:541-586) -- `target_files[t]`,
        `gradient_files[t]`, `zgap_files[t]` (file_data as mipio.read_mip takes it; a missing gradient (None) leaves the pair at
        (-1, -1) like Shape2DMatch...Algorithm.java:155-158; a missing zgap is synthesised on the device from the target as the
        reference's tests do, Shape2DMatch...Test.java:171-174).  `batch` distinct targets at a time are decoded natively on a
        thread pool, uploaded into a dense library of `batch` slots and scored with one launch."""
        from concurrent.futures import ThreadPoolExecutor
        from . import mipio
        n = len(table)
        table.gradientAreaGap = np.full(n, -1, dtype=np.int64)
        table.highExpressionArea = np.full(n, -1, dtype=np.int64)
        table.normalizedScore = np.zeros(n, dtype=np.float32)
        if n == 0:
            return table
        ctx = self.library.ctx if self.library is not None else self.sparse_library.ctx
        alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, queryThreshold, mirrorMask, roiMask, excludedRegions, context=ctx)
        W, H = alg.width, alg.height
        regions = cds._resolve_regions(excludedRegions, W)
        distinct = np.unique(table.targetIndex)
        is_path = cds.is_native_path                 # a path, or the reference's zipEntry FileData as (archive.zip, entry name)
        # Round 5: plain files go through the library's own batch loader (cds_library_load_files: the three files of every target decoded on
        # native threads straight into pinned memory, 64 slots in flight behind the decode) and batch i + 1 is loaded into a second dense
        # library while batch i is scored -- 0.70 K targets/s became 4.5 K (tools/gradscore_files_rate.py: 4,096 targets, 16 cores); what is left
        # is inflating the 16-bit gradient PNGs (1.4 MB of pixels each), and it scales with the cores.  Anything else (image
        # bytes, file objects, archives that are not zip files: mipio.read_mip takes them) keeps the per-item path.
        native = all(is_path(target_files[int(t)]) and is_path(gradient_files[int(t)]) and (zgap_files is None or is_path(zgap_files[int(t)])) for t in distinct)

        def load_native(lib, chunk):
            """chunk's targets that have a gradient -> slots 0.. of `lib`; returns their slot numbers (-1: not loaded)"""
            keep = [int(t) for t in chunk if gradient_files[int(t)] is not None]
            slot = np.full(len(chunk), -1, dtype=np.int64)
            slot[[i for i, t in enumerate(chunk) if gradient_files[int(t)] is not None]] = np.arange(len(keep))
            if keep:
                zp = None if zgap_files is None else [zgap_files[t] for t in keep]
                lib.load_files(0, [target_files[t] for t in keep], [gradient_files[t] for t in keep], zp, threads=threads)
                no_z = np.array([zgap_files is None or zgap_files[t] is None for t in keep])
                for r0, r1 in _runs(no_z):                      # a missing zgap is made on the device from the target, as the reference's tests do
                    lib.make_zgap_variants(r0, r1 - r0, queryThreshold, zgap_radius, regions)
            return slot

        def load_items(lib, chunk, pool):
            def load(t):
                g = gradient_files[t]
                if g is None:
                    return None
                z = zgap_files[t] if zgap_files is not None else None
                return mipio.read_mip(target_files[t]), mipio.read_gradient(g), (None if z is None else mipio.read_mip(z))
            loaded = list(pool.map(load, [int(t) for t in chunk]))     # (the native decoder releases the GIL)
            slot = np.full(len(chunk), -1, dtype=np.int64)
            k = 0
            for i, item in enumerate(loaded):
                if item is None:
                    continue
                slot[i] = k
                lib.upload(k, item[0][None])
                lib.upload_variants(k, gradients=item[1][None], zgaps=None if item[2] is None else item[2][None])
                if item[2] is None:
                    lib.make_zgap_variants(k, 1, queryThreshold, zgap_radius, regions)
                k += 1
            return slot

        with ThreadPoolExecutor(max_workers=max(1, int(threads))) as pool:
            load = load_native if native else (lambda lib, chunk: load_items(lib, chunk, pool))
            info = self._score_in_batches(alg, table, ctx, batch, load, overlap=native)
        self.last_from_files = dict(info, native_loader=bool(native))
        return normalizeTable(table)                             # CalculateGradientScoresCmd.normalizeScores (:616-645)

    def calculateGradientScoresFromShardTable(self, masks, table: MatchTable, shard, queryThreshold, mirrorMask=True, roiMask=None,
                                              excludedRegions=cds.label_regions, batch=512, threads=8, zgap_radius=10,
                                              shard_targets=None) -> MatchTable:
        """calculateGradientScoresFromFilesTable with the library's variants in a variant shard (cds.VariantShard: every target's MIP,
        GradientImage and ZGapImage in one file, written once from the same files -- cds.VariantShardWriter.append_files): the selected
        targets' blobs are read into pinned memory and decoded on the GPU instead of being inflated on host cores.  Target t of the
        table is target `shard_targets[t]` of the shard (default: t).  A target stored without a gradient leaves its pairs at (-1, -1); one
        stored without a zgap gets it from the device as in the file path.  Same scores as from the files (tests/test_processor.py)."""
        n = len(table)
        table.gradientAreaGap = np.full(n, -1, dtype=np.int64)
        table.highExpressionArea = np.full(n, -1, dtype=np.int64)
        table.normalizedScore = np.zeros(n, dtype=np.float32)
        if n == 0:
            return table
        ctx = self.library.ctx if self.library is not None else self.sparse_library.ctx
        alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, queryThreshold, mirrorMask, roiMask, excludedRegions, context=ctx)
        if (alg.width, alg.height) != (shard.width, shard.height):
            raise cds.CdsInvalidArgument(cds.capi.CDS_ERR_INVALID_ARGUMENT, "the shard's image size differs from the masks'")
        regions = cds._resolve_regions(excludedRegions, alg.width)
        where = None if shard_targets is None else np.asarray(shard_targets, dtype=np.int64)

        def load_shard(lib, chunk):
            in_shard = chunk if where is None else where[chunk]
            has_g = (shard.flags[in_shard] & shard.HAS_GRADIENT) != 0
            slot = np.full(len(chunk), -1, dtype=np.int64)
            slot[has_g] = np.arange(int(has_g.sum()))
            if has_g.any():
                flags = lib.load_variants(0, shard, in_shard[has_g], threads=threads)
                for r0, r1 in _runs((flags & shard.HAS_ZGAP) == 0):
                    lib.make_zgap_variants(r0, r1 - r0, queryThreshold, zgap_radius, regions)
            return slot

        self.last_from_files = dict(self._score_in_batches(alg, table, ctx, batch, load_shard, overlap=True), variant_shard=True)
        return normalizeTable(table)

    def _score_in_batches(self, alg, table, ctx, batch, load, overlap):
        """The table's pairs scored `batch` distinct targets at a time: `load(lib, targets)` fills slots 0.. of a dense library of `batch`
        slots with the targets it can load and returns each target's slot (-1: none); with `overlap` batch i + 1 is loaded into a second
        library on a loader thread while batch i is scored."""
        from concurrent.futures import ThreadPoolExecutor
        # the table's rows grouped by target, once (a chunk's rows are then a slice of `order`)
        order = np.argsort(table.targetIndex, kind="stable")
        sorted_t = table.targetIndex[order]
        distinct, starts = np.unique(sorted_t, return_index=True)
        ends = np.append(starts[1:], len(sorted_t))
        cap = min(int(batch), len(distinct))
        n_libs = 2 if overlap and len(distinct) > cap else 1
        libs = [cds.Library(alg.width, alg.height, cap, context=ctx) for _ in range(n_libs)]

        def score(lib, c0, c1, slot):
            rows_sorted = np.arange(starts[c0], ends[c1 - 1])                      # positions in `order` of the chunk's rows
            slots = np.repeat(slot, ends[c0:c1] - starts[c0:c1])
            ok = slots >= 0
            rows = order[rows_sorted[ok]]
            if len(rows) == 0:
                return
            gap, high, _ = alg.scoreLibraryPairs(lib, table.maskIndex[rows].astype(np.int32), slots[ok])
            table.gradientAreaGap[rows] = gap
            table.highExpressionArea[rows] = high

        chunks = [(c0, min(c0 + cap, len(distinct))) for c0 in range(0, len(distinct), cap)]
        if n_libs == 2:
            with ThreadPoolExecutor(max_workers=1) as loader:
                pending = loader.submit(load, libs[0], distinct[chunks[0][0]:chunks[0][1]])
                for i, (c0, c1) in enumerate(chunks):
                    slot = pending.result()
                    if i + 1 < len(chunks):                                           # the next batch loads while this one is scored
                        n0, n1 = chunks[i + 1]
                        pending = loader.submit(load, libs[(i + 1) % 2], distinct[n0:n1])
                    score(libs[i % 2], c0, c1, slot)
        else:
            for c0, c1 in chunks:
                score(libs[0], c0, c1, load(libs[0], distinct[c0:c1]))
        return {"dense_libraries": n_libs, "batch": cap, "distinct_targets": int(len(distinct))}

    # ---- gradientScores ---------------------------------------------------------------------------
    def selectBestMatches(self, matches: List[CDMatch], topLineMatches, topSamplesPerLine, topMatchesPerSample) -> List[CDMatch]:
        """Per mask, as CalculateGradientScoresCmd.getCDMatchesForMaskMipIDs does (:323-379)."""
        by_mask = {}
        for mt in matches:
            by_mask.setdefault(mt.maskIndex, []).append(mt)
        out = []
        for m in sorted(by_mask):
            lines = selectTopRankedElements(by_mask[m], lambda v: self.targets[v.targetIndex].publishedName,
                                            lambda v: v.matchingPixels, topLineMatches, -1)
            for _, _, line_items in lines:
                samples = selectTopRankedElements(line_items, lambda v: self.targets[v.targetIndex].neuronId,
                                                  lambda v: v.matchingPixels, topSamplesPerLine, topMatchesPerSample)
                for _, _, sample_items in samples:
                    out.extend(sample_items)
        return out

    def calculateGradientScores(self, masks, matches: List[CDMatch], queryThreshold, mirrorMask=True, roiMask=None,
                                excludedRegions=cds.label_regions) -> List[CDMatch]:
        """Shape scores for `matches` (one launch), then the per-mask normalisation."""
        if not matches:
            return matches
        alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, queryThreshold, mirrorMask, roiMask, excludedRegions,
                                                        context=self.library.ctx)
        pm = np.array([mt.maskIndex for mt in matches], dtype=np.int32)
        ps = self._slots([mt.targetIndex for mt in matches])
        gap, high, mir = alg.scoreLibraryPairs(self.library, pm, ps)
        for mt, g, h in zip(matches, gap, high):
            # "only set the scores that were actually calculated" (CalculateGradientScoresCmd.java:564-570); the
            # match keeps the mirrored flag of the pixel-match scan
            if g != -1:
                mt.gradientAreaGap = int(g)
            if h != -1:
                mt.highExpressionArea = int(h)
        self.normalizeScores(matches)
        return matches

    @staticmethod
    def normalizeScores(matches: List[CDMatch]):
        """CalculateGradientScoresCmd.normalizeScores (:616-645)."""
        by_mask = {}
        for mt in matches:
            by_mask.setdefault(mt.maskIndex, []).append(mt)
        for items in by_mask.values():
            max_pix = max([-1] + [mt.matchingPixels for mt in items])
            max_grad = max([-1] + [mt.getGradScore() for mt in items])
            for mt in items:
                mt.normalizedScore = float(np.float32(cds.calculateNormalizedScore(mt.matchingPixels, mt.getGradScore(), max_pix, max_grad)))
