"""The caller side: match filter, top-N selection, gradient scores, normalisation.
CPU part: selectTopRankedElements restated from the reference's ItemsHandlingTest.  GPU part: the whole
colorDepthSearch -> gradientScores pipeline on a synthetic library equals the same pipeline over the oracle."""
import numpy as np
import pytest

from colormipsearch_amd import processor as P

W, H = 1210, 566


def _test_data():
    # same shape as ItemsHandlingTest.createTestData: several lines, several matches per line, distinct scores
    out = []
    k = 0
    for line in ("l1", "l2", "l3", "l4"):
        for j in range(3):
            out.append(P.CDMatch(0, k, matchingPixels=10 * (k % 5) + j + (7 if line == "l3" else 0), matchingPixelsRatio=0.1, mirrored=False))
            k += 1
    names = [f"l{1 + i // 3}" for i in range(12)]
    return out, names


def test_selectAllElementsWithAllSubResults():
    data, names = _test_data()
    ranked = P.selectTopRankedElements(data, lambda m: names[m.targetIndex], lambda m: m.matchingPixels, -1, -1)
    assert len(ranked) == 4
    assert all(len(items) == 3 for _, _, items in ranked)


def test_selectAllElementsWithLimitedSubResults():
    data, names = _test_data()
    for top_sub in (1, 2, 3):
        ranked = P.selectTopRankedElements(data, lambda m: names[m.targetIndex], lambda m: m.matchingPixels, -1, top_sub)
        assert len(ranked) == 4 and all(len(items) == top_sub for _, _, items in ranked)


def test_selectTopRankedElements_order_and_cut():
    data, names = _test_data()
    for top in (1, 2, 3):
        ranked = P.selectTopRankedElements(data, lambda m: names[m.targetIndex], lambda m: m.matchingPixels, top, -1)
        assert len(ranked) == top
        best = [s for _, s, _ in ranked]
        assert best == sorted(best, reverse=True)
        for _, s, items in ranked:
            scores = [m.matchingPixels for m in items]
            assert scores == sorted(scores, reverse=True) and s == scores[0]
    # blank group names fall into "UNKNOWN"
    ranked = P.selectTopRankedElements(data, lambda m: " ", lambda m: m.matchingPixels, -1, -1)
    assert [k for k, _, _ in ranked] == ["UNKNOWN"]


def test_normalizeScores_matches_reference_formula(oracle):
    ms = [P.CDMatch(0, 0, 636, 0.1, False, 156, 1897), P.CDMatch(0, 1, 679, 0.1, False, 1114361, 0),
          P.CDMatch(0, 2, 10, 0.1, False, None, None), P.CDMatch(1, 3, 5, 0.1, False, -1, -1)]
    P.GpuColorMIPSearchProcessor.normalizeScores(ms)
    assert abs(ms[0].normalizedScore - 46833.58) <= 0.1            # GradientAreaGapUtilsTest.java:31-32
    assert ms[2].normalizedScore == 10.0 and ms[3].normalizedScore == 5.0   # no shape score: the pixel score itself


def test_vectorised_ranking_equals_list_version():
    """_top_ranked (numpy) reproduces selectTopRankedElements: group order, in-group order, both cuts, ties."""
    rng = np.random.default_rng(1)
    for trial in range(400):
        n = int(rng.integers(1, 70))
        groups, scores = rng.integers(0, 6, size=n), rng.integers(0, 8, size=n)
        top, lim = int(rng.integers(-1, 5)), int(rng.integers(-1, 4))
        ref = P.selectTopRankedElements(list(range(n)), lambda i: f"g{groups[i]}", lambda i: scores[i], top, lim)
        want = [i for _, _, items in ref for i in items]
        got = P._top_ranked(np.arange(n), groups.astype(np.int64), scores, top, lim, [f"g{k}" for k in range(6)]).tolist()
        assert got == want, (trial, n, top, lim)


def test_columnar_selection_equals_list_selection():
    """selectBestMatchesTable (all masks, lines and samples in a few sorts) == selectBestMatches (the reference's nested
    loops) on random match lists with many ties, blank names and every combination of cuts."""
    rng = np.random.default_rng(3)
    for trial in range(300):
        nt, nm = int(rng.integers(1, 60)), int(rng.integers(1, 4))
        infos = [P.NeuronInfo(str(i), f"L{rng.integers(0, 7)}" if rng.integers(0, 9) else " ", f"S{rng.integers(0, 9)}") for i in range(nt)]
        proc = P.GpuColorMIPSearchProcessor.__new__(P.GpuColorMIPSearchProcessor)   # (selection needs the metadata only)
        proc.targets = infos
        rows = [(m, t, int(rng.integers(1, 6))) for m in range(nm) for t in range(nt) if rng.integers(0, 3)]
        if not rows:
            continue
        table = P.MatchTable(np.array([r[0] for r in rows], dtype=np.int32), np.array([r[1] for r in rows], dtype=np.int64),
                             np.array([r[2] for r in rows], dtype=np.int32), np.zeros(len(rows), dtype=np.float32), np.zeros(len(rows), dtype=bool))
        a, b, c = int(rng.integers(-1, 5)), int(rng.integers(-1, 4)), int(rng.integers(-1, 3))
        want = [(m.maskIndex, m.targetIndex) for m in proc.selectBestMatches(table.to_matches(), a, b, c)]
        got = proc.selectBestMatchesTable(table, a, b, c)
        assert list(zip(got.maskIndex.tolist(), got.targetIndex.tolist())) == want, (trial, a, b, c)


def test_tied_groups_follow_java_hashmap_order():
    """Groups of equal best score keep the iteration order of the HashMap that Collectors.groupingBy builds
    (ItemsHandling.java:86-103): bucket = (h ^ h >>> 16) & (capacity - 1) of String.hashCode, insertion order inside a bucket."""
    assert P.java_string_hash("Aa") == P.java_string_hash("BB") == 2112          # the classic colliding pair
    assert P.java_string_hash("line7") == 102977219 and P.java_string_hash("") == 0
    assert P.java_string_hash("R_\u00e9") == ((ord("R") * 31 + ord("_")) * 31 + 0xE9)   # UTF-16 code units
    # "a", "b", "c" hash to buckets 1, 2, 3 of the 16-bucket table whatever the insertion order; "q" (113) shares bucket 1
    # with "a" (97) and keeps its insertion position relative to it; "Aa" / "BB" collide completely
    assert P.java_hashmap_order(["c", "b", "a"]) == [2, 1, 0]
    assert P.java_hashmap_order(["q", "c", "a"]) == [0, 2, 1]
    assert P.java_hashmap_order(["BB", "Aa"]) == [0, 1] and P.java_hashmap_order(["Aa", "BB"]) == [0, 1]
    # 13 keys make the table grow to 32 buckets: "g16".."g28" spread differently than in 16
    keys = [f"g{i}" for i in range(16, 29)]
    def bucket32(k):
        h = P.java_string_hash(k)
        return (h ^ (h >> 16)) & 31
    assert P.java_hashmap_order(keys) == sorted(range(13), key=lambda i: (bucket32(keys[i]), i))
    # tied lines at a topResults cut: the survivors are the first in HashMap order, not the first seen
    names = ["c", "b", "a", "c", "b", "a"]
    items = list(range(6))
    ranked = P.selectTopRankedElements(items, lambda i: names[i], lambda i: 5, 2, -1)
    assert [k for k, _, _ in ranked] == ["a", "b"]
    codes, keys = P._group_codes(names)
    got = P._top_ranked(np.arange(6), codes, np.full(6, 5), 2, -1, keys).tolist()
    assert got == [2, 5, 1, 4]


def test_vectorised_normalisation_equals_the_scalar_function():
    """normalizedScoresArray == cds_normalized_score (GradientAreaGapUtils.java:219-235) element by element, including the
    branches that return the pixel score unchanged (zero scores, missing shape score, non-positive maxima)."""
    from colormipsearch_amd import cds
    rng = np.random.default_rng(3)
    pix = np.r_[rng.integers(0, 5000, size=400), [0, 0, 7, 7, 7, 7]]
    shape = np.r_[rng.integers(-1, 200000, size=400), [5, -1, -1, 0, 9, 9]]
    mp = np.r_[rng.integers(1, 5000, size=400), [10, 10, 10, 10, 0, 10]]
    ms = np.r_[rng.integers(1, 200000, size=400), [10, 10, 10, 10, 10, -1]]
    got = P.normalizedScoresArray(pix, shape, mp, ms)
    want = np.array([cds.calculateNormalizedScore(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(pix, shape, mp, ms)])
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_batched_gradient_scores_equal_the_resident_ones(oracle):
    """calculateGradientScoresBatchedTable (targets regenerated window by window into a few dense slots) == the same pairs
    scored over a library that holds every target with its variants."""
    from colormipsearch_amd import cds
    n_masks, n_t = 3, 50
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(n_masks)])
    lib = cds.Library(W, H, n_t)
    lib.generate_synthetic(0xC0FFEE, 7000, 0, n_t, kind=0, with_variants=True)
    slib = cds.SparseLibrary(W, H, n_t, 40000 * n_t)
    slib.append_from(lib)
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 1, 2, cds.label_regions)
    proc = P.GpuColorMIPSearchProcessor(cds.ColorMIPSearch(0.0, 20, provider), lib, None, sparse_library=slib)
    table = proc.findAllColorDepthMatchesTable(masks)
    assert len(table) > 20
    full = proc.calculateGradientScoresTable(masks, table.take(np.arange(len(table))), 20)
    small = cds.Library(W, H, 16)
    calls = []

    def load(dst, first, count):
        calls.append((first, count))
        dst.generate_synthetic(0xC0FFEE, 7000 + first, 0, count, kind=0, with_variants=True)

    batched = proc.calculateGradientScoresBatchedTable(masks, table.take(np.arange(len(table))), load, 20, batch=16, library=small)
    assert np.array_equal(batched.gradientAreaGap, full.gradientAreaGap) and np.array_equal(batched.highExpressionArea, full.highExpressionArea)
    assert np.array_equal(batched.normalizedScore, full.normalizedScore)
    assert (batched.gradientAreaGap >= 0).all() and proc.last_gradscore["pairs"] == len(table)
    assert all(f % 16 == 0 and f + c <= n_t for f, c in calls) and calls[-1] == (48, 2)
    # only the SELECTED targets resident (calculateGradientScoresSelectedTable): the distinct targets of the pairs, 16 at a time,
    # through the list generator (== the contiguous generator image by image)
    loaded = []

    def load_sel(dst, idx):
        loaded.append(idx.copy())
        dst.generate_synthetic_list(0xC0FFEE, 7000 + idx, 0, kind=0, with_variants=True)

    sel = proc.calculateGradientScoresSelectedTable(masks, table.take(np.arange(len(table))), load_sel, 20, batch=16, library=small)
    assert np.array_equal(sel.gradientAreaGap, full.gradientAreaGap) and np.array_equal(sel.highExpressionArea, full.highExpressionArea)
    assert np.array_equal(sel.normalizedScore, full.normalizedScore)
    distinct = np.unique(table.targetIndex)
    assert np.array_equal(np.concatenate(loaded), distinct) and proc.last_gradscore["distinct_targets"] == len(distinct)
    assert proc.last_gradscore["pairs"] == len(table) and all(len(i) <= 16 for i in loaded)
    # a processor over a sub-range of the library (first != 0): table indices are relative, the loaders get LIBRARY indices
    proc2 = P.GpuColorMIPSearchProcessor(cds.ColorMIPSearch(0.0, 20, provider), lib, None, first=10, count=30, sparse_library=slib)
    t2 = proc2.findAllColorDepthMatchesTable(masks)
    assert len(t2) > 5 and t2.targetIndex.min() >= 0 and t2.targetIndex.max() < 30
    want2 = proc2.calculateGradientScoresTable(masks, t2.take(np.arange(len(t2))), 20)
    calls.clear()
    got2 = proc2.calculateGradientScoresBatchedTable(masks, t2.take(np.arange(len(t2))), load, 20, batch=16, library=small)
    assert np.array_equal(got2.gradientAreaGap, want2.gradientAreaGap) and np.array_equal(got2.highExpressionArea, want2.highExpressionArea)
    assert all(10 <= f and f + c <= 40 for f, c in calls)          # (never beyond first + count: ADVICE round 3)
    got3 = proc2.calculateGradientScoresSelectedTable(masks, t2.take(np.arange(len(t2))), load_sel, 20, batch=16, library=small)
    assert np.array_equal(got3.gradientAreaGap, want2.gradientAreaGap)
    # the list generator == the contiguous one
    a, b = cds.Library(W, H, 3), cds.Library(W, H, 3)
    a.generate_synthetic(0xC0FFEE, 7003, 0, 3, kind=0, with_variants=True)
    b.generate_synthetic_list(0xC0FFEE, [7005, 7003, 7004], 0, kind=0, with_variants=True)
    assert np.array_equal(a.download(0, 3), b.download(0, 3)[[1, 2, 0]])
    ga, za = a.download_variants(0, 3)
    gb, zb = b.download_variants(0, 3)
    assert np.array_equal(ga, gb[[1, 2, 0]]) and np.array_equal(za, zb[[1, 2, 0]])


@pytest.mark.gpu
def test_mask_sets_prepared_while_the_previous_one_is_scanned(oracle):
    """findAllColorDepthMatchesTable over a sparse library with 21 masks in sets of 8 (preparation of set i + 1 on a second
    thread during the scan of set i) == one set of 21."""
    from colormipsearch_amd import cds
    n_masks, n_t = 21, 300
    mlib = cds.Library(W, H, n_masks)
    mlib.generate_synthetic(0xBADC0DE, 0, 0, n_masks, kind=2)
    masks = mlib.download(0, n_masks)
    lib = cds.Library(W, H, n_t)
    lib.generate_synthetic(0xC0FFEE, 4000, 0, n_t, kind=0)
    lib.upload(5, masks[[3, 17]])
    slib = cds.SparseLibrary(W, H, n_t, 40000 * n_t)
    slib.append_from(lib)
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 1, 2, cds.label_regions)
    proc = P.GpuColorMIPSearchProcessor(cds.ColorMIPSearch(0.01, 20, provider), None, None, sparse_library=slib)
    one = proc.findAllColorDepthMatchesTable(masks, mask_set_size=64)
    piped = proc.findAllColorDepthMatchesTable(masks, mask_set_size=8)
    ramp = proc.findAllColorDepthMatchesTable(masks, mask_set_size=16, first_set_size=8)     # sets of 8 and 13 masks: a smaller first set
    for f in ("maskIndex", "targetIndex", "matchingPixels", "matchingPixelsRatio", "mirrored"):
        assert np.array_equal(getattr(ramp, f), getattr(one, f)), f
    for f in ("maskIndex", "targetIndex", "matchingPixels", "matchingPixelsRatio", "mirrored"):
        assert np.array_equal(getattr(one, f), getattr(piped, f)), f
    assert len(one) > 20 and set(one.maskIndex.tolist()) >= {3, 17}


@pytest.mark.gpu
def test_a_wrong_size_mask_or_target_costs_only_its_own_pairs(oracle):
    """The reference catches Throwable per (mask, target) pair, keeps the message and drops that pair
    (AbstractColorMIPSearchProcessor.java:68-83, LocalColorMIPSearchProcessor.java:102-107): a target or a mask of another image size
    ('Invalid image size', PixelMatchColorDepthSearchAlgorithm.java:171-175) loses its own pairs and nothing else.  One wrong-size
    mask among 11 and one wrong-size target (and one unreadable one) among 90, over a growable window of 32 resident targets: the rows
    equal those of the search without the bad images, with the caller's indices, and `errors` names the three."""
    from colormipsearch_amd import cds
    n_masks, n_t = 10, 88
    mlib = cds.Library(W, H, n_masks)
    mlib.generate_synthetic(0xBADC0DE, 0, 0, n_masks, kind=2)
    good_masks = mlib.download(0, n_masks)
    lib = cds.Library(W, H, n_t)
    lib.generate_synthetic(0xC0FFEE, 7000, 0, n_t, kind=0)
    lib.upload(3, good_masks[[1, 8]])
    good_targets = lib.download(0, n_t)
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 1, 2, cds.label_regions)
    search = cds.ColorMIPSearch(0.01, 20, provider)
    whole = cds.SparseLibrary(W, H, n_t, 40000 * n_t)
    whole.append(good_targets)
    with P.GpuColorMIPSearchProcessor(search, None, None, sparse_library=whole) as ref:
        want = ref.findAllColorDepthMatchesTable(good_masks, mask_set_size=8)
        assert ref.errors == []
    # the caller's lists: mask 4 and targets 20 and 61 are bad
    masks = list(good_masks[:4]) + [np.zeros((H // 2, W, 3), np.uint8)] + list(good_masks[4:])
    targets = list(good_targets[:20]) + [np.zeros((H, W - 10, 3), np.uint8)] + list(good_targets[20:60]) + [None] + list(good_targets[60:])
    mask_of = np.array([0, 1, 2, 3, 5, 6, 7, 8, 9, 10])
    target_of = np.array([i for i in range(len(targets)) if i not in (20, 61)])
    window = cds.SparseLibrary(W, H)                                    # growable, starts small
    assert window.capacity_images == 1024
    with P.GpuColorMIPSearchProcessor(search, None, None, sparse_library=window) as proc:
        got = proc.findAllColorDepthMatchesOverChunksTable(masks, len(targets), lambda a, n: targets[a:a + n], 32, mask_set_size=8)
        assert [c["resident"] for c in proc.last_chunks] == [31, 31, 26]
        kinds = sorted((k, i) for k, i, _ in proc.errors)
        assert kinds == [("mask", 4), ("target", 20), ("target", 61)], proc.errors
        assert all("Invalid image size" in m or m == "no image" for _, _, m in proc.errors)
    assert np.array_equal(got.maskIndex, mask_of[want.maskIndex]) and np.array_equal(got.targetIndex, target_of[want.targetIndex])
    for f in ("matchingPixels", "matchingPixelsRatio", "mirrored"):
        assert np.array_equal(getattr(got, f), getattr(want, f)), f
    assert len(want) > 10 and {1, 8} <= set(want.maskIndex.tolist())
    # the list form reports the caller's mask indices too
    with P.GpuColorMIPSearchProcessor(search, None, None, sparse_library=whole) as proc:
        rows = proc.findAllColorDepthMatches(masks)
        assert [(r.maskIndex, r.targetIndex, r.matchingPixels) for r in rows] == list(zip(mask_of[want.maskIndex].tolist(), want.targetIndex.tolist(), want.matchingPixels.tolist()))
        assert [e[:2] for e in proc.errors] == [("mask", 4)]


@pytest.mark.gpu
def test_a_failing_mask_set_leaves_no_worker_behind(oracle):
    """when one task of the pipelined search raises, the call must not return while other tasks still scan the library
    (the caller may close it next).  A mask set whose preparation fails -- xyShift 4, which the library refuses as the reference's
    table overrun -- in the middle of four sets: the error arrives, no task of the call is still running, the pools still work."""
    from colormipsearch_amd import cds
    n_masks, n_t = 32, 64
    mlib = cds.Library(W, H, n_masks)
    mlib.generate_synthetic(0xBADC0DE, 0, 0, n_masks, kind=2)
    masks = mlib.download(0, n_masks)
    slib = cds.SparseLibrary(W, H)
    lib = cds.Library(W, H, n_t)
    lib.generate_synthetic(0xC0FFEE, 100, 0, n_t, kind=0)
    slib.append_from(lib)
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 1, 2, cds.label_regions)
    search = cds.ColorMIPSearch(0.01, 20, provider)
    with P.GpuColorMIPSearchProcessor(search, None, None, sparse_library=slib) as proc:
        good = proc.findAllColorDepthMatchesTable(masks, mask_set_size=8)
        make, calls = search.createQueryColorDepthSearchWithDefaultThreshold, []

        def failing(m):
            calls.append(len(m))
            if len(calls) == 3:
                raise cds.CdsError(cds.capi.CDS_ERR_UNSUPPORTED, "injected failure of the third set")
            return make(m)

        search.createQueryColorDepthSearchWithDefaultThreshold = failing
        try:
            with pytest.raises(cds.CdsError, match="injected"):
                proc.findAllColorDepthMatchesTable(masks, mask_set_size=8)
        finally:
            search.createQueryColorDepthSearchWithDefaultThreshold = make
        import threading
        busy = [t.name for t in threading.enumerate() if t.name.startswith("cds-")]
        # (the pool threads exist, idle: a second search through the same pools gives the same rows)
        again = proc.findAllColorDepthMatchesTable(masks, mask_set_size=8)
        for f in ("maskIndex", "targetIndex", "matchingPixels", "mirrored"):
            assert np.array_equal(getattr(again, f), getattr(good, f)), f
        assert busy
    assert not [t.name for t in threading.enumerate() if t.name.startswith("cds-")]          # close() joined them


@pytest.mark.gpu
def test_sharded_processor_equals_single_library(oracle):
    """ShardedGpuColorMIPSearchProcessor (one process, cds.Comm over three contexts of the one GPU: uneven shards) == the processor
    over one sparse library holding all targets: same match rows, same order."""
    from colormipsearch_amd import cds
    n_masks, n_t = 11, 120
    mlib = cds.Library(W, H, n_masks)
    mlib.generate_synthetic(0xBADC0DE, 0, 0, n_masks, kind=2)
    masks = mlib.download(0, n_masks)
    lib = cds.Library(W, H, n_t)
    lib.generate_synthetic(0xC0FFEE, 9000, 0, n_t, kind=0)
    lib.upload(7, masks[[2, 9]])
    lib.upload(100, np.fliplr(masks[5])[None].copy())
    targets = lib.download(0, n_t)
    whole = cds.SparseLibrary(W, H, n_t, 40000 * n_t)
    whole.append(targets)
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 1, 2, cds.label_regions)
    search = cds.ColorMIPSearch(0.01, 20, provider)
    want = P.GpuColorMIPSearchProcessor(search, None, None, sparse_library=whole).findAllColorDepthMatchesTable(masks, mask_set_size=64)
    bounds = [0, 50, 57, n_t]
    ctxs = [cds.Context(0) for _ in range(3)]
    comm = cds.Comm(ctxs)
    shards = []
    for r in range(3):
        s = cds.SparseLibrary(W, H, max(1, bounds[r + 1] - bounds[r]), 40000 * (bounds[r + 1] - bounds[r]) + 64, context=ctxs[r])
        s.append(targets[bounds[r]:bounds[r + 1]])
        shards.append(s)
    got = P.ShardedGpuColorMIPSearchProcessor(search, comm, shards).findAllColorDepthMatchesTable(masks, mask_set_size=8)
    for f in ("maskIndex", "targetIndex", "matchingPixels", "matchingPixelsRatio", "mirrored"):
        assert np.array_equal(getattr(want, f), getattr(got, f)), f
    assert len(want) > 10 and {2, 5, 9} <= set(want.maskIndex.tolist())
    comm.close()


@pytest.mark.gpu
def test_pipeline_equals_oracle_pipeline(oracle):
    from colormipsearch_amd import cds
    rects = cds.label_regions(W)
    n_masks, n_t = 2, 40
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(n_masks)])
    lib = cds.Library(W, H, n_t)
    lib.generate_synthetic(0xC0FFEE, 500, 0, n_t - 4, kind=0, with_variants=True)
    # a few targets that really match: the masks themselves (plain / mirrored) with generated variants of another image
    extra = np.stack([masks[0], np.fliplr(masks[0]), masks[1], np.roll(masks[1], 2, axis=1)])
    lib.upload(n_t - 4, extra)
    g, z = lib.download_variants(0, 4)
    lib.upload_variants(n_t - 4, g, z)
    targets = lib.download(0, n_t)
    grads, zgaps = lib.download_variants(0, n_t)
    infos = [P.NeuronInfo(str(i), f"line{i % 7}", f"sample{i % 11}") for i in range(n_t)]

    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 1, 2, cds.label_regions)
    search = cds.ColorMIPSearch(0.01, 20, provider)     # low pctPositivePixels so that sparse synthetic targets match too
    proc = P.GpuColorMIPSearchProcessor(search, lib, infos)
    matches = proc.findAllColorDepthMatches(masks)

    want = []
    for m in range(n_masks):
        q = oracle.PixelMatchQuery(masks[m], 20, 20, 0.01, 2, True, rects)
        s, mir = q.scan(targets, n_threads=8)
        for t in range(n_t):
            if oracle.is_match(int(s[t]), q.size, 0.01):
                want.append((m, t, int(s[t]), bool(mir[t])))
    assert [(mt.maskIndex, mt.targetIndex, mt.matchingPixels, mt.mirrored) for mt in matches] == want
    assert len(want) >= 4

    best = proc.selectBestMatches(matches, 3, 2, 1)
    assert 0 < len(best) <= n_masks * 3 * 2
    for m in range(n_masks):
        lines = {infos[mt.targetIndex].publishedName for mt in best if mt.maskIndex == m}
        assert len(lines) <= 3
    scored = proc.calculateGradientScores(masks, best, 20, True, None, cds.label_regions)
    for mt in scored:
        sq = oracle.ShapeQuery(masks[mt.maskIndex], 20, True, rects)
        gap, high, _, _ = sq.score(targets[mt.targetIndex], grads[mt.targetIndex], zgaps[mt.targetIndex])
        assert (mt.gradientAreaGap, mt.highExpressionArea) == (gap, high)
    for m in range(n_masks):
        items = [mt for mt in scored if mt.maskIndex == m]
        mp, mg = max(mt.matchingPixels for mt in items), max(mt.getGradScore() for mt in items)
        for mt in items:
            assert mt.normalizedScore == float(np.float32(oracle.normalized_score(mt.matchingPixels, mt.getGradScore(), mp, mg)))
    # the columnar pipeline gives the same rows in the same order
    key = lambda mt: (mt.maskIndex, mt.targetIndex, mt.matchingPixels, mt.matchingPixelsRatio, mt.mirrored,
                      mt.gradientAreaGap, mt.highExpressionArea, mt.normalizedScore)
    table = proc.findAllColorDepthMatchesTable(masks)
    assert [key(a)[:5] for a in table.to_matches()] == [key(b)[:5] for b in matches]
    best_t = proc.selectBestMatchesTable(table, 3, 2, 1)
    assert [key(a)[:5] for a in best_t.to_matches()] == [key(b)[:5] for b in best]
    scored_t = proc.calculateGradientScoresTable(masks, best_t, 20, True, None, cds.label_regions)
    assert [key(a) for a in scored_t.to_matches()] == [key(b) for b in scored]
    # production split: the scan over a sparse resident library, the shape scorer over dense slots that hold only the
    # selected targets (in another order) with their variants
    slib = cds.SparseLibrary(W, H, n_t, 40000 * n_t)
    slib.append_from(lib)
    chosen = sorted({mt.targetIndex for mt in best})[::-1]
    small = cds.Library(W, H, len(chosen))
    small.upload(0, targets[chosen])
    small.upload_variants(0, grads[chosen], zgaps[chosen])
    slot_of = np.full(n_t, -1, dtype=np.int64)
    slot_of[chosen] = np.arange(len(chosen))
    proc2 = P.GpuColorMIPSearchProcessor(search, small, infos, sparse_library=slib, dense_slots=slot_of)
    matches2 = proc2.findAllColorDepthMatches(masks)
    assert [key(a)[:5] for a in matches2] == [key(b)[:5] for b in matches]
    scored2 = proc2.calculateGradientScores(masks, proc2.selectBestMatches(matches2, 3, 2, 1), 20, True, None, cds.label_regions)
    assert [key(a) for a in scored2] == [key(b) for b in scored]


@pytest.mark.gpu
def test_gradient_scores_from_files(tmp_path, fixtures):
    """The file-backed gradient-score step on the reference's own images: targets resident only as a sparse library, the
    selected targets and their GradientImage / ZGapImage variants read from TIFF / PNG files in two batches; the reference's
    JUnit tuples come out (Shape2DMatchColorDepthSearchAlgorithmTest.java:86-132, :230-291); a target without a gradient
    file stays at (-1, -1)."""
    PIL = pytest.importorskip("PIL.Image")
    from colormipsearch_amd import cds
    from tests.golden import expected as E
    tkeys = ["lm_VT033614", "lm_BJD", "lm_VT016795", "lm_GMR"]
    gkeys = {"lm_VT033614": "grad_VT033614", "lm_BJD": "grad_BJD", "lm_VT016795": "grad_VT016795"}
    tfiles, gfiles, zfiles = [], [], []
    for i, k in enumerate(tkeys):
        p = tmp_path / f"{k}.tif"
        PIL.fromarray(fixtures[k], "RGB").save(p, compression="packbits")
        tfiles.append(str(p))
        if k in gkeys:
            g = tmp_path / f"{k}_grad.png"
            PIL.fromarray(fixtures[gkeys[k]]).save(g)
            gfiles.append(str(g))
        else:
            gfiles.append(None)
        zfiles.append(None)                                   # zgap synthesised from the target (as the reference's tests do)
    masks = np.stack([fixtures["em_12191"], fixtures["em_12191_FL"]])
    sl = cds.SparseLibrary(W, H, len(tkeys), len(tkeys) * 80000)
    sl.append_files(tfiles, threads=4)
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 1, 2, cds.label_regions)
    proc = P.GpuColorMIPSearchProcessor(cds.ColorMIPSearch(0.01, 20, provider), None, [P.NeuronInfo(k, k, k) for k in tkeys], sparse_library=sl)
    table = proc.findAllColorDepthMatchesTable(masks)
    pix = {(int(m), tkeys[int(t)]): int(s) for m, t, s in zip(table.maskIndex, table.targetIndex, table.matchingPixels)}
    assert pix[(0, "lm_VT033614")] == 439 and pix[(0, "lm_BJD")] == 414 and pix[(1, "lm_VT016795")] == 483
    scored = proc.calculateGradientScoresFromFilesTable(masks, table, tfiles, gfiles, zfiles, 20, batch=2, threads=3)
    got = {(int(m), tkeys[int(t)]): (int(g), int(h)) for m, t, g, h in zip(scored.maskIndex, scored.targetIndex, scored.gradientAreaGap, scored.highExpressionArea)}
    mk = {"em_12191": 0, "em_12191_FL": 1}
    for mask, target, grad, zgap, gap, high, score, mirrored in E.SHAPE_SCORES:
        if zgap is None and gkeys.get(target) == grad:
            assert got[(mk[mask], target)] == (gap, high), (mask, target)
    assert all(v == (-1, -1) for (m, t), v in got.items() if t == "lm_GMR")
    assert (scored.normalizedScore > 0).any()
    # the same library as a variant shard (written from the same files, no GPU involved; decoded on the device): the same scores
    shard_path = tmp_path / "variants.cdsv"
    with cds.VariantShardWriter(shard_path, W, H) as w:
        w.append_files(tfiles, gfiles, zfiles, threads=2)
    shard = cds.VariantShard(shard_path)
    assert (shard.flags & shard.HAS_GRADIENT).tolist() == [1, 1, 1, 0] and (shard.flags & shard.GRADIENT_DIFF2).tolist() == [8, 8, 8, 0]
    again = proc.calculateGradientScoresFromShardTable(masks, table.take(np.arange(len(table))), shard, 20, batch=2, threads=3)
    assert proc.last_from_files["variant_shard"] and proc.last_from_files["dense_libraries"] == 2
    assert np.array_equal(again.gradientAreaGap, scored.gradientAreaGap) and np.array_equal(again.highExpressionArea, scored.highExpressionArea)
    assert np.array_equal(again.normalizedScore, scored.normalizedScore)
    # ... and with the shard's targets in another order than the library's
    back = cds.VariantShardWriter(tmp_path / "reversed.cdsv", W, H)
    back.append_files(tfiles[::-1], gfiles[::-1], None)
    back.finish()
    third = proc.calculateGradientScoresFromShardTable(masks, table.take(np.arange(len(table))), cds.VariantShard(tmp_path / "reversed.cdsv"), 20,
                                                       batch=8, shard_targets=[3, 2, 1, 0])
    assert np.array_equal(third.gradientAreaGap, scored.gradientAreaGap) and np.array_equal(third.highExpressionArea, scored.highExpressionArea)
