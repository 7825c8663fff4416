"""Parity of the workloads tools/bench_extra.py times (BASELINE configs[2] and configs[4] in miniature): a 128-mask batch
sharing one pass over a synthetic library, and GradientAreaGap over the top hits of several masks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 1210, 566


def test_mask_batch_over_synthetic_library(oracle):
    from colormipsearch_amd import cds
    rects = cds.label_regions(W)
    n_masks, n_targets = 128, 3072
    mlib = cds.Library(W, H, n_masks)
    mlib.generate_synthetic(0xBADC0DE, 0, 0, n_masks, kind=2)
    masks = mlib.download(0, n_masks)
    lib = cds.Library(W, H, n_targets)
    lib.generate_synthetic(0xC0FFEE, 0, 0, n_targets, kind=0)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects)
    scores, mir = alg.searchLibrary(lib)
    assert (scores > 0).any()
    for base in (0, 1500, n_targets - 48):
        t = lib.download(base, 48)
        for m in (0, 63, 64, 127):
            ws, wm = oracle.PixelMatchQuery(masks[m], 20, 20, 0.01, 2, True, rects).scan(t, n_threads=8)
            assert np.array_equal(ws, scores[m, base:base + 48]), (m, base)
            assert np.array_equal(wm.astype(bool), mir[m, base:base + 48]), (m, base)


def test_gradient_scores_of_top_hits(oracle):
    from colormipsearch_amd import cds
    rects = cds.label_regions(W)
    n_masks, n_targets, top = 3, 256, 24
    mlib = cds.Library(W, H, n_masks)
    mlib.generate_synthetic(0xBADC0DE, 0, 0, n_masks, kind=2)
    masks = mlib.download(0, n_masks)
    lib = cds.Library(W, H, n_targets)
    lib.generate_synthetic(0xC0FFEE, 0, 0, n_targets, kind=0, with_variants=True)
    alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, 20, True, None, rects)
    rng = np.random.default_rng(5)
    pm = np.repeat(np.arange(n_masks), top).astype(np.int32)
    ps = np.concatenate([rng.choice(n_targets, size=top, replace=False) for _ in range(n_masks)]).astype(np.int64)
    gap, high, mir = alg.scoreLibraryPairs(lib, pm, ps)
    for k in range(0, len(pm), 5):
        t = lib.download(int(ps[k]), 1)[0]
        g, z = lib.download_variants(int(ps[k]), 1)
        want = oracle.ShapeQuery(masks[pm[k]], 20, True, rects).score(t, g[0], z[0])
        assert (int(gap[k]), int(high[k]), bool(mir[k])) == (want[0], want[1], want[3]), k
