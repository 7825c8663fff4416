"""GPU parity tests of kernel 1 (pixel-match scan) through the C ABI: the reference's own known answers on
its own fixture images, then oracle parity on seeded synthetic libraries and on the edge cases the
reference's semantics define.  Bar: bit-exact integer counts and mirrored flags."""
import os

import numpy as np
import pytest

from tests.golden import expected as E

pytestmark = pytest.mark.gpu

W, H = 1210, 566


@pytest.fixture(scope="module")
def cds():
    from colormipsearch_amd import cds as m
    m.default_context()  # fails loudly without a gfx950 GPU
    return m


def test_loaded_native_library(cds):
    from colormipsearch_amd import capi
    assert b"gfx950" in capi.load().cds_version()
    rcp = cds.default_context().device_rcp_table()
    assert rcp[1] == 1.0 and abs(rcp[3] - 1 / 3) < 1e-6


# ---- the reference's JUnit tests, restated over the mirror API -------------------------------------
def test_pixelMatchScore(cds, fixtures):
    # PixelMatchColorDepthSearchAlgorithmTest.pixelMatchScore (:33-53)
    e = E.PIXEL_MATCH_DIRECT
    alg = cds.PixelMatchColorDepthSearchAlgorithm(fixtures[e["mask"]], 20, True, None, 0, False, 20, 0.01, 2, e["rects"])
    score = alg.calculateMatchingScore(fixtures[e["target"]], {})
    assert score.getScore() == 87
    assert not score.isMirrored()


@pytest.mark.parametrize("mask,target,expected,mirrored", E.PIXEL_MATCH_PROVIDER)
def test_multiplePixelScores(cds, fixtures, mask, target, expected, mirrored):
    # PixelMatchColorDepthSearchAlgorithmTest.multiplePixelScores (:56-153)
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(
        True, 20, 1, 2, cds.label_regions)
    alg = provider.createColorDepthQuerySearchAlgorithmWithDefaultParams(fixtures[mask], 20, 0)
    s = alg.calculateMatchingScore(fixtures[target], {})
    assert s.getScore() == expected
    assert s.isMirrored() == mirrored
    assert s.getNormalizedScore() == float(np.float32(expected / alg.getQuerySize()))


def test_fixture_matrix_all_transforms(cds, fixtures, oracle):
    """3 EM masks x 4 LM targets, every per-transform count, 4 parameter combinations, batched masks."""
    masks = np.stack([fixtures[k] for k in ("em_12191", "em_12191_FL", "em_LPLC2")])
    tkeys = ("lm_BJD", "lm_GMR", "lm_VT016795", "lm_VT033614")
    targets = np.stack([fixtures[k] for k in tkeys])
    lib = cds.Library(W, H, len(tkeys))
    lib.upload(0, targets)
    for xy, mirror in ((2, True), (2, False), (0, True), (0, False)):
        alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, mirror, None, 0, False, 20, 0.01, xy, E.STD_RECTS)
        scores, mir = alg.searchLibrary(lib)
        counts = alg.searchLibraryCounts(lib)
        for m in range(3):
            q = oracle.PixelMatchQuery(masks[m], 20, 20, 0.01, xy, mirror, E.STD_RECTS)
            assert alg.getQuerySize(m) == q.size
            for t in range(len(tkeys)):
                s, mm, c = q.score(targets[t])
                assert (int(scores[m, t]), bool(mir[m, t])) == (s, mm), (xy, mirror, m, t)
                assert np.array_equal(counts[m, t], c), (xy, mirror, m, t, counts[m, t], c)
        # single-mask kernel variant agrees with the batched one
        one = cds.PixelMatchColorDepthSearchAlgorithm(masks[1], 20, mirror, None, 0, False, 20, 0.01, xy, E.STD_RECTS)
        s1, m1 = one.searchLibrary(lib)
        assert np.array_equal(s1[0], scores[1]) and np.array_equal(m1[0], mir[1])
        assert np.array_equal(one.searchLibraryCounts(lib)[0], counts[1])


# ---- synthetic libraries --------------------------------------------------------------------------
def test_device_generator_equals_cpu_twin(cds, oracle):
    lib = cds.Library(W, H, 6)
    lib.generate_synthetic(0xC0FFEE, 5, 0, 3, kind=0)
    lib.generate_synthetic(0xC0FFEE, 2, 3, 2, kind=1)
    lib.generate_synthetic(0xBADC0DE, 7, 5, 1, kind=2)
    got = lib.download(0, 6)
    for i in range(3):
        assert np.array_equal(got[i], oracle.synth_image(0xC0FFEE, 5 + i, 0))
    for i in range(2):
        assert np.array_equal(got[3 + i], oracle.synth_image(0xC0FFEE, 2 + i, 1))
    assert np.array_equal(got[5], oracle.synth_image(0xBADC0DE, 7, 2))



def test_generator_beyond_one_launch(cds, oracle):
    """A launch is limited to < 2^32 work-items: 26,000 images need more than one (the generator once stopped silently
    after ~15,700 of 65,536 images)."""
    n = 26000
    lib = cds.Library(W, H, n)
    lib.generate_synthetic(0xC0FFEE, 5, 0, n, kind=0)
    for slot in (0, 24928, 24929, n - 1):
        got = lib.download(slot, 1)[0]
        assert np.array_equal(got, oracle.synth_image(0xC0FFEE, 5 + slot, 0)), slot


@pytest.mark.parametrize("kind,n", [(0, 48), (1, 6)])
def test_synthetic_library_parity(cds, oracle, kind, n):
    """BASELINE config 2 at a size the oracle finishes in seconds: 1 mask x n targets, xyShift 2, mirror."""
    rects = cds.label_regions(W)
    mask = oracle.synth_image(0xBADC0DE, 0, 2)
    lib = cds.Library(W, H, n)
    lib.generate_synthetic(0xC0FFEE, 0, 0, n, kind=kind)
    targets = lib.download(0, n)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
    scores, mir = alg.searchLibrary(lib)
    q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, 2, True, rects)
    want_s, want_m = q.scan(targets, n_threads=8)
    assert np.array_equal(scores[0], want_s)
    assert np.array_equal(mir[0], want_m.astype(bool))
    if kind == 1:
        assert scores.max() > 0  # dense targets do hit


def test_self_match_and_mirror(cds, oracle):
    rects = cds.label_regions(W)
    mask = oracle.synth_image(0xBADC0DE, 3, 2)
    targets = np.stack([mask, np.fliplr(mask), np.roll(mask, 2, axis=1), np.roll(mask, -2, axis=0), np.zeros_like(mask)])
    lib = cds.Library(W, H, len(targets))
    lib.upload(0, targets)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
    scores, mir = alg.searchLibrary(lib)
    P = alg.getQuerySize()
    q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, 2, True, rects)
    for t in range(len(targets)):
        s, m, _ = q.score(targets[t])
        assert (int(scores[0, t]), bool(mir[0, t])) == (s, m)
    assert scores[0, 0] <= P and scores[0, 0] > 0.9 * P and not mir[0, 0]
    assert mir[0, 1] and scores[0, 4] == 0


def test_batched_masks_and_parameter_sweep(cds, oracle):
    """BASELINE config 3 in miniature: a batch of masks shares one pass over the library."""
    rects = cds.label_regions(W)
    n_masks, n_t = 5, 12
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(n_masks)])
    lib = cds.Library(W, H, n_t)
    lib.generate_synthetic(0xC0FFEE, 100, 0, n_t - 2, kind=0)
    lib.upload(n_t - 2, np.stack([masks[1], np.fliplr(masks[2])]))
    targets = lib.download(0, n_t)
    for (mthr, dthr, ztol, xy, mirror) in ((20, 20, 0.01, 2, True), (100, 100, 0.02, 0, False), (20, 50, 0.3, 2, False)):
        alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, mthr, mirror, None, 0, False, dthr, ztol, xy, rects)
        scores, mir = alg.searchLibrary(lib)
        for m in range(n_masks):
            q = oracle.PixelMatchQuery(masks[m], mthr, dthr, ztol, xy, mirror, rects)
            ws, wm = q.scan(targets, n_threads=8)
            assert np.array_equal(scores[m], ws), (m, mthr, dthr, ztol, xy, mirror)
            assert np.array_equal(mir[m], wm.astype(bool))


def test_wide_windows_and_many_masks(cds, oracle):
    """Kernel variants the benchmark mask does not reach: a full-width window (5 load segments per row), a very wide
    image (8 segments), and mask batches larger than one launch."""
    rects = cds.label_regions(W)
    # (a) a sparse-lut TARGET image used as the mask: strokes all over the frame -> window = whole rows
    mask = oracle.synth_image(0xC0FFEE, 9001, 0)
    targets = np.stack([oracle.synth_image(0xC0FFEE, i, 0) for i in (9001, 9002, 9003)] + [np.fliplr(mask), oracle.synth_image(0xC0FFEE, 3, 1)])
    lib = cds.Library(W, H, len(targets))
    lib.upload(0, targets)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
    counts = alg.searchLibraryCounts(lib)
    q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, 2, True, rects)
    for t in range(len(targets)):
        assert np.array_equal(counts[0, t], q.score(targets[t])[2]), t
    # (b) 1600 x 64 image, mask pixels near both ends -> 8 segments
    w2, h2 = 1600, 64
    rng = np.random.default_rng(11)
    lut = oracle.depth_lut()
    m2 = np.zeros((h2, w2, 3), dtype=np.uint8)
    for x0 in (3, 700, 1500):
        m2[10:50, x0:x0 + 40] = lut[rng.integers(0, 256, size=(40, 40))]
    t2 = np.stack([m2, np.fliplr(m2), lut[rng.integers(0, 256, size=(h2, w2))].astype(np.uint8)])
    lib2 = cds.Library(w2, h2, len(t2))
    lib2.upload(0, t2)
    a2 = cds.PixelMatchColorDepthSearchAlgorithm(m2, 20, True, None, 0, False, 20, 0.01, 2, [])
    c2 = a2.searchLibraryCounts(lib2)
    q2 = oracle.PixelMatchQuery(m2, 20, 20, 0.01, 2, True, [])
    for t in range(len(t2)):
        assert np.array_equal(c2[0, t], q2.score(t2[t])[2]), t
    # (c) 34 masks in one set, 32 per launch: two launches
    os.environ["CDS_SCAN_MASKS_PER_LAUNCH"] = "32"
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(34)])
    tg = np.stack([masks[33], np.fliplr(masks[5]), oracle.synth_image(0xC0FFEE, 77, 1)])
    lib3 = cds.Library(W, H, len(tg))
    lib3.upload(0, tg)
    a3 = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects)
    s3, m3 = a3.searchLibrary(lib3)
    for m in (0, 5, 31, 32, 33):
        ws, wm = oracle.PixelMatchQuery(masks[m], 20, 20, 0.01, 2, True, rects).scan(tg, n_threads=4)
        assert np.array_equal(s3[m], ws) and np.array_equal(m3[m], wm.astype(bool)), m
    assert m3[5, 1] and s3[33, 0] > 0.9 * a3.getQuerySize(33)
    # (d) 33 masks: the second launch holds ONE mask in the batch layout
    a4 = cds.PixelMatchColorDepthSearchAlgorithm(masks[:33], 20, True, None, 0, False, 20, 0.01, 2, rects)
    del os.environ["CDS_SCAN_MASKS_PER_LAUNCH"]
    s4, m4 = a4.searchLibrary(lib3)
    assert np.array_equal(s4, s3[:33]) and np.array_equal(m4, m3[:33])
    # (e) the per-mask fallback of a batch (used when the window is too wide for the batch kernel's LDS)
    os.environ["CDS_SCAN_MASKS_PER_LAUNCH"] = "1"
    try:
        a5 = cds.PixelMatchColorDepthSearchAlgorithm(masks[:3], 20, True, None, 0, False, 20, 0.01, 2, rects)
    finally:
        del os.environ["CDS_SCAN_MASKS_PER_LAUNCH"]
    s5, m5 = a5.searchLibrary(lib3)
    assert np.array_equal(s5, s3[:3]) and np.array_equal(m5, m3[:3])
    a5b = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects)   # default: one launch of 34
    s5b, m5b = a5b.searchLibrary(lib3)
    assert np.array_equal(s5b, s3) and np.array_equal(m5b, m3)
    # (f) batches over wide windows: whole-frame masks (5 segments) and the 1600-pixel image (8 segments)
    wm = np.stack([oracle.synth_image(0xC0FFEE, i, 0) for i in (9001, 9002, 9003)])
    a6 = cds.PixelMatchColorDepthSearchAlgorithm(wm, 20, True, None, 0, False, 20, 0.01, 2, rects)
    c6 = a6.searchLibraryCounts(lib)
    for m in range(3):
        q6 = oracle.PixelMatchQuery(wm[m], 20, 20, 0.01, 2, True, rects)
        for t in range(len(targets)):
            assert np.array_equal(c6[m, t], q6.score(targets[t])[2]), (m, t)
    mm2 = np.stack([m2, np.roll(m2, 7, axis=1)])
    a7 = cds.PixelMatchColorDepthSearchAlgorithm(mm2, 20, True, None, 0, False, 20, 0.01, 2, [])
    c7 = a7.searchLibraryCounts(lib2)
    for m in range(2):
        q7 = oracle.PixelMatchQuery(mm2[m], 20, 20, 0.01, 2, True, [])
        for t in range(len(t2)):
            assert np.array_equal(c7[m, t], q7.score(t2[t])[2]), (m, t)


def test_sub_ranges_and_slots(cds, oracle):
    rects = cds.label_regions(W)
    mask = oracle.synth_image(0xBADC0DE, 0, 2)
    lib = cds.Library(W, H, 10)
    lib.generate_synthetic(0xC0FFEE, 40, 0, 10, kind=0)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
    full, fm = alg.searchLibrary(lib)
    part, pm = alg.searchLibrary(lib, 3, 4)
    assert np.array_equal(part, full[:, 3:7]) and np.array_equal(pm, fm[:, 3:7])
    empty, _ = alg.searchLibrary(lib, 2, 0)
    assert empty.shape == (1, 0)


# ---- edge cases of the reference's semantics ------------------------------------------------------
def test_mask_touching_every_border(cds, oracle):
    """Shifted positions falling off the image are skipped (-1, PixelMatch...Algorithm.java:132-144,241-243),
    mirrored columns use the full width; also a small non-standard geometry."""
    w, h = 64, 40
    rng = np.random.default_rng(7)
    lut = oracle.depth_lut()
    mask = np.zeros((h, w, 3), dtype=np.uint8)
    for (y, x) in [(0, 0), (0, w - 1), (h - 1, 0), (h - 1, w - 1), (1, 1), (h // 2, w // 2)]:
        mask[y, x] = lut[rng.integers(0, 256)]
    mask[0, 5:40] = lut[40]
    mask[h - 1, 10:60] = lut[200]
    mask[5:35, 0] = lut[100]
    mask[3:30, w - 1] = lut[150]
    targets = np.stack([mask, np.fliplr(mask), np.roll(np.roll(mask, 2, 0), -2, 1),
                        lut[rng.integers(0, 256, size=(h, w))].astype(np.uint8)])
    lib = cds.Library(w, h, len(targets))
    lib.upload(0, targets)
    for xy, mirror in ((2, True), (0, True), (2, False)):
        alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, mirror, None, 0, False, 20, 0.01, xy, [])
        counts = alg.searchLibraryCounts(lib)
        q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, xy, mirror, [])
        for t in range(len(targets)):
            assert np.array_equal(counts[0, t], q.score(targets[t])[2]), (xy, mirror, t)


@pytest.mark.parametrize("single_as_batch", [False, True])
def test_random_geometry_and_parameter_sweep(cds, oracle, single_as_batch, monkeypatch):
    """Seeded random sweep over image sizes, mask counts / shapes, thresholds, tolerances, shift and mirror settings:
    all 18 per-transform counts of every (mask, target) pair against the oracle.  Exercises the single-mask kernel, the
    batch kernel (different numbers of masks per launch) and every window / segment configuration small images reach."""
    if single_as_batch:   # single masks through the batch kernel (it is otherwise only reached with >= 2 masks)
        monkeypatch.setenv("CDS_SCAN_SINGLE_AS_BATCH", "1")
    rng = np.random.default_rng(int(os.environ.get("CDS_TEST_SWEEP_SEED", "20261003")))
    lut = oracle.depth_lut()

    def blobs(w, h, n, dense):
        img = np.zeros((h, w, 3), dtype=np.uint8)
        for _ in range(n):
            bw, bh = int(rng.integers(1, max(2, w // 3))), int(rng.integers(1, max(2, h // 3)))
            x0, y0 = int(rng.integers(0, w - bw + 1)), int(rng.integers(0, h - bh + 1))
            if dense:
                img[y0:y0 + bh, x0:x0 + bw] = rng.integers(0, 256, size=(bh, bw, 3))
            else:
                z = rng.integers(0, 256, size=(bh, bw))
                br = rng.uniform(0.3, 1.0, size=(bh, bw, 1))
                img[y0:y0 + bh, x0:x0 + bw] = (lut[z] * br).astype(np.uint8)
        return img

    for case in range(int(os.environ.get("CDS_TEST_SWEEP_CASES", "40"))):   # (a one-off soak run uses more)
        w, h = int(rng.integers(24, 420)), int(rng.integers(8, 150))
        n_masks = int(rng.choice([1, 1, 2, 3, 5, 9]))
        masks = np.stack([blobs(w, h, int(rng.integers(1, 5)), False) for _ in range(n_masks)])
        targets = [blobs(w, h, int(rng.integers(1, 8)), bool(rng.integers(0, 2))) for _ in range(4)]
        targets += [masks[0], np.fliplr(masks[-1]), np.roll(masks[0], 2, axis=1), np.zeros_like(masks[0])]
        targets = np.stack(targets)
        mthr, dthr = int(rng.choice([0, 20, 60, 150])), int(rng.choice([0, 20, 90]))
        ztol = float(rng.choice([0.0, 0.01, 0.02, 0.1, 0.5]))
        xy, mirror = int(rng.choice([0, 2])), bool(rng.integers(0, 2))
        rects = [(0, 0, w // 4, h // 5)] if rng.integers(0, 2) else []
        if n_masks > 2 and rng.integers(0, 2):
            os.environ["CDS_SCAN_MASKS_PER_LAUNCH"] = str(int(rng.integers(2, n_masks)))
        try:
            alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, mthr, mirror, None, 0, False, dthr, ztol, xy, rects)
        finally:
            os.environ.pop("CDS_SCAN_MASKS_PER_LAUNCH", None)
        lib = cds.Library(w, h, len(targets))
        lib.upload(0, targets)
        counts = alg.searchLibraryCounts(lib)
        scores, mir = alg.searchLibrary(lib)
        for m in range(n_masks):
            q = oracle.PixelMatchQuery(masks[m], mthr, dthr, ztol, xy, mirror, rects)
            for t in range(len(targets)):
                s, mi, c = q.score(targets[t])
                ctx_info = (case, w, h, n_masks, m, t, mthr, dthr, ztol, xy, mirror)
                assert np.array_equal(counts[m, t], c), ctx_info
                assert scores[m, t] == s and bool(mir[m, t]) == mi, ctx_info


def test_wrapped_uninitialised_memory(cds, oracle):
    """cds_library_wrap over memory full of garbage: uploads must leave black padding columns behind (shifted and mirrored
    probes just beyond the right image border read them)."""
    import torch
    w, h, n = 70, 33, 3   # pitch 240 B = 80 pixels: 10 padding columns
    rng = np.random.default_rng(3)
    lut = oracle.depth_lut()
    mask = np.zeros((h, w, 3), dtype=np.uint8)
    mask[4:28, w - 12:w] = lut[rng.integers(0, 256, size=(24, 12))]
    mask[10:20, 0:6] = lut[rng.integers(0, 256, size=(10, 6))]
    targets = np.stack([mask, np.fliplr(mask), lut[rng.integers(0, 256, size=(h, w))].astype(np.uint8)])
    buf = torch.full((cds.Library.bytes_required(w, h, n),), 0xA7, dtype=torch.uint8, device="cuda")
    lib = cds.Library(w, h, n, device_buffer=buf)
    lib.upload(0, targets)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.5, 2, [])
    counts = alg.searchLibraryCounts(lib)
    q = oracle.PixelMatchQuery(mask, 20, 20, 0.5, 2, True, [])
    for t in range(n):
        assert np.array_equal(counts[0, t], q.score(targets[t])[2]), t


def test_dense_full_frame_masks(cds, oracle):
    """Every pixel of the frame is a mask pixel (684 K of them: ~230 chunks per band, far more than the prefetched ones),
    one mask and a batch of two, sparse and dense targets."""
    rects = cds.label_regions(W)
    rng = np.random.default_rng(99)
    lut = oracle.depth_lut()
    masks = np.stack([lut[rng.integers(0, 256, size=(H, W))].astype(np.uint8) for _ in range(2)])
    targets = np.stack([oracle.synth_image(0xC0FFEE, 11, 0), oracle.synth_image(0xC0FFEE, 12, 1), masks[1]])
    lib = cds.Library(W, H, len(targets))
    lib.upload(0, targets)
    want = []
    for m in range(2):
        q = oracle.PixelMatchQuery(masks[m], 20, 20, 0.01, 2, True, rects)
        want.append(np.stack([q.score(t)[2] for t in targets]))
    one = cds.PixelMatchColorDepthSearchAlgorithm(masks[0], 20, True, None, 0, False, 20, 0.01, 2, rects)
    assert np.array_equal(one.searchLibraryCounts(lib)[0], want[0])
    both = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects)
    c = both.searchLibraryCounts(lib)
    assert np.array_equal(c[0], want[0]) and np.array_equal(c[1], want[1])
    assert c[1, 2, 4] > 0.9 * both.getQuerySize(1)   # the mask against itself, unshifted


@pytest.mark.parametrize("slice_bands,per_launch", [(1, None), (2, 3), (7, None), (3, 2)])
def test_band_slices_of_a_mask_batch(cds, oracle, slice_bands, per_launch, monkeypatch):
    """A mask batch scanned in band slices (several launches per group; the per-transform counts travel between them,
    through the caller's counts array or an internal scratch buffer) gives the results of the single launch."""
    rects = cds.label_regions(W)
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(5)])
    targets = np.stack([oracle.synth_image(0xC0FFEE, 300 + i, 0) for i in range(5)] + [masks[2], np.fliplr(masks[4])])
    lib = cds.Library(W, H, len(targets))
    lib.upload(0, targets)
    monkeypatch.setenv("CDS_SCAN_BAND_SLICE", str(slice_bands))
    if per_launch:   # several groups of masks (launch sets), each in band slices
        monkeypatch.setenv("CDS_SCAN_MASKS_PER_LAUNCH", str(per_launch))
    alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects)
    counts = alg.searchLibraryCounts(lib)          # slices accumulate in the caller's counts
    scores, mir = alg.searchLibrary(lib)           # slices accumulate in the scratch buffer
    for m in range(len(masks)):
        q = oracle.PixelMatchQuery(masks[m], 20, 20, 0.01, 2, True, rects)
        for t in range(len(targets)):
            s, mi, c = q.score(targets[t])
            assert np.array_equal(counts[m, t], c), (m, t)
            assert scores[m, t] == s and bool(mir[m, t]) == mi, (m, t)


def test_full_geometry_parameter_sweep(cds, oracle):
    """Seeded random thresholds / tolerances / shift / mirror / mask shapes at the production geometry (1210 x 566): the
    window, segment count and band structure vary with the mask, the code intervals with the tolerance."""
    rng = np.random.default_rng(int(os.environ.get("CDS_TEST_SWEEP_SEED", "77")))
    rects = cds.label_regions(W)
    base = [oracle.synth_image(0xBADC0DE, i, 2) for i in range(3)]
    tg = [oracle.synth_image(0xC0FFEE, 700 + i, 0) for i in range(4)] + [oracle.synth_image(0xC0FFEE, 3, 1)]
    for case in range(int(os.environ.get("CDS_TEST_SWEEP_CASES", "40")) // 7 + 1):
        n_masks = int(rng.choice([1, 2, 3]))
        masks = []
        for i in range(n_masks):
            m = base[int(rng.integers(0, 3))].copy()
            if rng.integers(0, 2):      # keep a random sub-rectangle only: narrower windows, fewer bands
                x0, x1 = sorted(rng.integers(300, 900, size=2)); y0, y1 = sorted(rng.integers(160, 340, size=2))
                keep = np.zeros((H, W), dtype=bool); keep[y0:y1 + 8, x0:x1 + 16] = True
                m[~keep] = 0
            if rng.integers(0, 3) == 0:
                m = np.roll(m, int(rng.integers(-300, 300)), axis=1)   # away from the centre (asymmetric windows)
            masks.append(m)
        masks = np.stack(masks)
        targets = np.stack(tg + [masks[0], np.fliplr(masks[-1])])
        mthr, dthr = int(rng.choice([20, 20, 50, 100])), int(rng.choice([20, 20, 60]))
        ztol = float(rng.choice([0.01, 0.01, 0.02, 0.05, 0.3]))
        xy, mirror = int(rng.choice([0, 2, 2])), bool(rng.integers(0, 4) > 0)
        lib = cds.Library(W, H, len(targets))
        lib.upload(0, targets)
        alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, mthr, mirror, None, 0, False, dthr, ztol, xy, rects)
        counts = alg.searchLibraryCounts(lib)
        for m in range(n_masks):
            q = oracle.PixelMatchQuery(masks[m], mthr, dthr, ztol, xy, mirror, rects)
            for t in range(len(targets)):
                assert np.array_equal(counts[m, t], q.score(targets[t])[2]), (case, m, t, mthr, dthr, ztol, xy, mirror)


def test_empty_and_classless_masks(cds, oracle):
    rects = cds.label_regions(W)
    black = np.zeros((H, W, 3), dtype=np.uint8)
    gray = black.copy()
    gray[200:220, 400:500] = 128          # r == g == b: no hue class, never matches, but counts in the query size
    in_label = black.copy()
    in_label[10:50, 10:200] = (0, 200, 255)   # entirely inside the name-label region
    target = oracle.synth_image(0xC0FFEE, 1, 0)
    for mask, size in ((black, 0), (gray, 2000), (in_label, 0)):
        alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
        assert alg.getQuerySize() == size == oracle.PixelMatchQuery(mask, 20, 20, 0.01, 2, True, rects).size
        s = alg.calculateMatchingScore(target, {})
        assert (s.getScore(), s.getNormalizedScore(), s.isMirrored()) == (0, 0.0, False)


def test_reference_error_behaviour(cds, fixtures):
    mask = fixtures["em_12191"]
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 1, 3, cds.label_regions)
    with pytest.raises(ValueError, match="XY shift parameter must be an even number"):   # Factory:58-60
        provider.createColorDepthQuerySearchAlgorithmWithDefaultParams(mask, 20, 0)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, cds.label_regions)
    with pytest.raises(ValueError, match="Invalid image size"):                           # PixelMatch...:171-175
        alg.calculateMatchingScore(np.zeros((100, 200, 3), dtype=np.uint8), {})
    with pytest.raises(cds.CdsError):                                                     # xyShift 4 overruns in the reference
        cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 4, cds.label_regions)
    # per-mask parameter overrides flow through ColorDepthSearchParams (Factory:39-43,55-68)
    p = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(False, 100, 2, 0, cds.label_regions)
    a = p.createColorDepthSearchAlgorithm(mask, 20, 0, cds.ColorDepthSearchParams().setParam("mirrorMask", True)
                                          .setParam("dataThreshold", 20).setParam("pixColorFluctuation", 1.0).setParam("xyShift", 2))
    assert a.calculateMatchingScore(fixtures["lm_VT016795"], {}).getScore() == 426


def test_is_match_set(cds, oracle):
    """The match set (ColorMIPSearch.isMatch, pctPositivePixels 1) equals the oracle's on a synthetic library."""
    rects = cds.label_regions(W)
    mask = oracle.synth_image(0xBADC0DE, 0, 2)
    n = 16
    lib = cds.Library(W, H, n)
    lib.generate_synthetic(0xC0FFEE, 300, 0, n - 1, kind=0)
    lib.upload(n - 1, mask[None])
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 1, 2, cds.label_regions)
    search = cds.ColorMIPSearch(1.0, 20, provider)
    alg = search.createQueryColorDepthSearchWithDefaultThreshold(mask)
    scores, mir = alg.searchLibrary(lib)
    got = search.isMatchArray(scores[0], alg.getQuerySize())
    want = np.array([oracle.is_match(int(s), alg.getQuerySize(), 1.0) for s in scores[0]])
    assert np.array_equal(got, want) and got[n - 1]
    one = search.isMatch(cds.PixelMatchScore(int(scores[0, 0]), scores[0, 0] / alg.getQuerySize(), False))
    assert one == bool(want[0])


def test_size_independent_properties_large(cds, oracle):
    """At a library size the oracle cannot check directly: sub-range consistency, permutation
    equivariance and a checksum over repeated scans (determinism)."""
    rects = cds.label_regions(W)
    mask = oracle.synth_image(0xBADC0DE, 0, 2)
    n = 2048
    lib = cds.Library(W, H, n)
    lib.generate_synthetic(0xC0FFEE, 0, 0, n, kind=0)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
    s1, m1 = alg.searchLibrary(lib)
    s2, m2 = alg.searchLibrary(lib)
    assert np.array_equal(s1, s2) and np.array_equal(m1, m2)
    assert (s1 >= 0).all() and (s1 <= alg.getQuerySize()).all()
    # slots generated as a different index order give the permuted scores
    lib2 = cds.Library(W, H, 64)
    for k, idx in enumerate(range(1000, 1064)):
        lib2.generate_synthetic(0xC0FFEE, idx, 63 - k, 1, kind=0)
    s3, _ = alg.searchLibrary(lib2)
    assert np.array_equal(s3[0][::-1], s1[0][1000:1064])
    # spot-check a few against the oracle
    q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, 2, True, rects)
    for t in (0, 777, 2047):
        s, m, _ = q.score(oracle.synth_image(0xC0FFEE, t, 0))
        assert (int(s1[0, t]), bool(m1[0, t])) == (s, m)


def test_per_pair_calls_from_many_threads(cds, oracle, fixtures):
    """The reference shares ONE algorithm object between all worker threads (LocalColorMIPSearchProcessor.java:88-105): 12
    threads call calculateMatchingScore of one pixel-match and one shape algorithm concurrently (each call takes a resident
    slot of the mask set's pool, with its own stream); every result equals the single-threaded one and the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    from tests.golden import expected as E
    mask = fixtures["em_12191"]
    tkeys = ["lm_BJD", "lm_VT016795", "lm_VT033614", "lm_GMR"]
    targets = [fixtures[k] for k in tkeys] + [oracle.synth_image(0xC0FFEE, 70 + i, 0) for i in range(4)]
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, E.STD_RECTS)
    q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, 2, True, E.STD_RECTS)
    want = [q.score(t)[:2] for t in targets]
    jobs = [i % len(targets) for i in range(96)]

    def one(i):
        s = alg.calculateMatchingScore(targets[i])
        return (s.getScore(), s.isMirrored())

    with ThreadPoolExecutor(max_workers=12) as pool:
        got = list(pool.map(one, jobs))
    assert got == [want[i] for i in jobs]
    assert want[0] == (414, False) and want[1] == (426, True)
    shp = cds.Shape2DMatchColorDepthSearchAlgorithm(mask, 20, True, None, E.STD_RECTS)
    sq = oracle.ShapeQuery(mask, 20, True, E.STD_RECTS)
    gkeys = {"lm_BJD": "grad_BJD", "lm_VT016795": "grad_VT016795", "lm_VT033614": "grad_VT033614"}
    triples = []
    for k in gkeys:
        z = oracle.make_zgap(fixtures[k], 20, 10, E.STD_RECTS)
        triples.append((fixtures[k], fixtures[gkeys[k]], z, sq.score(fixtures[k], fixtures[gkeys[k]], z)))

    def one_shape(i):
        t, g, z, _ = triples[i % 3]
        s = shp.calculateMatchingScore(t, {cds.ComputeFileType.GradientImage: lambda: g, cds.ComputeFileType.ZGapImage: lambda: z})
        return (s.getGradientAreaGap(), s.getHighExpressionArea(), s.getScore(), s.isMirrored())

    with ThreadPoolExecutor(max_workers=12) as pool:
        got_s = list(pool.map(one_shape, range(48)))
    assert got_s == [triples[i % 3][3] for i in range(48)]
