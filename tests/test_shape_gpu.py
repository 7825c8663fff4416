"""GPU parity tests of kernel 2 (GradientAreaGap scorer) and of the mask preparation / max filter it needs."""
import os

import numpy as np
import pytest

from tests.golden import expected as E

pytestmark = pytest.mark.gpu

W, H = 1210, 566


@pytest.fixture(scope="module")
def cds():
    from colormipsearch_amd import cds as m
    m.default_context()
    return m


@pytest.fixture(autouse=True, params=["policy", "zgap-bytes", "zslice-plane"])
def s_of_z_source(request, monkeypatch):
    """Every test of this module runs three times: with the library's own choice of where the scorer gets S(z) of the zgap pixels
    (round 4: the zgap bytes when the targets serve few pairs, the converted zslice plane otherwise) and with either source forced
    (CDS_SHAPE_ZSLICE, read per call) -- the two kernels must give the reference's numbers alike."""
    if request.param == "policy":
        monkeypatch.delenv("CDS_SHAPE_ZSLICE", raising=False)
    else:
        monkeypatch.setenv("CDS_SHAPE_ZSLICE", "0" if request.param == "zgap-bytes" else "1")
    return request.param


def test_overExpressesMaskExpression(cds, fixtures):
    # Shape2DMatchColorDepthSearchAlgorithmTest.overExpressesMaskExpression (:32-60)
    alg = cds.Shape2DMatchColorDepthSearchAlgorithm(fixtures["em_12191_FL"], 20, True, None, cds.label_regions)
    sizeMask, sizeHigh = alg.maskStats()
    assert sizeMask[0] == 17340
    assert sizeHigh[0] == 70640


def test_gpu_max_filter_equals_oracle(cds, fixtures, oracle):
    for key in ("lm_VT033614", "lm_GMR"):
        got = cds.default_context().make_zgap(fixtures[key], 20, 10, cds.label_regions)
        want = oracle.make_zgap(fixtures[key], 20, 10, E.STD_RECTS)
        assert np.array_equal(got, want)


@pytest.fixture(scope="module")
def shape_algs(cds, fixtures):
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createShapeMatchCDSAlgorithmProvider(True, None, cds.label_regions)
    return {k: provider.createColorDepthQuerySearchAlgorithmWithDefaultParams(fixtures[k], 20, 0) for k in ("em_12191", "em_12191_FL")}


@pytest.mark.parametrize("mask,target,grad,zgap,gaps,high,score,mirrored", E.SHAPE_SCORES)
def test_computeShapeScore(cds, fixtures, shape_algs, mask, target, grad, zgap, gaps, high, score, mirrored):
    # Shape2DMatchColorDepthSearchAlgorithmTest.computeShapeScoreUsing{AlgorithmProvider,DirectAlgorithmConstructor}
    targetImageArray = fixtures[target]
    variantSuppliers = {
        cds.ComputeFileType.GradientImage: lambda: fixtures[grad],
        cds.ComputeFileType.ZGapImage: (lambda: fixtures[zgap]) if zgap else
        (lambda: cds.default_context().make_zgap(targetImageArray, 20, 10, cds.label_regions)),
    }
    s = shape_algs[mask].calculateMatchingScore(targetImageArray, variantSuppliers)
    assert s.getGradientAreaGap() == gaps
    assert s.getHighExpressionArea() == high
    assert s.getScore() == score
    assert s.isMirrored() == mirrored


def test_missing_variants_give_minus_one(cds, fixtures, shape_algs):
    # Shape2DMatchColorDepthSearchAlgorithm.java:155-158
    s = shape_algs["em_12191"].calculateMatchingScore(fixtures["lm_BJD"], {cds.ComputeFileType.GradientImage: lambda: fixtures["grad_BJD"]})
    assert (s.getGradientAreaGap(), s.getHighExpressionArea(), s.getScore(), s.isMirrored()) == (-1, -1, -1, False)
    assert shape_algs["em_12191"].getRequiredTargetVariantTypes() == {cds.ComputeFileType.GradientImage, cds.ComputeFileType.ZGapImage}


def test_synthetic_pairs_parity(cds, oracle):
    """BASELINE config 5 in miniature: several masks x several resident targets with generated variants;
    mirror on/off and an ROI mask."""
    rects = cds.label_regions(W)
    n_masks, n_t = 3, 6
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(n_masks)])
    lib = cds.Library(W, H, n_t + 1)
    lib.generate_synthetic(0xC0FFEE, 10, 0, n_t, kind=0, with_variants=True)
    targets = lib.download(0, n_t)
    grads, zgaps = lib.download_variants(0, n_t)
    for i in range(2):  # device generator of the variants == CPU twin
        z, g = oracle.synth_variants(0xC0FFEE, 10 + i)
        assert np.array_equal(zgaps[i], z) and np.array_equal(grads[i], g)
    roi = np.zeros((H, W), dtype=np.uint8)
    roi[150:300, 300:800] = 1
    pm = np.repeat(np.arange(n_masks), n_t).astype(np.int32)
    ps = np.tile(np.arange(n_t), n_masks).astype(np.int64)
    for mirror, use_roi in ((True, False), (False, False), (True, True)):
        alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, 20, mirror, roi if use_roi else None, rects)
        gap, high, mir = alg.scoreLibraryPairs(lib, pm, ps)
        for m in range(n_masks):
            sq = oracle.ShapeQuery(masks[m], 20, mirror, rects, roi if use_roi else None)
            for t in range(n_t):
                want = sq.score(targets[t], grads[t], zgaps[t])
                k = m * n_t + t
                assert (int(gap[k]), int(high[k]), bool(mir[k])) == (want[0], want[1], want[3]), (mirror, use_roi, m, t)
    # a slot without variants reports (-1,-1,false); it does not poison its neighbours
    lib.upload(n_t, targets[:1])
    alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks[0], 20, True, None, rects)
    gap, high, mir = alg.scoreLibraryPairs(lib, [0, 0], [n_t, 0])
    assert (gap[0], high[0], mir[0]) == (-1, -1, False) and gap[1] >= 0


def test_gray8_gradient_images(cds, fixtures, oracle):
    """The reference's scorer reads the gradient through ImageArray.get, so a GRAY8 gradient (ByteImageArray.java:3-16) is as good as a GRAY16
    one; the C ABI widens it (cds_library_upload_gradient_u8).  The reference's own gradient fixtures clipped to 8 bits, two slots, against the
    oracle fed the same values, and against the 16-bit upload of the same clipped image."""
    keys = ("lm_VT033614", "lm_BJD")
    g8 = np.stack([np.minimum(fixtures["grad_VT033614"], 255).astype(np.uint8), np.minimum(fixtures["grad_BJD"], 255).astype(np.uint8)])
    targets = np.stack([fixtures[k] for k in keys])
    lib8, lib16 = cds.Library(W, H, 2), cds.Library(W, H, 2)
    for lib, g in ((lib8, g8), (lib16, g8.astype(np.uint16))):
        lib.upload(0, targets)
        lib.upload_variants(0, gradients=g)
        lib.make_zgap_variants(0, 2, 20, 10, cds.label_regions)
    grads, zgaps = lib8.download_variants(0, 2)
    assert grads.dtype == np.uint16 and np.array_equal(grads, g8)
    alg = cds.Shape2DMatchColorDepthSearchAlgorithm(fixtures["em_12191"], 20, True, None, cds.label_regions)
    got8, got16 = alg.scoreLibraryPairs(lib8, [0, 0], [0, 1]), alg.scoreLibraryPairs(lib16, [0, 0], [0, 1])
    for a, b in zip(got8, got16):
        assert np.array_equal(a, b)
    sq = oracle.ShapeQuery(fixtures["em_12191"], 20, True, E.STD_RECTS, None)
    for t in range(2):
        want = sq.score(targets[t], grads[t], zgaps[t])
        assert (int(got8[0][t]), int(got8[1][t]), bool(got8[2][t])) == (want[0], want[1], want[3]), t
    assert int(got8[0][0]) > 0


def test_device_side_zgap_variants(cds, fixtures, oracle):
    """ZGapImage variants generated on the device from resident targets equal the reference's CPU construction, and
    scoring through them reproduces the goldens (12191 x VT033614 -> 21365 / 731, Shape2DMatch...Test.java:86-94)."""
    keys = ("lm_VT033614", "lm_BJD")
    lib = cds.Library(W, H, len(keys))
    lib.upload(0, np.stack([fixtures[k] for k in keys]))
    lib.upload_variants(0, gradients=np.stack([fixtures["grad_VT033614"], fixtures["grad_BJD"]]))
    lib.make_zgap_variants(0, len(keys), 20, 10, cds.label_regions)
    _, z = lib.download_variants(0, len(keys))
    for i, k in enumerate(keys):
        assert np.array_equal(z[i], oracle.make_zgap(fixtures[k], 20, 10, E.STD_RECTS))
    alg = cds.Shape2DMatchColorDepthSearchAlgorithm(fixtures["em_12191"], 20, True, None, cds.label_regions)
    gap, high, mir = alg.scoreLibraryPairs(lib, [0, 0], [0, 1])
    assert (int(gap[0]), int(high[0]), bool(mir[0])) == (21365, 731, False)
    assert (int(gap[1]), int(high[1]), bool(mir[1])) == (23359, 523, False)


@pytest.mark.parametrize("w,h", [(131, 70), (200, 64)])
def test_small_and_odd_geometry(cds, oracle, w, h):
    """Non-standard image sizes (odd width: the reversed zslice run is unaligned), one custom excluded rectangle."""
    rng = np.random.default_rng(w)
    lut = oracle.depth_lut()
    mask = np.zeros((h, w, 3), dtype=np.uint8)
    mask[20:45, 30:90] = lut[rng.integers(0, 256, size=(25, 60))]
    mask[rng.random((h, w)) < 0.5] = 0
    rects = [(0, 0, 25, 12)]
    n = 4
    targets = np.zeros((n, h, w, 3), dtype=np.uint8)
    for i in range(n):
        t = lut[rng.integers(0, 256, size=(h, w))].astype(np.uint8)
        t[rng.random((h, w)) < 0.7] = 0
        targets[i] = t
    targets[1] = np.fliplr(mask)
    grads = rng.integers(0, 600, size=(n, h, w)).astype(np.uint16)
    zgaps = np.stack([oracle.make_zgap(targets[i], 20, 3, rects) for i in range(n)])
    lib = cds.Library(w, h, n)
    lib.upload(0, targets)
    lib.upload_variants(0, grads, zgaps)
    for mirror in (True, False):
        alg = cds.Shape2DMatchColorDepthSearchAlgorithm(mask, 20, mirror, None, rects)
        gap, high, mir = alg.scoreLibraryPairs(lib, [0] * n, list(range(n)))
        sq = oracle.ShapeQuery(mask, 20, mirror, rects)
        for i in range(n):
            want = sq.score(targets[i], grads[i], zgaps[i])
            assert (int(gap[i]), int(high[i]), bool(mir[i])) == (want[0], want[1], want[3]), (w, mirror, i)
    # device-side zgap generation at this geometry
    lib.make_zgap_variants(0, n, 20, 3, rects)
    _, z = lib.download_variants(0, n)
    assert np.array_equal(z, zgaps)


def test_random_geometry_and_parameter_sweep(cds, oracle):
    """Seeded random sweep for the shape scorer: image sizes (odd and even widths), several masks per set, thresholds,
    mirroring, ROI masks, excluded rectangles, 8/16-bit-range gradients, sparse and dense targets."""
    rng = np.random.default_rng(int(os.environ.get("CDS_TEST_SWEEP_SEED", "424242")))
    lut = oracle.depth_lut()

    def strokes(w, h, n, lo=0.3):
        img = np.zeros((h, w, 3), dtype=np.uint8)
        for _ in range(n):
            bw, bh = int(rng.integers(1, max(2, w // 3))), int(rng.integers(1, max(2, h // 3)))
            x0, y0 = int(rng.integers(0, w - bw + 1)), int(rng.integers(0, h - bh + 1))
            z = rng.integers(0, 256, size=(bh, bw))
            br = rng.uniform(lo, 1.0, size=(bh, bw, 1))
            img[y0:y0 + bh, x0:x0 + bw] = (lut[z] * br).astype(np.uint8)
        return img

    for case in range(int(os.environ.get("CDS_TEST_SWEEP_CASES", "10"))):
        w, h = int(rng.integers(40, 300)), int(rng.integers(24, 140))
        n_masks, n_t = int(rng.integers(1, 4)), 5
        masks = np.stack([strokes(w, h, int(rng.integers(1, 4))) for _ in range(n_masks)])
        targets = np.stack([strokes(w, h, int(rng.integers(1, 9)), lo=0.05) for _ in range(n_t - 2)] +
                           [masks[0], np.fliplr(masks[-1])])
        thr = int(rng.choice([0, 20, 50]))
        mirror = bool(rng.integers(0, 2))
        rects = [(0, 0, int(w // 5), int(h // 6))] if rng.integers(0, 2) else []
        roi = (rng.random((h, w)) < 0.7).astype(np.uint8) if rng.integers(0, 3) == 0 else None
        grads = rng.integers(0, int(rng.choice([5, 255, 700, 65535])), size=(n_t, h, w)).astype(np.uint16)
        zgaps = np.stack([oracle.make_zgap(targets[i], thr, int(rng.choice([1, 3, 10])), rects) for i in range(n_t)])
        lib = cds.Library(w, h, n_t)
        lib.upload(0, targets)
        lib.upload_variants(0, grads, zgaps)
        alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, thr, mirror, roi, rects)
        pm = np.repeat(np.arange(n_masks), n_t)
        ps = np.tile(np.arange(n_t), n_masks)
        gap, high, mir = alg.scoreLibraryPairs(lib, pm, ps)
        for k, (m, t) in enumerate(zip(pm, ps)):
            want = oracle.ShapeQuery(masks[m], thr, mirror, rects, roi).score(targets[t], grads[t], zgaps[t])
            assert (int(gap[k]), int(high[k]), bool(mir[k])) == (want[0], want[1], want[3]), (case, w, h, m, t, thr, mirror, roi is not None)


def test_normalized_scores_per_mask(cds, oracle):
    # GradientAreaGapUtils.calculateNormalizedScore applied per mask as CalculateGradientScoresCmd.java:616-645 does
    pix = np.array([636, 500, 0, 679])
    shape = np.array([788, 1114361, 50, -1])
    got = [cds.calculateNormalizedScore(int(p), int(s), int(pix.max()), int(shape.max())) for p, s in zip(pix, shape)]
    want = [oracle.normalized_score(int(p), int(s), int(pix.max()), int(shape.max())) for p, s in zip(pix, shape)]
    assert got == want


@pytest.mark.parametrize("border", [10, 37])
def test_query_border_size(cds, fixtures, oracle, border):
    """queryBorderSize != 0 (ColorDepthSearchAlgorithmProviderFactory.java:103; the CLI's --borderSize): every fold over the
    query skips a frame of that many pixels (LImage.java:89-113).  A mask with signal INSIDE the frame (it still feeds the max
    filters), the scorer against the oracle for plain / mirrored winners and with an ROI; mask sums and getQuerySize too."""
    rng = np.random.default_rng(border)
    lut = oracle.depth_lut()
    mask = fixtures["em_12191"].copy()
    mask[120:200, 2:60] = lut[rng.integers(0, 256, size=(80, 58))]            # signal inside the left frame and next to it
    mask[H - 30:H - 3, 500:700] = lut[rng.integers(0, 256, size=(27, 200))]   # ... and in the bottom frame
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createShapeMatchCDSAlgorithmProvider(True, None, cds.label_regions)
    alg = provider.createColorDepthQuerySearchAlgorithmWithDefaultParams(mask, 20, border)
    alg0 = provider.createColorDepthQuerySearchAlgorithmWithDefaultParams(mask, 20, 0)
    q = oracle.ShapeQuery(mask, 20, True, E.STD_RECTS, border=border)
    sm, sh = alg.maskStats()
    assert (int(sm[0]), int(sh[0])) == (int(q.QM.sum()), int(q.QH.sum()))
    assert sm[0] < alg0.maskStats()[0][0]                                     # the frame really held mask pixels
    inner = np.zeros((H, W), dtype=bool)
    inner[border:H - border, border:W - border] = True
    cleared = mask.copy()
    for (x0, y0, x1, y1) in E.STD_RECTS:
        cleared[y0:y1, x0:x1] = 0
    assert alg.getQuerySize() == int(((cleared.max(axis=2) > 20) & inner).sum())
    differs = 0
    for tk, gk in (("lm_VT033614", "grad_VT033614"), ("lm_BJD", "grad_BJD"), ("lm_VT016795", "grad_VT016795")):
        t = fixtures[tk].copy()
        t[100:220, 0:80] = lut[rng.integers(0, 256, size=(120, 80))]          # target signal under the frame as well
        zg = oracle.make_zgap(t, 20, 10, E.STD_RECTS)
        sup = {cds.ComputeFileType.GradientImage: (lambda g=fixtures[gk]: g), cds.ComputeFileType.ZGapImage: (lambda z=zg: z)}
        s = alg.calculateMatchingScore(t, sup)
        want = q.score(t, fixtures[gk], zg)
        assert (s.getGradientAreaGap(), s.getHighExpressionArea(), s.getScore(), s.isMirrored()) == want, tk
        s0 = alg0.calculateMatchingScore(t, sup)
        differs += (s0.getGradientAreaGap(), s0.getHighExpressionArea()) != (want[0], want[1])
    assert differs > 0                                                        # the border changes the result on these inputs
    roi = np.zeros((H, W), dtype=np.uint8)
    roi[60:400, 0:800] = 1
    alg_r = cds.Shape2DMatchColorDepthSearchAlgorithm(mask, 20, True, roi, cds.label_regions, queryBorderSize=border)
    q_r = oracle.ShapeQuery(mask, 20, True, E.STD_RECTS, roi=roi, border=border)
    t = fixtures["lm_VT016795"]
    zg = oracle.make_zgap(t, 20, 10, E.STD_RECTS)
    s = alg_r.calculateMatchingScore(t, {cds.ComputeFileType.GradientImage: lambda: fixtures["grad_VT016795"], cds.ComputeFileType.ZGapImage: lambda: zg})
    assert (s.getGradientAreaGap(), s.getHighExpressionArea(), s.getScore(), s.isMirrored()) == q_r.score(t, fixtures["grad_VT016795"], zg)


def test_where_the_slice_numbers_come_from(cds, oracle, s_of_z_source):
    """The lazy zslice plane (round 4): uploading a zgap converts nothing; a scoring call converts the slots it needs only when it
    brings enough pairs per slot (default 8), otherwise the scorer reads the zgap bytes; converted slots stay converted until their
    zgap is written again; both sources give identical sums, equal to the oracle's."""
    if s_of_z_source != "policy":
        pytest.skip("the policy itself is under test")
    rects = cds.label_regions(W)
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(9)])
    lib = cds.Library(W, H, 4)
    lib.generate_synthetic(0xC0FFEE, 60, 0, 4, kind=0, with_variants=True)
    alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, 20, True, None, rects)
    # 2 pairs per slot: the bytes
    few = alg.scoreLibraryPairs(lib, [0, 1, 2, 3], [0, 0, 1, 1])
    assert alg.lastCallInfo()[:2] == (False, 0)
    # 9 pairs on slot 2: its plane is made (one slot converted), slot 3 is not touched
    pm = np.arange(9, dtype=np.int32)
    many = alg.scoreLibraryPairs(lib, pm, np.full(9, 2))
    plane, conv, ms = alg.lastCallInfo()
    assert (plane, conv) == (True, 1) and ms > 0
    # again: nothing left to convert, still the plane; mixed with a stale slot and few pairs per stale slot: the bytes
    again = alg.scoreLibraryPairs(lib, pm, np.full(9, 2))
    assert alg.lastCallInfo()[:2] == (True, 0)
    assert all(np.array_equal(a, b) for a, b in zip(many, again))
    mixed = alg.scoreLibraryPairs(lib, [0, 1, 0], [2, 2, 3])
    assert alg.lastCallInfo()[0] is False
    assert (mixed[0][0], mixed[1][0]) == (many[0][0], many[1][0]) and (mixed[0][1], mixed[1][1]) == (many[0][1], many[1][1])
    # the oracle on both sources
    for (res, rows) in ((few, [(0, 0), (1, 0), (2, 1), (3, 1)]), (many, [(m, 2) for m in (0, 4, 8)])):
        for k, (m, t) in enumerate(rows):
            kk = k if res is few else m
            zg, gr = oracle.synth_variants(0xC0FFEE, 60 + t)
            want = oracle.ShapeQuery(masks[m], 20, True, rects).score(oracle.synth_image(0xC0FFEE, 60 + t, 0), gr, zg)
            assert (int(res[0][kk]), int(res[1][kk]), bool(res[2][kk])) == (want[0], want[1], want[3])
    # a new zgap in a converted slot makes its plane stale: the next many-pair call converts it again and sees the new image
    z3, _ = oracle.synth_variants(0xC0FFEE, 63)
    lib.upload_variants(2, zgaps=z3[None])
    changed = alg.scoreLibraryPairs(lib, pm, np.full(9, 2))
    assert alg.lastCallInfo()[:2] == (True, 1)
    _, g2 = oracle.synth_variants(0xC0FFEE, 62)
    want = oracle.ShapeQuery(masks[5], 20, True, rects).score(oracle.synth_image(0xC0FFEE, 62, 0), g2, z3)
    assert (int(changed[0][5]), int(changed[1][5]), bool(changed[2][5])) == (want[0], want[1], want[3])
    assert not np.array_equal(changed[0], many[0])
