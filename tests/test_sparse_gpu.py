"""GPU parity tests of kernel 1 over a SPARSE resident library (cds_sparse_library, cds_search_sparse*) through the C ABI:
the record stream against a numpy model of its specification, the reference's known answers on its own fixture images,
oracle parity (all 18 per-transform counts) on synthetic libraries, random geometries and the edge cases of the
reference's semantics, and equality with the dense scan at sizes the oracle cannot check.  Bar: bit-exact."""
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


@pytest.fixture(autouse=True, params=["stream16", "records32"])
def scan_kernel(request, monkeypatch):
    """Every test of this module runs over both single-mask scan kernels: the one that streams the 16-bit stage-1 stream of the
    library (round 4, the default whenever the library holds that stream) and round 3's, which streams the 4-byte records
    (CDS_SPARSE_SCAN32=1, read per launch; also what a library without the stream gets).  Same results, bit for bit."""
    monkeypatch.delenv("CDS_SPARSE_SCAN32", raising=False)
    monkeypatch.delenv("CDS_SPARSE_SCAN16", raising=False)
    monkeypatch.setenv("CDS_SPARSE_SCAN32" if request.param == "records32" else "CDS_SPARSE_SCAN16", "1")   # (forced: small searches default to the records)
    return request.param


def sparse_of(cds, images, w=None, h=None, slack=0):
    images = np.ascontiguousarray(images, dtype=np.uint8)
    if images.ndim == 3:
        images = images[None]
    n, hh, ww, _ = images.shape
    sl = cds.SparseLibrary(ww, hh, n + slack, int((images.max(axis=3) > 0).sum()) + 16 * n * hh + 64)   # (<= 16 strips: one filler per empty cell)
    sl.append(images)
    return sl


_RCP = {}


def device_rcp():
    """v_rcp_f32(0..255) of the GPU under test (the band of a record is made with it)."""
    if "t" not in _RCP:
        from colormipsearch_amd import cds as m
        _RCP["t"] = m.default_context().device_rcp_table()
    return _RCP["t"]


def records_model(img, n_strips=1, strip_w=None):
    """The specification of the record stream (include/cds_hip.h, cds_internal.hpp): strip by strip (column strips of
    strip_w pixels), row by row, x ascending, the pixels with a strict maximum channel as
    (x - strip start) << 22 | band << 16 | second << 8 | max  with  band = class * 5 + floor(5 * fl32(second) * rcp[max])
    (float32 arithmetic, the device's reciprocal),  bit 31 on the first record of a (strip, row) cell; a cell without such a
    pixel holds the single filler record 1 << 31."""
    img = img.astype(np.int64)
    strip_w = strip_w or img.shape[1]
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    mx = img.max(axis=2)
    md = np.median(img, axis=2).astype(np.int64)
    cls = np.zeros(mx.shape, dtype=np.int64)
    bmax, gmax, rmax = (b > r) & (b > g), (g > b) & (g > r), (r > b) & (r > g)
    cls[bmax & (r > g)] = 1          # BR
    cls[bmax & ~(r > g)] = 2         # BG
    cls[gmax & (b > r)] = 3          # GB
    cls[gmax & ~(b > r)] = 4         # GR
    cls[rmax & (g > b)] = 5          # RG
    cls[rmax & ~(g > b)] = 6         # RB
    out, offs, pos = [], np.zeros((n_strips, img.shape[0] + 1), dtype=np.uint32), 0
    for k in range(n_strips):
        for y in range(img.shape[0]):
            offs[k, y] = pos
            xs = np.flatnonzero(cls[y, k * strip_w:(k + 1) * strip_w] > 0) + k * strip_w
            if len(xs) == 0:
                rec = np.array([1 << 31], dtype=np.int64)
            else:
                hf = (md[y, xs].astype(np.float32) * device_rcp()[mx[y, xs]]).astype(np.float32)
                band = cls[y, xs] * 5 + (hf * np.float32(5.0)).astype(np.int64)
                rec = ((xs - k * strip_w) << 22) | (band << 16) | (md[y, xs] << 8) | mx[y, xs]
                rec[0] |= 1 << 31
            out.append(rec)
            pos += len(rec)
        offs[k, img.shape[0]] = pos
    return np.concatenate(out).astype(np.uint32), offs


def test_record_stream_matches_its_specification(cds, fixtures, oracle):
    rng = np.random.default_rng(5)
    imgs = [fixtures["lm_VT016795"], oracle.synth_image(0xC0FFEE, 4, 0), np.zeros((H, W, 3), dtype=np.uint8),
            oracle.synth_image(0xC0FFEE, 2, 1)]
    ties = np.zeros((H, W, 3), dtype=np.uint8)   # no strict maximum anywhere except one pixel per kind of class
    ties[5, 5:50] = (90, 90, 10)
    ties[6, 5:50] = (128, 128, 128)
    ties[7, W - 1] = (1, 0, 0)
    ties[H - 1, 0] = (0, 0, 255)
    ties[9, 3:9] = [(10, 5, 20), (5, 10, 20), (5, 20, 10), (10, 20, 5), (20, 10, 5), (20, 5, 10)]
    imgs.append(ties)
    imgs.append(rng.integers(0, 4, size=(H, W, 3)).astype(np.uint8))   # small values: many ties, zero numerators
    sl = sparse_of(cds, np.stack(imgs))
    assert len(sl) == len(imgs)
    total = 0
    for i, img in enumerate(imgs):
        rec, rows = sl.image_records(i)
        want_rec, want_rows = records_model(img, sl.n_strips, sl.strip_width)
        assert np.array_equal(rows, want_rows), i
        assert np.array_equal(rec, want_rec), i
        total += len(rec)
    assert sl.pixels == total


def test_append_batches_capacity_and_clear(cds, oracle):
    dense = cds.Library(W, H, 6)
    dense.generate_synthetic(0xC0FFEE, 20, 0, 6, kind=0)
    imgs = dense.download(0, 6)
    probe = cds.SparseLibrary(W, H, 1, 16)
    ns, sw = probe.n_strips, probe.strip_width
    assert (ns, sw) == (4, 304)
    n_rec = [len(records_model(im, ns, sw)[0]) for im in imgs]
    sl = cds.SparseLibrary(W, H, 6, sum(n_rec[:5]) + 10)
    sl.append_from(dense, 0, 2)
    sl.append(imgs[2:4])
    sl.append_from(dense, 4, 1)
    assert len(sl) == 5 and sl.pixels == sum(n_rec[:5])
    for i in range(5):
        assert np.array_equal(sl.image_records(i)[0], records_model(imgs[i], ns, sw)[0]), i
    with pytest.raises(cds.CdsError, match="full"):      # record capacity
        sl.append_from(dense, 5, 1)
    assert len(sl) == 5
    sl.clear()
    assert len(sl) == 0 and sl.pixels == 0
    sl.append_from(dense, 3, 3)
    assert np.array_equal(sl.image_records(2)[0], records_model(imgs[5], ns, sw)[0])
    with pytest.raises(cds.CdsError):                    # image capacity
        sl.append_from(dense, 0, 4)
    with pytest.raises(ValueError):
        sl.append(np.zeros((1, 10, 10, 3), dtype=np.uint8))


# ---- the reference's JUnit answers ------------------------------------------------------------------
def test_reference_known_answers(cds, fixtures):
    # PixelMatchColorDepthSearchAlgorithmTest.pixelMatchScore (:33-53) and multiplePixelScores (:56-153)
    e = E.PIXEL_MATCH_DIRECT
    alg = cds.PixelMatchColorDepthSearchAlgorithm(fixtures[e["mask"]], 20, True, None, 0, False, 20, 0.01, 2, e["rects"])
    s, m = alg.searchSparseLibrary(sparse_of(cds, fixtures[e["target"]]))
    assert (int(s[0, 0]), bool(m[0, 0])) == (87, False)
    for mask, target, expected, mirrored in E.PIXEL_MATCH_PROVIDER:
        alg = cds.PixelMatchColorDepthSearchAlgorithm(fixtures[mask], 20, True, None, 0, False, 20, 0.01, 2, E.STD_RECTS)
        s, m = alg.searchSparseLibrary(sparse_of(cds, fixtures[target]))
        assert (int(s[0, 0]), bool(m[0, 0])) == (expected, mirrored), (mask, target)


@pytest.mark.parametrize("lds_kb", [None, 24, 160])
def test_fixture_matrix_all_transforms(cds, fixtures, oracle, lds_kb, monkeypatch):
    """3 EM masks x 4 LM targets, every per-transform count, 4 parameter combinations; one slab, the default layout and
    many slabs per mask."""
    if lds_kb:
        monkeypatch.setenv("CDS_SPARSE_LDS_KB", str(lds_kb))
    masks = np.stack([fixtures[k] for k in ("em_12191", "em_12191_FL", "em_LPLC2")])
    tkeys = ("lm_BJD", "lm_GMR", "lm_VT016795", "lm_VT033614")
    targets = np.stack([fixtures[k] for k in tkeys])
    sl = sparse_of(cds, targets)
    for xy, mirror in ((2, True), (2, False), (0, True), (0, False)):
        alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, mirror, None, 0, False, 20, 0.01, xy, E.STD_RECTS)
        scores, mir = alg.searchSparseLibrary(sl)
        counts = alg.searchSparseLibraryCounts(sl)
        for m in range(3):
            q = oracle.PixelMatchQuery(masks[m], 20, 20, 0.01, xy, mirror, E.STD_RECTS)
            for t in range(len(tkeys)):
                s, mm, c = q.score(targets[t])
                assert np.array_equal(counts[m, t], c), (xy, mirror, m, t, counts[m, t], c)
                assert (int(scores[m, t]), bool(mir[m, t])) == (s, mm), (xy, mirror, m, t)


# ---- synthetic libraries ------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,n", [(0, 48), (1, 6)])
def test_synthetic_library_parity(cds, oracle, kind, n):
    """BASELINE config 2 at a size the oracle finishes in seconds: 1 mask x n targets, xyShift 2, mirror."""
    rects = cds.label_regions(W)
    mask = oracle.synth_image(0xBADC0DE, 0, 2)
    lib = cds.Library(W, H, n)
    lib.generate_synthetic(0xC0FFEE, 0, 0, n, kind=kind)
    targets = lib.download(0, n)
    sl = cds.SparseLibrary(W, H, n, n * W * H)
    sl.append_from(lib)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
    scores, mir = alg.searchSparseLibrary(sl)
    q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, 2, True, rects)
    want_s, want_m = q.scan(targets, n_threads=8)
    assert np.array_equal(scores[0], want_s)
    assert np.array_equal(mir[0], want_m.astype(bool))
    if kind == 1:
        assert scores.max() > 0  # dense targets do hit


def test_self_match_mirror_and_sub_ranges(cds, oracle):
    rects = cds.label_regions(W)
    mask = oracle.synth_image(0xBADC0DE, 3, 2)
    targets = np.stack([mask, np.fliplr(mask), np.roll(mask, 2, axis=1), np.roll(mask, -2, axis=0), np.zeros_like(mask)] +
                       [oracle.synth_image(0xC0FFEE, 60 + i, 0) for i in range(5)])
    sl = sparse_of(cds, targets)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
    scores, mir = alg.searchSparseLibrary(sl)
    P = alg.getQuerySize()
    q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, 2, True, rects)
    for t in range(len(targets)):
        s, m, _ = q.score(targets[t])
        assert (int(scores[0, t]), bool(mir[0, t])) == (s, m)
    assert scores[0, 0] <= P and scores[0, 0] > 0.9 * P and not mir[0, 0]
    assert mir[0, 1] and scores[0, 4] == 0
    part, pm = alg.searchSparseLibrary(sl, 3, 4)
    assert np.array_equal(part, scores[:, 3:7]) and np.array_equal(pm, mir[:, 3:7])
    empty, _ = alg.searchSparseLibrary(sl, 2, 0)
    assert empty.shape == (1, 0)


def test_batched_masks_and_parameter_sweep(cds, oracle):
    rects = cds.label_regions(W)
    n_masks, n_t = 5, 12
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(n_masks)])
    targets = np.stack([oracle.synth_image(0xC0FFEE, 100 + i, 0) for i in range(n_t - 2)] + [masks[1], np.fliplr(masks[2])])
    sl = sparse_of(cds, targets)
    for (mthr, dthr, ztol, xy, mirror) in ((20, 20, 0.01, 2, True), (100, 100, 0.02, 0, False), (20, 50, 0.3, 2, False)):
        alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, mthr, mirror, None, 0, False, dthr, ztol, xy, rects)
        scores, mir = alg.searchSparseLibrary(sl)
        for m in range(n_masks):
            q = oracle.PixelMatchQuery(masks[m], mthr, dthr, ztol, xy, mirror, rects)
            ws, wm = q.scan(targets, n_threads=8)
            assert np.array_equal(scores[m], ws), (m, mthr, dthr, ztol, xy, mirror)
            assert np.array_equal(mir[m], wm.astype(bool))


def test_wide_masks_and_many_colours(cds, oracle):
    """(a) a whole-frame mask (strokes all over the image: the bounding box is the frame), (b) a very wide image,
    (c) a mask with thousands of distinct colours: its interval table does not fit the LDS table and is read from L2."""
    rects = cds.label_regions(W)
    mask = oracle.synth_image(0xC0FFEE, 9001, 0)
    targets = np.stack([oracle.synth_image(0xC0FFEE, i, 0) for i in (9001, 9002, 9003)] + [np.fliplr(mask), oracle.synth_image(0xC0FFEE, 3, 1)])
    sl = sparse_of(cds, targets)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
    counts = alg.searchSparseLibraryCounts(sl)
    q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, 2, True, rects)
    for t in range(len(targets)):
        assert np.array_equal(counts[0, t], q.score(targets[t])[2]), t
    w2, h2 = 1600, 64
    rng = np.random.default_rng(11)
    lut = oracle.depth_lut()
    m2 = np.zeros((h2, w2, 3), dtype=np.uint8)
    for x0 in (3, 700, 1500):
        m2[10:50, x0:x0 + 40] = lut[rng.integers(0, 256, size=(40, 40))]
    t2 = np.stack([m2, np.fliplr(m2), lut[rng.integers(0, 256, size=(h2, w2))].astype(np.uint8)])
    sl2 = sparse_of(cds, t2)
    a2 = cds.PixelMatchColorDepthSearchAlgorithm(m2, 20, True, None, 0, False, 20, 0.01, 2, [])
    c2 = a2.searchSparseLibraryCounts(sl2)
    q2 = oracle.PixelMatchQuery(m2, 20, 20, 0.01, 2, True, [])
    for t in range(len(t2)):
        assert np.array_equal(c2[0, t], q2.score(t2[t])[2]), t
    w3, h3 = 400, 200
    m3 = np.zeros((h3, w3, 3), dtype=np.uint8)
    m3[20:180, 50:350] = rng.integers(0, 256, size=(160, 300, 3))
    t3 = np.stack([m3, np.fliplr(m3), rng.integers(0, 256, size=(h3, w3, 3)).astype(np.uint8), np.roll(m3, 2, axis=0)])
    sl3 = sparse_of(cds, t3)
    a3 = cds.PixelMatchColorDepthSearchAlgorithm(m3, 20, True, None, 0, False, 20, 0.02, 2, [])
    c3 = a3.searchSparseLibraryCounts(sl3)
    q3 = oracle.PixelMatchQuery(m3, 20, 20, 0.02, 2, True, [])
    for t in range(len(t3)):
        assert np.array_equal(c3[0, t], q3.score(t3[t])[2]), t


# ---- edge cases of the reference's semantics ------------------------------------------------------------
@pytest.mark.parametrize("lds_kb", [None, 8])
def test_mask_touching_every_border(cds, oracle, lds_kb, monkeypatch):
    """Shifted positions falling off the image are skipped (-1, PixelMatch...Algorithm.java:132-144,241-243),
    mirrored columns use the full width; also a small non-standard geometry."""
    if lds_kb:
        monkeypatch.setenv("CDS_SPARSE_LDS_KB", str(lds_kb))
        monkeypatch.setenv("CDS_SPARSE_THREADS", "64")
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
    sl = sparse_of(cds, targets)
    for xy, mirror in ((2, True), (0, True), (2, False)):
        alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, mirror, None, 0, False, 20, 0.01, xy, [])
        counts = alg.searchSparseLibraryCounts(sl)
        q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, xy, mirror, [])
        for t in range(len(targets)):
            assert np.array_equal(counts[0, t], q.score(targets[t])[2]), (xy, mirror, t)


def test_off_centre_masks_and_the_two_orientation_map(cds, oracle, monkeypatch, scan_kernel):
    """The stream kernel's band map covers the UNION of the mask's window and its mirror image (one 8-byte cell serves both
    orientations): a mask at the far left or right of the frame makes that union as wide as the frame, a mask hugging the centre
    makes the two windows overlap cell by cell with different alignments, a mask of one column makes them one cell wide.  All 18 counts
    against the oracle, targets that hit the window's edge cells included; and a mask whose map would exceed the 64 KB that 16-bit
    cell offsets can address is scanned through the 4-byte records even when the stream is forced."""
    rects = cds.label_regions(W)
    rng = np.random.default_rng(31)
    lut = oracle.depth_lut()
    base = [oracle.synth_image(0xC0FFEE, 700 + i, 0) for i in range(3)]
    cases = []
    for name, x0, x1, y0, y1 in (("far left", 2, 60, 200, 330), ("far right", W - 75, W - 3, 100, 180), ("centre, odd alignment", W // 2 - 37, W // 2 + 52, 250, 400),
                                 ("one column", 411, 412, 90, 300), ("left of centre", W // 2 - 150, W // 2 - 20, 20, 90)):
        mask = np.zeros((H, W, 3), dtype=np.uint8)
        sel = rng.random((y1 - y0, x1 - x0)) < 0.35
        mask[y0:y1, x0:x1][sel] = lut[rng.integers(0, 256, int(sel.sum()))]
        cases.append((name, mask))
    targets = []
    for name, mask in cases:
        for t in base:
            u = t.copy()
            m = mask.max(axis=2) > 0
            u[m] = mask[m]                                   # the mask itself laid over a synthetic target: hits in every window cell
            targets.append(np.roll(u, rng.integers(-2, 3), axis=1))
            targets.append(np.fliplr(u))
    targets = np.stack(targets)
    sl = sparse_of(cds, targets)
    for name, mask in cases:
        alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.02, 2, rects)
        counts = alg.searchSparseLibraryCounts(sl)
        if scan_kernel == "stream16":
            assert alg.lastScanKernel()[0] == 1, name
        q = oracle.PixelMatchQuery(mask, 20, 20, 0.02, 2, True, rects)
        for t in range(0, len(targets), 3):
            assert np.array_equal(counts[0, t], q.score(targets[t])[2]), (name, t)
        assert counts.max() > 0, name
    # a tall mask at the left edge: union window = the whole frame, 1,224 bytes per cell row x 130 cell rows > 64 KB -> the 4-byte records
    tall = np.zeros((H, W, 3), dtype=np.uint8)
    tall[20:530, 3:40][rng.random((510, 37)) < 0.2] = lut[77]
    alg = cds.PixelMatchColorDepthSearchAlgorithm(tall, 20, True, None, 0, False, 20, 0.02, 2, rects)
    counts = alg.searchSparseLibraryCounts(sl)
    assert alg.lastScanKernel()[0] == 0
    q = oracle.PixelMatchQuery(tall, 20, 20, 0.02, 2, True, rects)
    for t in (0, 7, len(targets) - 1):
        assert np.array_equal(counts[0, t], q.score(targets[t])[2]), t


@pytest.mark.parametrize("threads,lds_kb", [(None, None), (256, 12)])
def test_random_geometry_and_parameter_sweep(cds, oracle, threads, lds_kb, monkeypatch):
    """Seeded random sweep over image sizes, mask counts / shapes, thresholds, tolerances, shift and mirror settings:
    all 18 per-transform counts of every (mask, target) pair against the oracle, in the default layout and with small
    workgroups / many slabs."""
    if threads:
        monkeypatch.setenv("CDS_SPARSE_THREADS", str(threads))
        monkeypatch.setenv("CDS_SPARSE_LDS_KB", str(lds_kb))
    rng = np.random.default_rng(int(os.environ.get("CDS_TEST_SWEEP_SEED", "20261004")))
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

    for case in range(int(os.environ.get("CDS_TEST_SWEEP_CASES", "40"))):
        w, h = int(rng.integers(24, 420)), int(rng.integers(8, 150))
        n_masks = int(rng.choice([1, 1, 2, 3]))
        masks = np.stack([blobs(w, h, int(rng.integers(1, 5)), False) for _ in range(n_masks)])
        targets = [blobs(w, h, int(rng.integers(1, 8)), bool(rng.integers(0, 2))) for _ in range(4)]
        targets += [masks[0], np.fliplr(masks[-1]), np.roll(masks[0], 2, axis=1), np.zeros_like(masks[0])]
        targets = np.stack(targets)
        mthr, dthr = int(rng.choice([0, 20, 60, 150])), int(rng.choice([0, 20, 90]))
        ztol = float(rng.choice([0.0, 0.01, 0.02, 0.1, 0.5]))
        xy, mirror = int(rng.choice([0, 2])), bool(rng.integers(0, 2))
        rects = [(0, 0, w // 4, h // 5)] if rng.integers(0, 2) else []
        alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, mthr, mirror, None, 0, False, dthr, ztol, xy, rects)
        sl = sparse_of(cds, targets)
        counts = alg.searchSparseLibraryCounts(sl)
        scores, mir = alg.searchSparseLibrary(sl)
        for m in range(n_masks):
            q = oracle.PixelMatchQuery(masks[m], mthr, dthr, ztol, xy, mirror, rects)
            for t in range(len(targets)):
                s, mi, c = q.score(targets[t])
                info = (case, w, h, n_masks, m, t, mthr, dthr, ztol, xy, mirror)
                assert np.array_equal(counts[m, t], c), info
                assert scores[m, t] == s and bool(mir[m, t]) == mi, info


def test_empty_classless_masks_and_errors(cds, oracle, fixtures):
    rects = cds.label_regions(W)
    black = np.zeros((H, W, 3), dtype=np.uint8)
    gray = black.copy()
    gray[200:220, 400:500] = 128          # r == g == b: no hue class, never matches, but counts in the query size
    target = oracle.synth_image(0xC0FFEE, 1, 0)
    sl = sparse_of(cds, np.stack([target, gray]))
    for mask, size in ((black, 0), (gray, 2000)):
        alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
        assert alg.getQuerySize() == size
        s, m = alg.searchSparseLibrary(sl)
        assert not s.any() and not m.any()
    mixed = np.stack([black, oracle.synth_image(0xBADC0DE, 1, 2), gray])     # empty masks inside a set
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mixed, 20, True, None, 0, False, 20, 0.01, 2, rects)
    s, m = alg.searchSparseLibrary(sl)
    ws, wm = oracle.PixelMatchQuery(mixed[1], 20, 20, 0.01, 2, True, rects).scan(np.stack([target, gray]))
    assert not s[0].any() and not s[2].any() and np.array_equal(s[1], ws) and np.array_equal(m[1], wm.astype(bool))
    small = sparse_of(cds, np.zeros((1, 100, 200, 3), dtype=np.uint8))
    with pytest.raises(ValueError, match="Invalid image size"):                           # PixelMatch...:171-175
        alg.searchSparseLibrary(small)
    with pytest.raises(ValueError):
        alg.searchSparseLibrary(sl, 1, 5)


def test_equals_dense_scan_at_scale(cds, oracle):
    """At a library size the oracle cannot check directly: the sparse scan equals the dense scan target by target
    (scores and flags), is deterministic, and a permuted library gives the permuted scores."""
    rects = cds.label_regions(W)
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(2)])
    n = 4096
    lib = cds.Library(W, H, n)
    lib.generate_synthetic(0xC0FFEE, 0, 0, n, kind=0)
    sl = cds.SparseLibrary(W, H, n, 45000 * n)
    sl.append_from(lib, 0, 1000)
    sl.append_from(lib, 1000, n - 1000)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects)
    d_s, d_m = alg.searchLibrary(lib)
    s1, m1 = alg.searchSparseLibrary(sl)
    s2, m2 = alg.searchSparseLibrary(sl)
    assert np.array_equal(s1, d_s) and np.array_equal(m1, d_m)
    assert np.array_equal(s1, s2) and np.array_equal(m1, m2)
    assert (s1 > 0).sum() > 100
    sl2 = cds.SparseLibrary(W, H, 64, 45000 * 64)
    for k in range(64):
        sl2.append_from(lib, 1063 - k, 1)
    s3, _ = alg.searchSparseLibrary(sl2)
    assert np.array_equal(s3[:, ::-1], s1[:, 1000:1064])


@pytest.mark.parametrize("strip_w", [None, 64, 20])
def test_odd_geometries_thresholds_and_strip_widths(cds, oracle, strip_w, monkeypatch):
    """Widths that are no multiple of 4, single-row and single-column-strip images, thresholds at the ends of the byte
    range, masks at the image corners; with the default strips, 64-pixel strips and 20-pixel strips (up to 16 runs per
    target and slab)."""
    if strip_w:
        monkeypatch.setenv("CDS_SPARSE_STRIP_W", str(strip_w))
    rng = np.random.default_rng(31)
    lut = oracle.depth_lut()
    for (w, h) in ((9, 1), (13, 7), (63, 5), (65, 33), (130, 2), (257, 19), (318, 40)):
        if strip_w == 20 and w > 320:
            continue
        mask = np.zeros((h, w, 3), dtype=np.uint8)
        mask[: max(1, h // 2), : max(2, w // 3)] = lut[rng.integers(0, 256, size=(max(1, h // 2), max(2, w // 3)))]
        mask[h - 1, w - 3:] = lut[77]
        targets = np.stack([mask, np.fliplr(mask), lut[rng.integers(0, 256, size=(h, w))].astype(np.uint8), np.zeros_like(mask),
                            rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)])
        sl = sparse_of(cds, targets)
        assert sl.n_strips == -(-w // sl.strip_width) and sl.n_strips <= 16
        for (mthr, dthr, ztol, xy, mirror) in ((0, 0, 0.01, 2, True), (20, 254, 0.5, 2, True), (20, 255, 0.5, 0, True), (254, 20, 0.1, 2, False),
                                               (20, 20, 0.0, 0, False)):
            alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, mthr, mirror, None, 0, False, dthr, ztol, xy, [])
            counts = alg.searchSparseLibraryCounts(sl)
            q = oracle.PixelMatchQuery(mask, mthr, dthr, ztol, xy, mirror, [])
            for t in range(len(targets)):
                assert np.array_equal(counts[0, t], q.score(targets[t])[2]), (w, h, mthr, dthr, ztol, xy, mirror, t)


def test_dense_full_frame_mask(cds, oracle):
    """Every pixel of the frame is a mask pixel (684 K of them): the plane is cut into several slabs, every record of every
    target is near the mask; sparse and dense targets."""
    rects = cds.label_regions(W)
    rng = np.random.default_rng(99)
    lut = oracle.depth_lut()
    mask = lut[rng.integers(0, 256, size=(H, W))].astype(np.uint8)
    targets = np.stack([oracle.synth_image(0xC0FFEE, 11, 0), oracle.synth_image(0xC0FFEE, 12, 1), mask])
    sl = sparse_of(cds, targets)
    q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, 2, True, rects)
    want = np.stack([q.score(t)[2] for t in targets])
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
    got = alg.searchSparseLibraryCounts(sl)[0]
    assert np.array_equal(got, want)
    assert got[2, 4] > 0.9 * alg.getQuerySize()   # the mask against itself, unshifted


def test_every_target_colour_against_uniform_masks(cds, oracle):
    """The hue-band map of stage 1 must never drop a pixel the evaluation would count.  One target holds EVERY (class, second,
    max) combination a record can have (6 x 32,640 = 195,840 pixels in a 400 x 500 block); masks are uniform blocks of one colour
    each -- ratios on and next to the band boundaries k/5, ratio 0, the colours next to a class change -- so every target pixel
    meets that colour under every shift; several tolerances, all 18 counts against the oracle."""
    hh, ww = 500, 400
    combos = []
    for mx in range(1, 256):
        sec = np.arange(mx)
        lo = np.zeros_like(sec)   # the third channel: anything not above the second one; 0 keeps the class unambiguous
        m = np.full_like(sec, mx)
        for order in ((2, 0, 1), (2, 1, 0), (1, 2, 0), (1, 0, 2), (0, 1, 2), (0, 2, 1)):   # (max channel, second, third)
            px = np.zeros((len(sec), 3), dtype=np.uint8)
            px[:, order[0]], px[:, order[1]], px[:, order[2]] = m, sec, lo
            combos.append(px)
    combos = np.concatenate(combos)
    assert len(combos) == 195840
    target = np.zeros((H, W, 3), dtype=np.uint8)
    block = np.zeros((hh * ww, 3), dtype=np.uint8)
    block[: len(combos)] = combos
    y0, x0 = 40, 420
    target[y0:y0 + hh, x0:x0 + ww] = block.reshape(hh, ww, 3)
    sl = sparse_of(cds, target)
    colours = [(255, 51, 0), (255, 50, 0), (255, 52, 3), (250, 100, 0), (250, 101, 7), (200, 120, 10), (0, 200, 160), (0, 201, 160),
               (10, 100, 255), (255, 0, 0), (30, 0, 255), (0, 255, 254), (254, 255, 0), (128, 127, 3), (64, 255, 13), (21, 20, 1)]
    for ci, col in enumerate(colours):
        mask = np.zeros((H, W, 3), dtype=np.uint8)
        mask[y0:y0 + hh, x0:x0 + ww] = col
        for ztol in ((0.01, 0.05, 0.25) if ci % 4 == 0 else (0.01,)):
            alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, ztol, 2, [])
            counts = alg.searchSparseLibraryCounts(sl)
            want = oracle.PixelMatchQuery(mask, 20, 20, ztol, 2, True, []).score(target)[2]
            assert np.array_equal(counts[0, 0], want), (col, ztol, counts[0, 0].tolist(), list(want))


def test_record_arena_beyond_2_pow_32(cds, oracle):
    """A library whose record arena crosses 2^32 records (8,192 dense-uniform images = 5.6 G records, 22 GB): the 64-bit
    paths -- img_off, the buffer-descriptor base of the scan, the append kernels' offsets -- are checked on the images stored
    BEYOND 2^32: every score against the dense-layout kernel, the first and the last images beyond the line (scores and all
    18 per-transform counts) against the oracle."""
    rects = cds.label_regions(W)
    n_stage, reps = 1024, 8
    n = n_stage * reps
    mask = oracle.synth_image(0xBADC0DE, 0, 2)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
    dense = cds.Library(W, H, n_stage)
    sl = cds.SparseLibrary(W, H, n, (W * H + 4 * H) * n)
    for r in range(reps):
        dense.generate_synthetic(0xC0FFEE, r * n_stage, 0, n_stage, kind=1)
        sl.append_from(dense, 0, n_stage)
    assert len(sl) == n and sl.pixels > 2 ** 32
    offs = sl.image_offsets()
    assert offs[0] == 0 and int(offs[-1]) == sl.pixels and np.all(np.diff(offs.astype(np.int64)) > 0)
    first_beyond = int(np.searchsorted(offs[:-1], np.uint64(2 ** 32)))
    assert 0 < first_beyond < n - n_stage          # more than a window of images lies beyond the line
    scores, mir = alg.searchSparseLibrary(sl)
    for r in range(reps):                          # every image, also those beyond 2^32: the dense kernel on regenerated slots
        dense.generate_synthetic(0xC0FFEE, r * n_stage, 0, n_stage, kind=1)
        s2, m2 = alg.searchLibrary(dense)
        assert np.array_equal(s2[0], scores[0, r * n_stage:(r + 1) * n_stage]), r
        assert np.array_equal(m2[0], mir[0, r * n_stage:(r + 1) * n_stage]), r
    assert scores.max() > 0
    q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, 2, True, rects)
    for lo in (first_beyond, n - 4):               # (the last window is still in `dense`)
        cnt = alg.searchSparseLibraryCounts(sl, lo, 4)
        part, pm = alg.searchSparseLibrary(sl, lo, 4)   # a sub-range whose first image starts beyond 2^32
        for i in range(4):
            img = oracle.synth_image(0xC0FFEE, lo + i, 1)
            s, m, c = q.score(img)
            assert (int(scores[0, lo + i]), bool(mir[0, lo + i])) == (s, m) == (int(part[0, i]), bool(pm[0, i])), lo + i
            assert np.array_equal(cnt[0, i], c), lo + i
    # the record stream of an image beyond the line equals its specification
    img = oracle.synth_image(0xC0FFEE, n - 1, 1)
    rec, rows = sl.image_records(n - 1)
    want_rec, want_rows = records_model(img, sl.n_strips, sl.strip_width)
    assert np.array_equal(rec, want_rec) and np.array_equal(rows, want_rows)


# ---- the multi-mask scan (groups of 8 masks share one pass over the records) ---------------------------------------
def test_multi_mask_groups_equal_single_mask_scans_and_oracle(cds, oracle, fixtures, monkeypatch):
    """19 masks = two full groups and one of 3: synthetic arbors, an EM fixture with few colours, a mask elsewhere in the
    frame (the group's union box grows), a mask touching the borders, an empty mask and a mask of one pixel.  Targets:
    synthetic ones, copies / mirror images / shifts of masks (full matches), a real LM image.  The scores of the multi-mask
    kernel == those of one launch per mask (CDS_SPARSE_MULTI=0) == the oracle's, for three parameter sets."""
    rects = cds.label_regions(W)
    lut = oracle.depth_lut()
    rng = np.random.default_rng(21)
    masks = [oracle.synth_image(0xBADC0DE, i, 2) for i in range(13)]
    masks.append(fixtures["em_12191"])
    far = np.zeros((H, W, 3), dtype=np.uint8)
    far[400:520, 20:140] = lut[rng.integers(0, 256, size=(120, 120))]
    masks.append(far)
    border = np.zeros((H, W, 3), dtype=np.uint8)
    border[:, :3] = lut[7]; border[:, W - 3:] = lut[200]; border[:2, :] = lut[90]; border[H - 2:, :] = lut[150]
    masks.append(border)
    masks.append(np.zeros((H, W, 3), dtype=np.uint8))
    one = np.zeros((H, W, 3), dtype=np.uint8)
    one[300, 600] = lut[128]
    masks.append(one)
    masks.append(fixtures["em_LPLC2"])
    masks = np.stack(masks)
    targets = np.stack([oracle.synth_image(0xC0FFEE, 500 + i, 0) for i in range(10)] +
                       [masks[1], np.fliplr(masks[2]), np.roll(masks[3], 2, axis=1), far, np.fliplr(border), border, one,
                        fixtures["lm_VT016795"], oracle.synth_image(0xC0FFEE, 5, 1)])
    sl = sparse_of(cds, targets)
    for (mthr, dthr, ztol, xy, mirror) in ((20, 20, 0.01, 2, True), (20, 30, 0.02, 0, True), (50, 20, 0.01, 2, False)):
        alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, mthr, mirror, None, 0, False, dthr, ztol, xy, rects)
        monkeypatch.delenv("CDS_SPARSE_MULTI", raising=False)
        groups, multi_groups, multi_masks = alg.sparseGroups()
        assert multi_groups >= 2 and multi_masks >= 16, (groups, multi_groups, multi_masks)   # (the kernel under test really runs)
        s_multi, m_multi = alg.searchSparseLibrary(sl)
        monkeypatch.setenv("CDS_SPARSE_MULTI", "0")
        s_one, m_one = alg.searchSparseLibrary(sl)
        monkeypatch.delenv("CDS_SPARSE_MULTI", raising=False)
        assert np.array_equal(s_multi, s_one) and np.array_equal(m_multi, m_one), (mthr, dthr, ztol, xy, mirror, np.argwhere(s_multi != s_one)[:5])
        for m in range(len(masks)):
            q = oracle.PixelMatchQuery(masks[m], mthr, dthr, ztol, xy, mirror, rects)
            ws, wm = q.scan(targets, n_threads=8)
            assert np.array_equal(s_multi[m], ws), (m, mthr, dthr, ztol, xy, mirror)
            assert np.array_equal(m_multi[m], wm.astype(bool)), m
        part, pm = alg.searchSparseLibrary(sl, 9, 7)           # a sub-range
        assert np.array_equal(part, s_multi[:, 9:16]) and np.array_equal(pm, m_multi[:, 9:16])
    assert s_multi[1, 10] > 0 and s_multi[14, 13] > 0 and s_multi[16].max() == 0


def test_multi_mask_large_library_equals_dense_batch(cds, oracle):
    """16 masks x 4,096 synthetic targets: multi-mask sparse scan == the dense batch kernel == per-mask sparse launches."""
    rects = cds.label_regions(W)
    n_m, n_t = 16, 4096
    mlib = cds.Library(W, H, n_m)
    mlib.generate_synthetic(0xBADC0DE, 0, 0, n_m, kind=2)
    masks = mlib.download(0, n_m)
    dense = cds.Library(W, H, n_t)
    dense.generate_synthetic(0xC0FFEE, 70000, 0, n_t, kind=0)
    sl = cds.SparseLibrary(W, H, n_t, 24000 * n_t + 4 * H * n_t)
    sl.append_from(dense)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects)
    assert alg.sparseGroups()[2] == n_m
    s1, m1 = alg.searchSparseLibrary(sl)
    s2, m2 = alg.searchLibrary(dense)
    assert np.array_equal(s1, s2) and np.array_equal(m1, m2)
    assert (s1 > 0).sum() > 100


def test_multi_mask_random_geometry(cds, oracle, monkeypatch):
    """Seeded random sweep for the multi-mask kernel: 9 .. 24 masks (full groups, a ragged last group, empty masks among them)
    over random image sizes, thresholds, tolerances, shift / mirror settings, colour-rich masks (2-byte planes) and LUT masks:
    scores and mirrored flags of every pair against the oracle and against one launch per mask."""
    rng = np.random.default_rng(int(os.environ.get("CDS_TEST_SWEEP_SEED", "20261005")))
    lut = oracle.depth_lut()

    def blobs(w, h, n, rich):
        img = np.zeros((h, w, 3), dtype=np.uint8)
        for _ in range(n):
            bw, bh = int(rng.integers(1, max(2, w // 3))), int(rng.integers(1, max(2, h // 3)))
            x0, y0 = int(rng.integers(0, w - bw + 1)), int(rng.integers(0, h - bh + 1))
            if rich:
                img[y0:y0 + bh, x0:x0 + bw] = rng.integers(0, 256, size=(bh, bw, 3))
            else:
                z = rng.integers(0, 256, size=(bh, bw))
                br = rng.uniform(0.3, 1.0, size=(bh, bw, 1))
                img[y0:y0 + bh, x0:x0 + bw] = (lut[z] * br).astype(np.uint8)
        return img

    ran_multi = 0
    for case in range(int(os.environ.get("CDS_TEST_MULTI_SWEEP_CASES", "10"))):
        w, h = int(rng.integers(40, 520)), int(rng.integers(12, 160))
        n_masks = int(rng.integers(9, 25))
        masks = [blobs(w, h, int(rng.integers(1, 4)), bool(rng.integers(0, 4) == 0)) for _ in range(n_masks)]
        masks[int(rng.integers(0, n_masks))] = np.zeros((h, w, 3), dtype=np.uint8)      # an empty mask inside a group
        masks = np.stack(masks)
        targets = [blobs(w, h, int(rng.integers(1, 8)), bool(rng.integers(0, 2))) for _ in range(5)]
        targets += [masks[0], np.fliplr(masks[-1]), np.roll(masks[n_masks // 2], 2, axis=0), np.zeros_like(masks[0])]
        targets = np.stack(targets)
        mthr, dthr = int(rng.choice([0, 20, 60])), int(rng.choice([0, 20, 90]))
        ztol = float(rng.choice([0.0, 0.01, 0.02, 0.1]))
        xy, mirror = int(rng.choice([0, 2])), bool(rng.integers(0, 2))
        rects = [(0, 0, w // 4, h // 5)] if rng.integers(0, 2) else []
        alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, mthr, mirror, None, 0, False, dthr, ztol, xy, rects)
        sl = sparse_of(cds, targets)
        monkeypatch.delenv("CDS_SPARSE_MULTI", raising=False)
        ran_multi += alg.sparseGroups()[1]
        scores, mir = alg.searchSparseLibrary(sl)
        monkeypatch.setenv("CDS_SPARSE_MULTI", "0")
        s_one, m_one = alg.searchSparseLibrary(sl)
        monkeypatch.delenv("CDS_SPARSE_MULTI", raising=False)
        info = (case, w, h, n_masks, mthr, dthr, ztol, xy, mirror)
        assert np.array_equal(scores, s_one) and np.array_equal(mir, m_one), (info, np.argwhere(scores != s_one)[:5])
        for m in range(n_masks):
            q = oracle.PixelMatchQuery(masks[m], mthr, dthr, ztol, xy, mirror, rects)
            ws, wm = q.scan(targets, n_threads=8)
            assert np.array_equal(scores[m], ws) and np.array_equal(mir[m], wm.astype(bool)), (info, m)
    assert ran_multi >= 10    # (the kernel under test really ran)


def test_stream_read_diagnostic(cds, oracle):
    """cds_debug_sparse_stream_read (bench.py's reference point): a plain read of the record arena reports a plausible rate and
    refuses a library too small to measure."""
    n = 512
    dense = cds.Library(W, H, n)
    dense.generate_synthetic(0xC0FFEE, 0, 0, n, kind=0)
    sl = cds.SparseLibrary(W, H, n, 24000 * n + 4 * H * n)
    sl.append_from(dense)
    gbs, nbytes = sl.stream_read_gbs()
    assert nbytes >= 16 << 20 and 50.0 < gbs < 9000.0, (gbs, nbytes)
    tiny = sparse_of(cds, oracle.synth_image(0xC0FFEE, 1, 0))
    with pytest.raises(cds.CdsInvalidArgument):
        tiny.stream_read_gbs()


def test_load_only_diagnostic_switches_on_and_off(cds, oracle):
    """cds_debug_sparse_load_only (bench.py's memory-floor measurement): while on, the single-mask search launches the loads-only
    twin (its scores are meaningless, the launch must complete); switched off, the scan is the real one again.  A layout without a
    twin (two workgroups per CU) refuses instead of running something else."""
    rects = cds.label_regions(W)
    mask = oracle.synth_image(0xBADC0DE, 0, 2)
    targets = np.stack([oracle.synth_image(0xC0FFEE, 900 + i, 0) for i in range(6)] + [mask])
    sl = sparse_of(cds, targets)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
    want, wm = alg.searchSparseLibrary(sl)
    ctx = cds.default_context()
    # the flag lives in the context: a second context on the same GPU keeps searching for real while the first one's flag is on
    ctx2 = cds.Context(0)
    sl2 = cds.SparseLibrary(W, H, len(targets), int((targets.max(axis=3) > 0).sum()) + 4 * H * len(targets) + 64, context=ctx2)
    sl2.append(targets)
    alg2 = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects, context=ctx2)
    try:
        ctx.sparse_load_only(True)
        assert ctx.sparse_load_only_state() and not ctx2.sparse_load_only_state()
        alg.searchSparseLibrary(sl)
        got2, gm2 = alg2.searchSparseLibrary(sl2)
        assert np.array_equal(got2, want) and np.array_equal(gm2, wm)
        small = np.zeros((H, W, 3), dtype=np.uint8)
        small[300:310, 500:520] = oracle.depth_lut()[100]
        alg_small = cds.PixelMatchColorDepthSearchAlgorithm(small, 20, True, None, 0, False, 20, 0.01, 2, rects)
        if os.environ.get("CDS_SPARSE_SCAN32"):   # the 4-byte-record kernel in its 80 KB layout: no twin
            with pytest.raises(cds.CdsError):
                alg_small.searchSparseLibrary(sl)
        else:                                     # the stream kernel has a twin for its two-workgroups-per-CU layout
            alg_small.searchSparseLibrary(sl)
    finally:
        ctx.sparse_load_only(False)
    got, gm = alg.searchSparseLibrary(sl)
    assert np.array_equal(got, want) and np.array_equal(gm, wm) and got[0, -1] > 0


def test_library_minimum_data_threshold(cds, oracle, fixtures, tmp_path):
    """cds_sparse_library_set_min_threshold: pixels at or below the library's minimum data threshold are not stored -- the record
    stream of an image equals that of the image with those pixels blacked, the scores of every search at or above the threshold equal
    those over a library that keeps everything (and the oracle's), a search below it is refused, the threshold can be raised but not
    lowered once images are held, and a shard file carries it."""
    from colormipsearch_amd import mipio
    rects = cds.label_regions(W)
    rng = np.random.default_rng(5)
    lut = oracle.depth_lut()
    dim = np.zeros((H, W, 3), dtype=np.uint8)                      # strokes of every brightness, many of them dim
    for k in range(40):
        y0, x0 = int(rng.integers(150, 340)), int(rng.integers(330, 860))
        br = rng.uniform(0.02, 1.0)
        dim[y0:y0 + 6, x0:x0 + 40] = (lut[rng.integers(0, 256, size=(6, 40))] * br).astype(np.uint8)
    targets = np.stack([dim, fixtures["lm_VT016795"], fixtures["lm_BJD"], oracle.synth_image(0xC0FFEE, 77, 0), fixtures["em_12191"]])
    T = 40
    full = sparse_of(cds, targets)
    lean = cds.SparseLibrary(W, H, len(targets), int((targets.max(axis=3) > 0).sum()) + 8 * H * len(targets) + 64, min_threshold=T)
    assert lean.min_threshold == T and full.min_threshold == 0
    lean.append(targets)
    assert lean.pixels < 0.9 * full.pixels
    blacked = targets.copy()
    blacked[targets.max(axis=3) <= T] = 0
    ref = sparse_of(cds, blacked)
    for i in range(len(targets)):
        a, ra = lean.image_records(i)
        b, rb = ref.image_records(i)
        assert np.array_equal(a, b) and np.array_equal(ra, rb), i
    masks = np.stack([fixtures["em_12191"], oracle.synth_image(0xBADC0DE, 0, 2), dim])
    for dthr in (T, T + 25, 200):
        alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, dthr, 0.01, 2, rects)
        s1, m1 = alg.searchSparseLibrary(lean)
        s0, m0 = alg.searchSparseLibrary(full)
        assert np.array_equal(s1, s0) and np.array_equal(m1, m0), dthr
        q = oracle.PixelMatchQuery(masks[0], 20, dthr, 0.01, 2, True, rects)
        ws, wm = q.scan(targets, n_threads=8)
        assert np.array_equal(s1[0], ws) and np.array_equal(m1[0], wm.astype(bool)), dthr
        one = cds.PixelMatchColorDepthSearchAlgorithm(masks[2], 20, True, None, 0, False, dthr, 0.01, 2, rects)
        assert np.array_equal(one.searchSparseLibrary(lean)[0], one.searchSparseLibrary(full)[0])
    assert s1[0, 4] > 0
    below = cds.PixelMatchColorDepthSearchAlgorithm(masks[0], 20, True, None, 0, False, T - 1, 0.01, 2, rects)
    with pytest.raises(cds.CdsInvalidArgument, match="minimum data threshold"):
        below.searchSparseLibrary(lean)
    with pytest.raises(cds.CdsInvalidArgument):
        lean.set_min_threshold(T - 10)              # its records between T - 10 and T are gone
    lean.set_min_threshold(T + 5)                   # raising is fine (the library then merely holds a few pixels it need not)
    assert lean.min_threshold == T + 5
    path = str(tmp_path / "lean.cdss")
    hd = mipio.write_sparse_shard(path, lean)
    assert hd["min_threshold"] == T + 5
    with pytest.raises(ValueError, match="minimum data threshold"):
        mipio.load_sparse_shard(cds.SparseLibrary(W, H, 8, lean.pixels + 4096), path)
    back = cds.SparseLibrary(W, H, 8, lean.pixels + 4096, min_threshold=T + 5)
    assert mipio.load_sparse_shard(back, path) == len(targets)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 60, 0.01, 2, rects)
    assert np.array_equal(alg.searchSparseLibrary(back)[0], alg.searchSparseLibrary(full)[0])


# ---- match list: ColorMIPSearch.isMatch + compaction on the device ------------------------------------------------------
def test_device_match_list_equals_host_filter(cds, oracle, monkeypatch):
    """cds_search_sparse_matches == scan + the host-side isMatch (oracle.is_match restates ColorMIPSearch.java:42-45) for several
    thresholds, with an empty mask (skipped like the reference's "empty" masks), in one chunk and in chunks of 3 masks, and
    with a buffer that is too small (the needed length is reported)."""
    import ctypes
    from colormipsearch_amd import capi
    rects = cds.label_regions(W)
    masks = np.stack([oracle.synth_image(0xBADC0DE, i, 2) for i in range(10)] + [np.zeros((H, W, 3), dtype=np.uint8)])
    n_t = 1500                                                   # (more than one 1024-target block per mask)
    dense = cds.Library(W, H, n_t)
    dense.generate_synthetic(0xC0FFEE, 31000, 0, n_t, kind=0)
    dense.upload(7, np.stack([masks[2], np.fliplr(masks[5]).copy()]))
    sl = cds.SparseLibrary(W, H, n_t, 24000 * n_t + 4 * H * n_t)
    sl.append_from(dense)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects)
    scores, mir = alg.searchSparseLibrary(sl)
    sizes = alg.getQuerySizes()
    assert sizes[10] == 0
    # the oracle's side of the list: the scores the filter runs over are the oracle's for two masks on the first 96 targets
    # (the planted copies among them), so for those rows the device list below IS the oracle pipeline scan -> isMatch
    imgs96 = dense.download(0, 96)
    for m_or in (2, 5):
        ws, wm = oracle.PixelMatchQuery(masks[m_or], 20, 20, 0.01, 2, True, rects).scan(imgs96, n_threads=8)
        assert np.array_equal(ws, scores[m_or, :96]) and np.array_equal(wm.astype(bool), mir[m_or, :96])
    for pct in (0.0, 0.01, 0.5, 2.0, 50.0):
        want = [(m, t, int(scores[m, t]), bool(mir[m, t])) for m in range(len(masks)) for t in range(n_t)
                if sizes[m] > 0 and oracle.is_match(int(scores[m, t]), int(sizes[m]), pct)]
        # one chunk of masks: list sized by a guess, one wait (and, with a guess of 1 row, the refill at the exact size); chunks of 3
        # masks: the length read back between count and fill, the list grown chunk by chunk
        for chunk, guess in ((None, None), (None, "1"), ("3", None)):
            for k, v in (("CDS_MATCH_CHUNK_MASKS", chunk), ("CDS_MATCH_GUESS_ROWS", guess)):
                if v:
                    monkeypatch.setenv(k, v)
                else:
                    monkeypatch.delenv(k, raising=False)
            m, t, s, mr = alg.searchSparseLibraryMatches(sl, pct)
            assert list(zip(m.tolist(), t.tolist(), s.tolist(), mr.tolist())) == want, (pct, chunk, guess, len(want), len(m))
    monkeypatch.delenv("CDS_MATCH_CHUNK_MASKS", raising=False)
    monkeypatch.delenv("CDS_MATCH_GUESS_ROWS", raising=False)
    assert len(want) >= 2                                        # the two planted copies pass even 50 %
    # the same rows as table columns (cds_match_list_fetch_columns): mask offset applied, ratio = (float)((double) pixels / query size)
    want0 = [(m, t, int(scores[m, t]), bool(mir[m, t])) for m in range(len(masks)) for t in range(n_t)
             if sizes[m] > 0 and oracle.is_match(int(scores[m, t]), int(sizes[m]), 0.01)]
    cm, ct, cs, cr, cmr = alg.searchSparseLibraryMatchColumns(sl, 0.01, mask_offset=40)
    assert (cm.dtype, ct.dtype, cs.dtype, cr.dtype, cmr.dtype) == (np.int32, np.int64, np.int32, np.float32, np.bool_)
    assert list(zip((cm - 40).tolist(), ct.tolist(), cs.tolist(), cmr.tolist())) == want0
    assert np.array_equal(cr, (cs.astype(np.float64) / np.maximum(np.asarray(sizes, dtype=np.float64)[cm - 40], 1.0)).astype(np.float32))
    part = alg.searchSparseLibraryMatches(sl, 0.01, 5, 100)      # a sub-range: target indices are relative to `first`
    want_p = [(m, t - 5) for m in range(len(masks)) for t in range(5, 105) if sizes[m] > 0 and oracle.is_match(int(scores[m, t]), int(sizes[m]), 0.01)]
    assert list(zip(part[0].tolist(), part[1].tolist())) == want_p
    buf = np.zeros((1, 3), dtype=np.int32)
    n = ctypes.c_int64()
    rc = capi.load().cds_search_sparse_matches(alg._h, sl._h, 0, n_t, 0.0, buf.ctypes.data_as(ctypes.c_void_p), 1, ctypes.byref(n))
    n_all = sum(1 for mm in range(len(masks)) for tt in range(n_t) if sizes[mm] > 0 and scores[mm, tt] > 0)
    assert rc == capi.CDS_ERR_OUT_OF_MEMORY and n.value == n_all


# ---- layout choice: a sparse library refuses images the dense layout scans faster ---------------------------------------
def test_density_limit_refuses_dense_images(cds, oracle):
    """Default policy (cds_sparse_library_set_density_limit, 0.25 of the frame): a batch of dense-uniform images is refused
    with a message that names the dense layout, nothing is appended; sparse images pass; the limit can be lifted; and
    processor.resident_library_for picks the layout accordingly."""
    from colormipsearch_amd import processor as P
    dense = cds.Library(W, H, 4)
    dense.generate_synthetic(0xC0FFEE, 0, 0, 4, kind=1)
    sl = cds.SparseLibrary(W, H, 8, 8 * (W * H + 4 * H), density_limit=0.25)
    with pytest.raises(cds.CdsError, match="dense-layout scan"):
        sl.append_from(dense, 0, 4)
    assert len(sl) == 0 and sl.pixels == 0
    imgs = dense.download(0, 2)
    with pytest.raises(cds.CdsError, match="density limit"):
        sl.append(imgs)
    dense.generate_synthetic(0xC0FFEE, 0, 0, 4, kind=0)
    sl.append_from(dense, 0, 4)                                   # 1-4 % signal: fine
    assert len(sl) == 4
    sl.set_density_limit(2.0)
    sl.append(imgs)
    assert len(sl) == 6
    lib_s = P.resident_library_for(dense.download(0, 4))
    lib_d = P.resident_library_for(imgs)
    assert isinstance(lib_s, cds.SparseLibrary) and isinstance(lib_d, cds.Library)
    mask = oracle.synth_image(0xBADC0DE, 0, 2)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, cds.label_regions(W))
    s_d, _ = alg.searchLibrary(lib_d)
    s_s, _ = alg.searchSparseLibrary(sl, 4, 2)
    assert np.array_equal(s_d, s_s)


def test_an_append_call_is_a_transaction(cds, oracle):
    """the density limit used to be tested per internal sub-batch (64 staged images), so a call mixing sparse and
    dense images failed midway and left a partially appended library.  Now the call's AVERAGE decides and a failing call leaves the
    library as it was -- whatever the reason (density, record capacity) and however the call is cut into batches."""
    sparse_imgs = np.stack([oracle.synth_image(0xC0FFEE, 300 + i, 0) for i in range(64)])
    dense_imgs = np.stack([oracle.synth_image(0xC0FFEE, i, 1) for i in range(3)])
    n_rec = int((sparse_imgs.max(axis=3) > 0).sum())
    sl = cds.SparseLibrary(W, H, 200, 3 * (W * H + 4 * H) + 4 * n_rec + 4 * H * 200, density_limit=0.05)
    sl.append(sparse_imgs[:5])
    before = (len(sl), sl.pixels)
    # 64 sparse images fill the first staging batch and pass on their own; the 3 dense ones that follow push the CALL's average
    # (3 x 100 % + 64 x ~3 % over 67 images = ~7 %) above the 5 % limit: nothing of the call stays
    with pytest.raises(cds.CdsError, match="density limit"):
        sl.append(np.concatenate([sparse_imgs, dense_imgs]))
    assert (len(sl), sl.pixels) == before
    # the other order (dense images first): refused as soon as it is certain, same outcome
    with pytest.raises(cds.CdsError, match="density limit"):
        sl.append(np.concatenate([dense_imgs, dense_imgs, dense_imgs, sparse_imgs]))
    assert (len(sl), sl.pixels) == before
    # a call whose average stays below the limit passes although one of its images is dense
    sl.set_density_limit(0.10)
    sl.append(np.concatenate([sparse_imgs, dense_imgs[:1]]))
    assert len(sl) == before[0] + 65
    # record capacity exceeded in the SECOND staging batch: the first batch's images do not stay either
    small = cds.SparseLibrary(W, H, 200, n_rec + 4 * H * 64 + (W * H) // 2, density_limit=2.0)
    with pytest.raises(cds.CdsError, match="full"):
        small.append(np.concatenate([sparse_imgs, dense_imgs[:1]]))
    assert (len(small), small.pixels) == (0, 0)
    # and the library is still usable: the same images score as they do in a fresh library
    small.append(sparse_imgs[:8])
    mask = oracle.synth_image(0xBADC0DE, 0, 2)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, cds.label_regions(W))
    got, _ = alg.searchSparseLibrary(small)
    want, _ = alg.searchSparseLibrary(sparse_of(cds, sparse_imgs[:8]))
    assert np.array_equal(got, want)


def test_import_checks_the_shards_minimum_data_threshold_in_the_c_abi(cds, oracle):
    """only mipio.py compared a shard's minimum data threshold with the library's; a C or Java caller could import
    records that lack the pixels <= T into a library that promises searches down to 0.  cds_sparse_library_import now takes the
    threshold the records were made with and refuses when it exceeds the library's own."""
    imgs = np.stack([oracle.synth_image(0xC0FFEE, 40 + i, 0) for i in range(3)])
    src = cds.SparseLibrary(W, H, 3, int((imgs.max(axis=3) > 0).sum()) + 4 * H * 3 + 64, min_threshold=60)
    src.append(imgs)
    offs, cells, rec = src.export()
    dst = cds.SparseLibrary(W, H, 3, len(rec) + 64)                       # promises searches down to data threshold 0
    with pytest.raises(cds.CdsInvalidArgument, match="minimum data threshold 60"):
        dst.import_arrays(offs, cells, rec, min_data_threshold=60)
    assert len(dst) == 0
    dst.set_min_threshold(60)
    dst.import_arrays(offs, cells, rec, min_data_threshold=60)
    assert len(dst) == 3 and dst.pixels == src.pixels
    with pytest.raises(cds.CdsInvalidArgument):
        dst.import_arrays(offs, cells, rec, min_data_threshold=300)


def test_device_topk_equals_host_selection(cds, oracle):
    """cds_search_sparse_topk == a stable host-side sort of the score rows (score descending, equal scores by ascending target),
    for k = 1, 37, 500, 1024 and more than the library holds, with many tied scores (zeros) in every row and a mask whose
    query size needs the two-level histogram (a whole-frame mask: scores up to hundreds of thousands)."""
    rects = cds.label_regions(W)
    masks = [oracle.synth_image(0xBADC0DE, i, 2) for i in range(9)]
    big = oracle.depth_lut()[np.random.default_rng(8).integers(0, 256, size=(H, W))].astype(np.uint8)   # a whole-frame mask: query size ~ W * H
    masks.append(big)
    masks = np.stack(masks)
    n_t = 2600
    dense = cds.Library(W, H, n_t)
    dense.generate_synthetic(0xC0FFEE, 52000, 0, n_t, kind=0)
    dense.upload(11, np.stack([masks[4], np.fliplr(masks[7]).copy(), big]))
    sl = cds.SparseLibrary(W, H, n_t, 30000 * n_t + 4 * H * n_t + W * H)
    sl.append_from(dense)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.01, 2, rects)
    assert alg.getQuerySize(9) > 65536
    scores, mir = alg.searchSparseLibrary(sl)
    for k in (1, 37, 500, 1024):
        t, s, m = alg.searchSparseLibraryTopK(sl, k)
        for mi in range(len(masks)):
            order = np.lexsort((np.arange(n_t), -scores[mi].astype(np.int64)))[:k]
            assert np.array_equal(t[mi], order), (k, mi, t[mi][:8], order[:8])
            assert np.array_equal(s[mi], scores[mi][order]) and np.array_equal(m[mi], mir[mi][order])
    t, s, m = alg.searchSparseLibraryTopK(sl, 64, 10, 40)          # fewer targets than k: the tail is -1
    for mi in (0, 4, 9):
        order = np.lexsort((np.arange(40), -scores[mi, 10:50].astype(np.int64)))
        assert np.array_equal(t[mi, :40], order) and (t[mi, 40:] == -1).all() and np.array_equal(s[mi, :40], scores[mi, 10:50][order])
    assert scores[9, 13] > 65536 and scores[4, 11] > 0


def test_stage1_stream_is_derived_from_the_records_and_optional(cds, oracle, monkeypatch, scan_kernel):
    """The 16-bit stream is derived data: a library created without it (CDS_SPARSE_STREAM16=0, or no device memory left for it) scans
    through the 4-byte records and gives the same scores; runs of odd length, data thresholds that only stage 2 can apply (the stream
    has no maximum channel) and targets appended after a rolled-back call are all covered by comparing the two kernels' 18 counts."""
    if scan_kernel != "stream16":
        pytest.skip("compares the two kernels itself")
    rects = cds.label_regions(W)
    rng = np.random.default_rng(21)
    lut = oracle.depth_lut()
    imgs = []
    for i in range(10):
        t = oracle.synth_image(0xC0FFEE, 800 + i, 0).copy()
        dim = rng.random((H, W)) < 0.3                       # many signal pixels at or below the data thresholds used below
        t[dim] = (t[dim].astype(np.int32) * 40 // 255).astype(np.uint8)
        ys, xs = rng.integers(0, H, 300), rng.integers(0, W, 300)
        t[ys, xs] = lut[rng.integers(0, 256, 300)]           # isolated pixels: rows with 1, 2, 3 records (odd run lengths)
        imgs.append(t)
    imgs = np.stack(imgs)
    mask = oracle.synth_image(0xBADC0DE, 3, 2)
    with_stream = sparse_of(cds, imgs)
    monkeypatch.setenv("CDS_SPARSE_STREAM16", "0")
    without = sparse_of(cds, imgs)
    monkeypatch.delenv("CDS_SPARSE_STREAM16")
    for thr, xy, mirror in ((20, 2, True), (60, 2, True), (35, 0, False), (100, 2, False)):
        alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, mirror, None, 0, False, thr, 0.02, xy, rects)
        c16 = alg.searchSparseLibraryCounts(with_stream)
        assert alg.lastScanKernel()[0] == 1
        monkeypatch.setenv("CDS_SPARSE_PLANE_LDS", "1")        # (the stream kernel with the plane in LDS instead of L2: same counts)
        assert np.array_equal(alg.searchSparseLibraryCounts(with_stream), c16)
        monkeypatch.delenv("CDS_SPARSE_PLANE_LDS")
        c32 = alg.searchSparseLibraryCounts(without)
        assert alg.lastScanKernel()[0] == 0                      # (no stream in that library: nothing to force)
        assert np.array_equal(c16, c32), (thr, xy, mirror)
        monkeypatch.delenv("CDS_SPARSE_SCAN16")
        monkeypatch.setenv("CDS_SPARSE_SCAN32", "1")
        assert np.array_equal(alg.searchSparseLibraryCounts(with_stream), c32)
        monkeypatch.delenv("CDS_SPARSE_SCAN32")
        monkeypatch.setenv("CDS_SPARSE_SCAN16", "1")
        q = oracle.PixelMatchQuery(mask, 20, thr, 0.02, xy, mirror, rects)
        for t in (0, 5, 9):
            assert np.array_equal(c16[0, t], q.score(imgs[t])[2]), (thr, xy, mirror, t)
        assert c16.max() > 0
    # after a refused (rolled back) append the stream of the NEXT append is right, the sentinel included (last image, last strip)
    lib = cds.SparseLibrary(W, H, 16, int((imgs.max(axis=3) > 0).sum()) + 4 * H * 16 + 64 + (W * H + 4 * H), density_limit=0.2)
    lib.append(imgs[:4])
    with pytest.raises(cds.CdsError):
        lib.append(np.stack([oracle.synth_image(0xC0FFEE, 1, 1)]))
    lib.append(imgs[4:])
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.02, 2, rects)
    assert np.array_equal(alg.searchSparseLibraryCounts(lib), alg.searchSparseLibraryCounts(without))


def test_a_growable_library_sizes_itself_and_the_cache_can_be_trimmed(cds, oracle, monkeypatch, scan_kernel):
    """cds_sparse_library_set_growable / _reserve / _capacity (the reference's MIP cache sizes itself with use, CachedMIPsUtils.java:60-100):
    a library created for 2 images and 1,000 records takes 23 images in four appends -- arrays regrown, records, row pointers and the
    16-bit stream copied -- and scans like one created at full size; a non-growable one still refuses; the stream is left out by
    policy when records + stream would exceed the configured share of the device's memory; cds_context_trim hands the context's
    idle blocks back."""
    rects = cds.label_regions(W)
    imgs = np.stack([oracle.synth_image(0xC0FFEE, 1500 + i, 0) for i in range(23)])
    mask = oracle.synth_image(0xBADC0DE, 5, 2)
    imgs[7] = mask
    full = sparse_of(cds, imgs)
    small = cds.SparseLibrary(W, H, 2, 1000)
    with pytest.raises(cds.CdsError, match="full"):
        small.append(imgs[:3])
    assert len(small) == 0
    small.set_growable(True)
    for a, b in ((0, 3), (3, 4), (4, 15), (15, 23)):
        small.append(imgs[a:b])
    assert len(small) == 23 and small.pixels == full.pixels
    assert small.capacity_images >= 23 and small.capacity_pixels >= small.pixels and small.has_stream16 == full.has_stream16
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
    want = alg.searchSparseLibraryCounts(full)
    assert np.array_equal(alg.searchSparseLibraryCounts(small), want) and want[0, 7].max() > 5000
    small.reserve(100, 10_000_000)                       # explicit growth keeps the contents
    assert small.capacity_images >= 100 and small.capacity_pixels >= 10_000_000
    assert np.array_equal(alg.searchSparseLibraryCounts(small), want)
    # default construction: growable from 1,024 images
    auto = cds.SparseLibrary(W, H)
    auto.append(imgs)
    assert np.array_equal(alg.searchSparseLibraryCounts(auto), want)
    # policy: no stream when records + stream would take more than the allowed share of the device
    monkeypatch.setenv("CDS_SPARSE_STREAM16_MAX_FRACTION", "1e-9")
    lean = sparse_of(cds, imgs)
    monkeypatch.delenv("CDS_SPARSE_STREAM16_MAX_FRACTION")
    assert not lean.has_stream16
    if scan_kernel == "stream16":
        monkeypatch.delenv("CDS_SPARSE_SCAN16")            # (nothing to force: the library has no stream)
    assert np.array_equal(alg.searchSparseLibraryCounts(lean), want)
    # the context's block cache: released mask-set blocks sit idle until trimmed
    ctx = cds.default_context()
    alg2 = cds.PixelMatchColorDepthSearchAlgorithm(np.stack([mask, imgs[3]]), 20, True, None, 0, False, 20, 0.01, 2, rects)
    alg2.searchSparseLibrary(full)
    alg2.close()
    assert ctx.trim() > 0 and ctx.trim() == 0
    ctx.set_cache_limit(0)                                 # keep nothing
    alg3 = cds.PixelMatchColorDepthSearchAlgorithm(np.stack([mask, imgs[3]]), 20, True, None, 0, False, 20, 0.01, 2, rects)
    alg3.searchSparseLibrary(full)
    alg3.close()
    assert ctx.trim() == 0
    ctx.set_cache_limit(6 << 30)


def test_the_record_behind_an_odd_run_is_not_scanned(cds, oracle, scan_kernel):
    """The geometry of round 4's one GPU-side abort (tree afdb1a9, case 13 of the seeded sweep: gpurun_out/r04e_tests.log).
    The stream kernel reads two records per lane, so a run of odd length also reads the record BEHIND it -- the first record of the
    next row, strip or image.  The first stream kernel let that record through as a pixel of row R + 4 (R = the slab's rows), counting on
    an empty map cell row there; for R mod 4 != 0 that row shares its cells with the halo rows R + 1 .. R + 3, the record passed the band
    map, and its evaluation read the plane's row R + 8, one past the plane (rows 0 .. R + 7): band-map bytes taken for a 16-bit interval
    index, and -- with more than 2,048 interval records the table lives in global memory -- a load up to 1 MB past a < 100 KB
    allocation: the fault.  Here: a mask of 22 rows (R mod 4 = 2) with > 2,048 interval records, xyShift 2; targets whose run over the
    mask's row window is odd and whose next row starts, in the same cell, with the colour of the mask's last row; and a second mask whose
    window ends with the image, so that the record behind the run is the next image's first one, or the sentinel behind the arena's last
    record.  All 18 counts against the oracle."""
    w, h = 200, 64
    rng = np.random.default_rng(5)

    def colours(n):   # distinct hue ratios by the thousand: maximum 200..255 in a random channel, the other two below it
        c = np.zeros((n, 3), dtype=np.uint8)
        mx = rng.integers(200, 256, n)
        ch = rng.integers(0, 3, n)
        for k in range(3):
            c[:, k] = np.where(ch == k, mx, rng.integers(1, 200, n))
        return c

    def mask_at(y0):
        m = np.zeros((h, w, 3), dtype=np.uint8)
        m[y0:y0 + 22, 20:160] = colours(22 * 140).reshape(22, 140, 3)
        return m

    top, bottom = mask_at(10), mask_at(40)       # bottom: rows 40 .. 61, its window [38, 64) ends with the image
    targets = []
    for extra in (0, 1, 2):                       # 0 / 1 / 2 more records in the window: the run's length is odd for one parity at least
        t = top.copy()
        t[34, 20:160] = top[31, 20:160]           # the row behind the window [8, 34) starts where -- and as -- the mask's last row does
        if extra >= 1:
            t[9, 50] = top[12, 50]
        if extra >= 2:
            t[9, 90] = top[12, 90]
        t[40:62, 20:160] = bottom[40:62, 20:160]  # (and the second mask's pixels: its run ends with the image's last record)
        if extra >= 1:
            t[39, 60] = bottom[41, 60]
        if extra >= 2:
            t[39, 61] = bottom[41, 61]
        targets.append(t)
    targets = np.stack(targets)
    sl = sparse_of(cds, targets)
    for name, mask in (("top", top), ("bottom", bottom)):
        for mirror in (False, True):
            alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, mirror, None, 0, False, 20, 0.01, 2, [])
            assert alg.sparseLayout(0)["interval_records"] > 2048, "the case needs the interval table in global memory"
            counts = alg.searchSparseLibraryCounts(sl)
            if scan_kernel == "stream16":
                assert alg.lastScanKernel()[0] == 1
            q = oracle.PixelMatchQuery(mask, 20, 20, 0.01, 2, mirror, [])
            for t in range(len(targets)):
                assert np.array_equal(counts[0, t], q.score(targets[t])[2]), (name, mirror, t)
            assert counts[0, :, 4].min() >= 22 * 140 * 0.9, name      # (the unshifted transform finds the mask in every target)


def test_the_pilot_picks_the_scan_kernel_per_mask_and_library(cds, oracle, fixtures, monkeypatch, scan_kernel):
    """Without a forced kernel the first search of >= 16,384 targets measures, on its first 4,096 targets, the share of visited records
    that reach an evaluation round, and scans the rest (and every later search of that library) through the 16-bit stream when the
    share is small -- the synthetic library, whose colours are unrelated to the mask's -- or through the 4-byte records when it is
    large -- the reference's LM images against an EM mask that overlaps them.  Scores equal the forced kernels' either way."""
    if scan_kernel != "stream16":
        pytest.skip("policy test: runs once")
    monkeypatch.delenv("CDS_SPARSE_SCAN16")
    n = 16384
    rects = cds.label_regions(W)
    # (a) synthetic library
    dense = cds.Library(W, H, 4096)
    sl = cds.SparseLibrary(W, H, n, 24000 * n + 4_000_000)
    for f in range(0, n, 4096):
        dense.generate_synthetic(0xC0FFEE, f, 0, 4096, kind=0)
        sl.append_from(dense, 0, 4096)
    mask = oracle.synth_image(0xBADC0DE, 0, 2)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(mask, 20, True, None, 0, False, 20, 0.01, 2, rects)
    assert alg.lastScanKernel() == (-1, -1.0)
    s_auto, m_auto = alg.searchSparseLibrary(sl)
    k, ratio = alg.lastScanKernel()
    assert k == 1 and 0 < ratio < 0.05, (k, ratio)
    s_again, _ = alg.searchSparseLibrary(sl, 100, 9000)            # (remembered: no second pilot, same kernel, any range)
    assert alg.lastScanKernel() == (1, ratio) and np.array_equal(s_again[0], s_auto[0, 100:9100])
    monkeypatch.setenv("CDS_SPARSE_SCAN32", "1")
    s32, m32 = alg.searchSparseLibrary(sl)
    monkeypatch.delenv("CDS_SPARSE_SCAN32")
    assert np.array_equal(s_auto, s32) and np.array_equal(m_auto, m32) and s_auto.max() > 0
    # (b) the reference's LM test images, replicated, against the EM mask they were chosen for
    lms = np.stack([fixtures[k] for k in ("lm_BJD", "lm_GMR", "lm_VT016795", "lm_VT033614")])
    dense.upload(0, lms)
    rl = cds.SparseLibrary(W, H, n, int((lms.max(axis=3) > 0).sum()) * (n // 4) + 4 * H * n + 64)
    for _ in range(n // 4):
        rl.append_from(dense, 0, 4)
    em = cds.PixelMatchColorDepthSearchAlgorithm(fixtures["em_12191"], 20, True, None, 0, False, 20, 0.01, 2, E.STD_RECTS)
    r_auto, rm_auto = em.searchSparseLibrary(rl)
    k, ratio = em.lastScanKernel()
    assert k == 0 and ratio > 0.05, (k, ratio)
    monkeypatch.setenv("CDS_SPARSE_SCAN16", "1")
    r16, rm16 = em.searchSparseLibrary(rl)
    assert em.lastScanKernel()[0] == 1
    assert np.array_equal(r_auto, r16) and np.array_equal(rm_auto, rm16)
    assert np.array_equal(r_auto[0, :4], r_auto[0, -4:]) and r_auto.max() > 400   # (the JUnit pairs are among them)
    # the synthetic mask over the SAME real library decides for itself (a choice belongs to a mask set, per library)
    monkeypatch.delenv("CDS_SPARSE_SCAN16")
    alg.searchSparseLibrary(rl)
    assert alg.lastScanKernel()[1] != ratio
