"""Parity at the image size of the VNC alignment spaces (573 x 1119: taller than the 1,000 rows the library accepted until round 5, narrower than
any test image so far): every kernel against the oracle -- the sparse scan over both record streams, the multi-mask kernel, the dense scan
(single-mask layout, batch layout, and a mask spanning more than 1,023 rows, which the single-mask records cannot hold), the shape scorer with
device-made zgap images, the variant shard.  The reference accepts any image size (PixelMatchColorDepthSearchAlgorithm.java:171-175 only compares
the target's with the query's); this pins the sizes its libraries actually come in.  Bar: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 573, 1119


@pytest.fixture(scope="module")
def cds():
    from colormipsearch_amd import cds as m
    m.default_context()  # fails loudly without a gfx950 GPU
    return m


def _neuron(rng, lut, w, h, strokes, tall=False):
    """a few thick polylines whose colour (depth) drifts along their length, on black: the shape of a MIP"""
    img = np.zeros((h, w, 3), np.uint8)
    for _ in range(strokes):
        x, y = float(rng.integers(20, w - 20)), float(rng.integers(2, 40) if tall else rng.integers(20, h - 20))
        z, br = float(rng.integers(0, 256)), float(rng.uniform(0.35, 1.0))
        steps = h + 200 if tall else int(rng.integers(60, 500))
        dx, dy = (rng.uniform(-0.3, 0.3), 1.0) if tall else (rng.uniform(-1, 1), rng.uniform(-1, 1))
        for _ in range(steps):
            x, y = min(max(x + dx + rng.uniform(-0.6, 0.6), 3), w - 4), min(max(y + dy + rng.uniform(-0.6, 0.6), 1), h - 2)
            z = min(max(z + rng.uniform(-1.5, 1.5), 0), 255)
            r = int(rng.integers(1, 4))
            x0, y0 = int(x), int(y)
            img[max(0, y0 - r):y0 + r + 1, max(0, x0 - r):x0 + r + 1] = (lut[int(z)] * br).astype(np.uint8)
    return img


@pytest.fixture(scope="module")
def images(oracle):
    rng = np.random.default_rng(573 * 1119)
    lut = oracle.depth_lut()
    masks = np.stack([_neuron(rng, lut, W, H, 3) for _ in range(8)] + [_neuron(rng, lut, W, H, 2, tall=True)])
    targets = [_neuron(rng, lut, W, H, int(rng.integers(3, 9))) for _ in range(5)]
    targets += [masks[0], np.fliplr(masks[1]), np.roll(masks[2], 2, axis=0), masks[8], np.zeros_like(masks[0])]
    return masks, np.stack(targets)


def _sparse(cds, images):
    sl = cds.SparseLibrary(W, H, len(images), int((images.max(axis=3) > 0).sum()) + 16 * len(images) * H + 64)
    sl.append(images)
    return sl


@pytest.mark.parametrize("kernel", ["stream16", "records32"])
def test_sparse_scan_counts_equal_oracle(cds, oracle, images, kernel, monkeypatch):
    masks, targets = images
    monkeypatch.delenv("CDS_SPARSE_SCAN32", raising=False)
    monkeypatch.delenv("CDS_SPARSE_SCAN16", raising=False)
    monkeypatch.setenv("CDS_SPARSE_SCAN32" if kernel == "records32" else "CDS_SPARSE_SCAN16", "1")
    rects = cds.label_regions(W)
    sl = _sparse(cds, targets)
    for m in (0, 3, 8):                                                       # (8: the mask that spans the whole height)
        alg = cds.PixelMatchColorDepthSearchAlgorithm(masks[m], 20, True, None, 0, False, 20, 0.02, 2, rects)
        counts = alg.searchSparseLibraryCounts(sl)
        scores, mir = alg.searchSparseLibrary(sl)
        q = oracle.PixelMatchQuery(masks[m], 20, 20, 0.02, 2, True, rects)
        for t in range(len(targets)):
            s, mi, c = q.score(targets[t])
            assert np.array_equal(counts[0, t], c), (m, t)
            assert scores[0, t] == s and bool(mir[0, t]) == mi, (m, t)
        assert m == 3 or scores[0].max() > 1000                               # (masks 0 and 8 are among the targets: they find themselves)


def test_multi_mask_kernel_and_match_list(cds, oracle, images):
    masks, targets = images
    rects = cds.label_regions(W)
    sl = _sparse(cds, targets)
    alg = cds.PixelMatchColorDepthSearchAlgorithm(masks, 20, True, None, 0, False, 20, 0.02, 2, rects)
    scores, mir = alg.searchSparseLibrary(sl)
    for m in range(len(masks)):
        q = oracle.PixelMatchQuery(masks[m], 20, 20, 0.02, 2, True, rects)
        for t in range(len(targets)):
            s, mi, _ = q.score(targets[t])
            assert scores[m, t] == s and bool(mir[m, t]) == mi, (m, t)


def test_dense_scan_single_batch_and_a_mask_taller_than_1023_rows(cds, oracle, images):
    masks, targets = images
    rects = cds.label_regions(W)
    lib = cds.Library(W, H, len(targets))
    lib.upload(0, targets)
    ys = np.flatnonzero(masks[8].max(axis=(1, 2)) > 20)
    assert ys[-1] - ys[0] > 1023                                              # the single-mask records' 10-bit row field cannot hold this one
    for sel in ([0], [8], list(range(9))):                                    # single-mask layout; the tall mask (batch layout); a mask batch
        alg = cds.PixelMatchColorDepthSearchAlgorithm(masks[sel], 20, True, None, 0, False, 20, 0.02, 2, rects)
        scores, mir = alg.searchLibrary(lib)
        for k, m in enumerate(sel):
            q = oracle.PixelMatchQuery(masks[m], 20, 20, 0.02, 2, True, rects)
            for t in range(len(targets)):
                s, mi, _ = q.score(targets[t])
                assert scores[k, t] == s and bool(mir[k, t]) == mi, (sel, m, t)


def test_shape_scores_with_device_made_zgap(cds, oracle, images):
    masks, targets = images
    rects = cds.label_regions(W)
    rng = np.random.default_rng(7)
    n_t = 4
    lib = cds.Library(W, H, n_t)
    lib.upload(0, targets[:n_t])
    steps = rng.integers(-1, 2, (n_t, H, W))
    steps[:, :, 0] = 0
    grads = np.clip(rng.integers(0, 600, (n_t, H, 1)) + np.cumsum(steps, axis=2), 0, 65535).astype(np.uint16)
    lib.upload_variants(0, gradients=grads)
    lib.make_zgap_variants(0, n_t, 20, 10, rects)
    _, zgaps = lib.download_variants(0, n_t)
    for t in range(2):
        assert np.array_equal(zgaps[t], oracle.make_zgap(targets[t], 20, 10, rects))
    sel = [0, 8]
    alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks[sel], 20, True, None, rects)
    pm = np.repeat(np.arange(len(sel)), n_t).astype(np.int32)
    ps = np.tile(np.arange(n_t), len(sel)).astype(np.int64)
    gap, high, mir = alg.scoreLibraryPairs(lib, pm, ps)
    for k, m in enumerate(sel):
        sq = oracle.ShapeQuery(masks[m], 20, True, rects, None)
        for t in range(n_t):
            want = sq.score(targets[t], grads[t], zgaps[t])
            i = k * n_t + t
            assert (int(gap[i]), int(high[i]), bool(mir[i])) == (want[0], want[1], want[3]), (m, t)


def test_variant_shard_and_shard_ingest(cds, images, tmp_path):
    masks, targets = images
    rng = np.random.default_rng(9)
    grads = (np.abs(np.arange(W)[None, :] - 280) + np.abs(np.arange(H)[:, None] - 500)).astype(np.uint16)[None].repeat(3, axis=0)
    zg = np.stack([np.roll(targets[i], 3, axis=0) for i in range(3)])
    path = tmp_path / "vnc.cdsv"
    with cds.VariantShardWriter(path, W, H) as w:
        w.append(targets[:3], grads, zg)
        w.append(targets[3:])
    shard = cds.VariantShard(path)
    assert (shard.width, shard.height, len(shard)) == (W, H, len(targets))
    lib = cds.Library(W, H, len(targets))
    order = rng.permutation(len(targets))
    lib.load_variants(0, shard, order)
    imgs = lib.download(0, len(targets))
    g, z = lib.download_variants(0, len(targets))
    for k, t in enumerate(order):
        assert np.array_equal(imgs[k], targets[t]), t
        if t < 3:
            assert np.array_equal(g[k], grads[t]) and np.array_equal(z[k], zg[t]), t
    a, b = cds.SparseLibrary(W, H), _sparse(cds, targets)
    a.append_from_shard(shard, batch=4)
    assert len(a) == len(b) and a.pixels == b.pixels
    for i in range(len(targets)):
        (ra, ca), (rb, cb) = a.image_records(i), b.image_records(i)
        assert np.array_equal(ra, rb) and np.array_equal(ca, cb), i


def test_sizes_beyond_the_limits_are_refused(cds):
    for w, h in ((2001, 100), (100, 1537), (2000, 1001), (4, 4)):            # too wide, too tall, too many pixels, too small
        with pytest.raises(cds.CdsError, match="image size"):
            cds.Library(w, h, 1)
    cds.Library(2000, 1000, 1)
    cds.Library(1302, 1536, 1)
