#!/usr/bin/env python3
"""Kernel 2 on a library that exists as FILES (processor.calculateGradientScoresFromFilesTable: BASELINE configs[4] on a real library,
VERDICT r4 weak #10): `--targets` distinct targets, each a PackBits TIFF + a 16-bit gradient PNG + a PackBits zgap TIFF written from the
synthetic generator's variants, `--pairs-per-target` pairs each against `--masks` masks; reports distinct targets per second through
decode + upload + scoring, and the scoring alone.    python tools/gradscore_files_rate.py [--targets 1024]"""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from PIL import Image
from colormipsearch_amd import cds
from colormipsearch_amd import processor as P
W, H = 1210, 566


def drop_cache(paths):
    """ask the kernel to forget the pages of these files (works on a disk-backed file system; a tmpfs keeps them)"""
    for q in set(os.path.realpath(x) for x in paths):
        fd = os.open(q, os.O_RDONLY)
        try:
            os.fsync(fd)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        finally:
            os.close(fd)


ap = argparse.ArgumentParser()
ap.add_argument("--targets", type=int, default=1024)
ap.add_argument("--masks", type=int, default=16)
ap.add_argument("--pairs-per-target", type=float, default=2.2)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--shard", action="store_true", help="also: the same targets through a variant shard written from the files (decoded on the GPU)")
ap.add_argument("--cold", action="store_true", help="drop the files' / the shard's pages from the page cache before the timed run (posix_fadvise DONTNEED)")
ap.add_argument("--real", action="store_true", help="the reference's three LM test targets with their gradient / zgap images instead of synthetic variants")
a = ap.parse_args()
n_files = min(a.targets, 256)
lib = cds.Library(W, H, n_files)
if a.real:   # (tests/golden: the images of the reference's own JUnit cases; the zgap of the two targets without a ZGapImage file is made on the device)
    fx = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cds_fixtures.npz"))
    n_files = 3
    lib = cds.Library(W, H, n_files)
    lib.upload(0, np.stack([fx["lm_BJD"], fx["lm_VT016795"], fx["lm_VT033614"]]))
    lib.upload_variants(0, gradients=np.stack([fx["grad_BJD"], fx["grad_VT016795"], fx["grad_VT033614"]]))
    lib.make_zgap_variants(0, 3, 20, 10, cds.label_regions)
else:
    lib.generate_synthetic(0xC0FFEE, 0, 0, n_files, kind=0, with_variants=True)
imgs = lib.download(0, n_files)
grads, zgaps = lib.download_variants(0, n_files)
mlib = cds.Library(W, H, a.masks)
mlib.generate_synthetic(0xBADC0DE, 0, 0, a.masks, kind=2)
masks = mlib.download(0, a.masks)
with tempfile.TemporaryDirectory() as d:
    tf, gf, zf = [], [], []
    for i in range(a.targets):
        p = [os.path.join(d, f"t{i}.tif"), os.path.join(d, f"g{i}.png"), os.path.join(d, f"z{i}.tif")]
        if i < n_files:
            Image.fromarray(imgs[i], "RGB").save(p[0], compression="packbits")
            Image.fromarray(grads[i]).save(p[1])
            Image.fromarray(zgaps[i], "RGB").save(p[2], compression="packbits")
        else:
            for q, src in zip(p, (tf, gf, zf)):
                os.link(src[i % n_files], q)
        tf.append(p[0]); gf.append(p[1]); zf.append(p[2])
    rng = np.random.default_rng(1)
    n_pairs = int(a.targets * a.pairs_per_target)
    t_idx = np.concatenate([np.arange(a.targets), rng.integers(0, a.targets, n_pairs - a.targets)]).astype(np.int64)
    m_idx = rng.integers(0, a.masks, n_pairs).astype(np.int32)
    order = np.lexsort((t_idx, m_idx))
    table = P.MatchTable(m_idx[order], t_idx[order], np.full(n_pairs, 100, np.int32), np.full(n_pairs, 0.01, np.float32), np.zeros(n_pairs, bool))
    provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 1, 2, cds.label_regions)
    proc = P.GpuColorMIPSearchProcessor(cds.ColorMIPSearch(0.01, 20, provider), lib, None)
    proc.calculateGradientScoresFromFilesTable(masks, table.take(np.arange(min(n_pairs, 64))), tf, gf, zf, 20, batch=a.batch, threads=a.threads)   # warm-up
    if a.cold:
        drop_cache(tf[:n_files] + gf[:n_files] + zf[:n_files])
    t0 = time.perf_counter()
    out = proc.calculateGradientScoresFromFilesTable(masks, table, tf, gf, zf, 20, batch=a.batch, threads=a.threads)
    t = time.perf_counter() - t0
    # the same pairs scored from the variants as they sit in memory (file i holds the variants of resident target i % n_files)
    ref = proc.calculateGradientScoresTable(masks, P.MatchTable(table.maskIndex, table.targetIndex % n_files, table.matchingPixels, table.matchingPixelsRatio, table.mirrored), 20)
    same = bool((ref.gradientAreaGap == out.gradientAreaGap).all() and (ref.highExpressionArea == out.highExpressionArea).all())
    bytes_files = sum(os.path.getsize(p) for p in tf[:n_files] + gf[:n_files] + zf[:n_files]) / n_files
    print(json.dumps({"distinct_targets": a.targets, "pairs": n_pairs, "masks": a.masks, "threads": a.threads, "batch": a.batch, "seconds": round(t, 3),
                      "targets_per_s": round(a.targets / t), "files_per_s": round(3 * a.targets / t), "pairs_per_s": round(n_pairs / t),
                      "file_bytes_per_target": round(bytes_files), "scored": int((out.gradientAreaGap >= 0).sum()), "equal_to_in_memory_scores": same, "path": proc.last_from_files}))
    if a.shard:
        sp = os.path.join(d, "variants.cdsv")
        t0 = time.perf_counter()
        with cds.VariantShardWriter(sp, W, H) as w:
            w.append_files(tf, gf, zf, threads=a.threads)
        t_write = time.perf_counter() - t0
        shard = cds.VariantShard(sp)
        fresh = lambda: P.MatchTable(table.maskIndex, table.targetIndex, table.matchingPixels, table.matchingPixelsRatio, table.mirrored)
        proc.calculateGradientScoresFromShardTable(masks, fresh().take(np.arange(min(n_pairs, 64))), shard, 20, batch=a.batch, threads=a.threads)   # warm-up
        if a.cold:
            drop_cache([sp])
        t0 = time.perf_counter()
        out2 = proc.calculateGradientScoresFromShardTable(masks, fresh(), shard, 20, batch=a.batch, threads=a.threads)
        t2 = time.perf_counter() - t0
        same2 = bool((ref.gradientAreaGap == out2.gradientAreaGap).all() and (ref.highExpressionArea == out2.highExpressionArea).all())
        print(json.dumps({"variant_shard": True, "distinct_targets": a.targets, "pairs": n_pairs, "threads": a.threads, "batch": a.batch, "seconds": round(t2, 3),
                          "targets_per_s": round(a.targets / t2), "shard_bytes_per_target": round(shard.file_bytes / a.targets),
                          "shard_read_GB_per_s": round(shard.file_bytes / t2 / 1e9, 2), "gradient_codes": {"second_differences": int((shard.flags & 8 != 0).sum()), "two_bits_per_pixel": int((shard.flags & 2 != 0).sum())},
                          "cold": bool(a.cold), "file_system": os.popen("df -T %s | tail -1" % d).read().split()[1:2], "write_seconds": round(t_write, 2), "write_targets_per_s": round(a.targets / t_write),
                          "equal_to_in_memory_scores": same2, "path": proc.last_from_files}))
