#!/usr/bin/env python3
"""Library ingest from image files: `--files` synthetic targets are written as TIFFs (`--compression`), then appended to a
sparse resident library with cds_sparse_library_append_files (native threaded decode || upload + conversion).  Reports
images/s for the native path and, on a sample, for the Pillow decode it replaces."""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from PIL import Image
from colormipsearch_amd import cds, mipio
W, H = 1210, 566
ap = argparse.ArgumentParser()
ap.add_argument("--files", type=int, default=2048)
ap.add_argument("--compression", default="packbits")
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--variant-shard", action="store_true", help="also: the same target files packed into a variant shard (no gradient / zgap) and loaded into dense slots through the GPU decoders")
a = ap.parse_args()
lib = cds.Library(W, H, 256)
lib.generate_synthetic(0xC0FFEE, 0, 0, 256, kind=0)
imgs = lib.download(0, 256)
with tempfile.TemporaryDirectory() as d:
    paths, nbytes = [], 0
    for i in range(a.files):
        p = os.path.join(d, f"t{i}.tif")
        if i < 256:
            Image.fromarray(imgs[i], "RGB").save(p, compression=a.compression)
        else:
            os.link(paths[i % 256], p)      # the same 256 files again (page-cached either way)
        paths.append(p)
        nbytes += os.path.getsize(p)
    sl = cds.SparseLibrary(W, H, a.files, 46000 * a.files)
    sl.append_files(paths[:64], threads=a.threads)   # warm-up (staging slots, kernels)
    sl.clear()
    t0 = time.perf_counter(); sl.append_files(paths, threads=a.threads); t = time.perf_counter() - t0
    t0 = time.perf_counter()
    for p in paths[:64]:
        mipio.read_mip(p, native=False)
    tp = (time.perf_counter() - t0) / 64
    extra = {}
    if a.variant_shard:   # kernel 2's loader on the same files: write once, then decode on the device (tools/gradscore_files_rate.py --shard adds the variants)
        sp = os.path.join(d, "targets.cdsv")
        t0 = time.perf_counter()
        with cds.VariantShardWriter(sp, W, H) as w:
            w.append_files(paths, threads=a.threads)
        tw = time.perf_counter() - t0
        shard = cds.VariantShard(sp)
        dl = cds.Library(W, H, min(a.files, 1024))
        dl.load_variants(0, shard, np.arange(min(a.files, 64)), threads=a.threads)          # warm-up
        t0 = time.perf_counter()
        for first in range(0, a.files, dl.capacity):
            dl.load_variants(0, shard, np.arange(first, min(first + dl.capacity, a.files)), threads=a.threads)
        tl = time.perf_counter() - t0
        same = bool(np.array_equal(dl.download(0, 4), np.stack([mipio.read_mip(q) for q in paths[(a.files - 1) // dl.capacity * dl.capacity:][:4]])))
        sl2 = cds.SparseLibrary(W, H, a.files, 46000 * a.files)
        t0 = time.perf_counter(); sl2.append_from_shard(shard, threads=a.threads); ts = time.perf_counter() - t0
        extra = {"variant_shard": {"sparse_append_images_per_s": round(a.files / ts), "sparse_records_equal": bool(sl2.pixels == sl.pixels),"write_images_per_s": round(a.files / tw), "load_images_per_s": round(a.files / tl), "bytes_per_image": round(shard.file_bytes / a.files),
                                   "read_GBps": round(shard.file_bytes / tl / 1e9, 2), "last_batch_equals_the_files": same}}
    print(json.dumps({**extra, "files": a.files, "compression": a.compression, "mean_file_bytes": round(nbytes / a.files), "threads": a.threads or os.cpu_count(),
                      "append_files_s": round(t, 3), "images_per_s": round(a.files / t), "decoded_GBps": round(a.files * 3 * W * H / t / 1e9, 2),
                      "pillow_decode_ms_per_image_single_thread": round(tp * 1e3, 2), "records": sl.pixels}))
