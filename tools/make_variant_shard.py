#!/usr/bin/env python3
"""Write a variant shard (cds.VariantShardWriter: every target's MIP, GradientImage and ZGapImage, lossless, decoded on the GPU by
Library.load_variants / processor.calculateGradientScoresFromShardTable) from a library's image files.  No GPU needed.

    python tools/make_variant_shard.py OUT.cdsv --targets DIR_OR_LIST [--gradients DIR_OR_LIST] [--zgaps DIR_OR_LIST] [--threads N]
    python tools/make_variant_shard.py OUT.cdsv --mips-json MIPS.json        # a JSON array of the reference's neuron entities: their
                                                                             # computeFiles (InputColorDepthImage, GradientImage, ZGapImage;
                                                                             # files or zip entries) in the array's order

DIR_OR_LIST is a directory (its *.tif / *.tiff / *.png files, sorted by name) or a text file with one path per line.  Variants are
matched to targets by file stem up to the first of the usual variant suffixes, the way the reference finds a MIP's variant in a
variant directory (same name, other directory / extension); a target without a match is stored without that variant.  Target t of
the shard is line / file t of --targets: keep that order for the match tables' target indices."""
import argparse, os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colormipsearch_amd import cds, mipio

EXT = (".tif", ".tiff", ".png")


def listing(arg):
    if arg is None:
        return None
    if os.path.isdir(arg):
        return sorted(os.path.join(arg, f) for f in os.listdir(arg) if f.lower().endswith(EXT))
    return [ln.strip() for ln in open(arg) if ln.strip()]


def stem(path):
    s = os.path.splitext(os.path.basename(path))[0]
    return re.sub(r"(_gradient|_grad|_zgap|_rgb20px|-gradient|-zgap)$", "", s, flags=re.I)


def matched(targets, variants):
    if variants is None:
        return None
    by_stem = {stem(v): v for v in variants}
    return [by_stem.get(stem(t)) for t in targets]


ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--targets")
ap.add_argument("--mips-json")
ap.add_argument("--gradients")
ap.add_argument("--zgaps")
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--batch", type=int, default=4096)
a = ap.parse_args()
if a.mips_json:
    import json
    from colormipsearch_amd import results
    targets, grads, zgaps = results.variant_files(json.load(open(a.mips_json)))
else:
    targets = listing(a.targets) if a.targets else None
    if not targets:
        sys.exit("no target files (--targets or --mips-json)")
    grads, zgaps = matched(targets, listing(a.gradients)), matched(targets, listing(a.zgaps))
H, W = mipio.read_mip(targets[0]).shape[:2]     # the library's image size = its first target's
t0 = time.perf_counter()
with cds.VariantShardWriter(a.out, W, H) as w:
    for first in range(0, len(targets), a.batch):
        sl = slice(first, first + a.batch)
        w.append_files(targets[sl], None if grads is None else grads[sl], None if zgaps is None else zgaps[sl], threads=a.threads)
        print(f"{min(first + a.batch, len(targets))} / {len(targets)} targets", file=sys.stderr)
t = time.perf_counter() - t0
sh = cds.VariantShard(a.out)
print(f"{len(sh)} targets of {W} x {H} in {sh.file_bytes / 1e6:.1f} MB ({sh.file_bytes / max(1, len(sh)) / 1e3:.0f} KB per target), "
      f"{int((sh.flags & sh.HAS_GRADIENT != 0).sum())} with a gradient, {int((sh.flags & sh.HAS_ZGAP != 0).sum())} with a zgap image; {len(sh) / t:.0f} targets/s")
