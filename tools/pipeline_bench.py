#!/usr/bin/env python3
"""colorDepthSearch -> top hits -> gradientScores end to end on one GPU (BASELINE configs[2] + configs[4] in miniature),
through the columnar processor pipeline.  Timing only (parity: tests/test_processor.py)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from colormipsearch_amd import cds, processor as P
W, H = 1210, 566
ap = argparse.ArgumentParser()
ap.add_argument("--masks", type=int, default=128)
ap.add_argument("--targets", type=int, default=16384)
ap.add_argument("--top", type=int, default=500)
a = ap.parse_args()
lib = cds.Library(W, H, a.targets)
t0 = time.perf_counter(); lib.generate_synthetic(0xC0FFEE, 0, 0, a.targets, kind=0, with_variants=True); gen = time.perf_counter() - t0
mlib = cds.Library(W, H, a.masks); mlib.generate_synthetic(0xBADC0DE, 0, 0, a.masks, kind=2); masks = mlib.download(0, a.masks)
infos = [P.NeuronInfo(str(i), f"line{i % 4099}", f"sample{i}") for i in range(a.targets)]
provider = cds.ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider(True, 20, 1, 2, cds.label_regions)
proc = P.GpuColorMIPSearchProcessor(cds.ColorMIPSearch(1.0, 20, provider), lib, infos)
t0 = time.perf_counter(); table = proc.findAllColorDepthMatchesTable(masks); t_search = time.perf_counter() - t0
t0 = time.perf_counter(); best = proc.selectBestMatchesTable(table, a.top, 1, 1); t_sel = time.perf_counter() - t0
t0 = time.perf_counter(); scored = proc.calculateGradientScoresTable(masks, best, 20, True, None, cds.label_regions); t_grad = time.perf_counter() - t0
print(json.dumps({"masks": a.masks, "targets": a.targets, "library_and_variants_generation_s": round(gen, 2),
                  "matches": len(table), "search_s": round(t_search, 3), "selected": len(best), "select_s": round(t_sel, 3),
                  "gradient_scores_s": round(t_grad, 3), "pairs_per_s_search": round(a.masks * a.targets / t_search),
                  "pairs_per_s_gradscore": round(len(best) / max(t_grad, 1e-9))}))
