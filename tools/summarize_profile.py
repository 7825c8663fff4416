#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_round.sh into (1) a per-LAUNCH-CLASS text summary that by itself reproduces
the ms per launch and the counters per unit of work, and (2) profiles/scan_pmc.json, which bench.py reads for its roofline
block (keyed by the SHA-256 of the kernel sources the counters were measured on).

A launch class = the dispatches of one kernel in one profiled command whose work size is known:
  sparse_scan_headline  `bench.py --no-aux --no-cpu-baseline --verify-windows 3`: the LAST `--steps` dispatches of
                        cds::sparse_scan_kernel (the timed scans of TARGETS targets; the first `--warmup` ones are dropped)
  sparse_scan_load_only same command: the 5 timed launches of the kernel's loads-only twin (template flag LO)
  dense_scan            same command: cds::scan_kernel dispatches of STAGE targets (the verification windows)
  sparse_multi          `bench.py --no-cpu-baseline --verify-windows 3`: cds::sparse_multi_kernel dispatches (aux.mask_batch:
                        one per mask group over TARGETS targets); per (mask, target) comparison
  gradscore_list        same command: cds::gradscore_list_kernel dispatches of PAIRS pairs (aux.gradient_scores)
HBM bytes from FETCH_SIZE (KB).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read
(128-B requests tallied at 64 B) -> x 2.  tools/fetchcal (profiles/r05_fetch_size_calibration.txt, round 5) measured the other case:
FETCH_SIZE = 64 B x TCC_EA0_RDREQ whatever the request's size; the REQUEST SIZE depends on the load's cache policy, not on its shape:
scattered 4- / 8- / 12-byte loads with the non-temporal hint go out as 64-B requests (face value is right), the same loads with the default
policy -- what every gather of this repository's kernels uses -- make L2 fetch the whole 128-B line (x 2 is right, as for streams).
So a class's bytes are taken from the request-size counters when its passes hold them (TCC_EA0_RDREQ_32B / _64B / _128B: the bytes
themselves -- in round 5's profiles every product kernel shows 1.00 of its requests at 128 B), and from 2 x FETCH_SIZE otherwise
(CORRECTION[class] could name another rule: "x1" face value, "twin" stream doubled + the rest once); every class also lists the
face-value and the doubled figure.
"""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["colormipsearch_amd/csrc/cds_sparse.hip", "colormipsearch_amd/csrc/cds_scan.hip", "colormipsearch_amd/csrc/cds_shape.hip",
           "colormipsearch_amd/csrc/cds_internal.hpp"]


def trace_rows(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def pmc_rows(dirs):
    """{counter: [(dispatch order, kernel name, value)]} -- one list per counter, dispatches in launch order."""
    out = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = collections.defaultdict(dict)
            for r in csv.DictReader(open(f)):
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] = (r["Kernel_Name"], float(r["Counter_Value"]), int(r["Grid_Size"]))
            for did in sorted(per):
                for c, (k, v, g) in per[did].items():
                    out[c].append((did, k, v, g))
    return out


LOAD_ONLY = ", true>(cds::SparseScanParams)"   # the loads-only twin of the scan kernel (template flag LO): a class of its own
LOAD_ONLY16 = ", true, false>(cds::SparseScanParams)"   # ... of sparse_scan16_kernel<..., LO, PG>


def pick(seq, name, last=None, every=False):
    """dispatches of kernel `name` with the class's (= the largest) grid -- or every one of them --, optionally only the
    last `last`; `name` = "A&B" asks for both substrings, the loads-only twin is only picked when asked for by LOAD_ONLY"""
    parts = name.split("&")
    def twin(name_):   # is this dispatch a loads-only twin?  (the flag's position differs between the two kernels' template lists)
        if "sparse_scan16_kernel<" in name_:   # <..., WPE, LO, PG>
            return re.search(r", \d+, true, (true|false)>\(cds::SparseScanParams\)", name_) is not None
        return LOAD_ONLY in name_ and "sparse_scan_kernel<" in name_
    want_twin = LOAD_ONLY in parts or LOAD_ONLY16 in parts
    parts = [q for q in parts if q not in (LOAD_ONLY, LOAD_ONLY16)]
    sel = [x for x in seq if all(q in x[1] for q in parts) and twin(x[1]) == want_twin]
    if sel and not every:
        g = max(x[3] for x in sel)
        sel = [x for x in sel if x[3] == g]
    return sel[-last:] if last else sel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir", help="gpurun_out/TAG as written by tools/profile_round.sh")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--targets", type=int, default=1 << 20)
    ap.add_argument("--stage", type=int, default=8192)
    ap.add_argument("--pairs", type=int, default=0, help="pairs scored by all gradscore launches of the `full` command (0: masks x 500 + 1)")
    ap.add_argument("--multi-masks", type=int, default=1024)
    ap.add_argument("--extra-multi", type=int, default=0, help="comparisons of multi-mask launches outside aux.mask_batch")
    ap.add_argument("--write-json", default=os.path.join(ROOT, "profiles", "scan_pmc.json"))
    a = ap.parse_args()
    o = a.out_dir
    lines = []
    classes = {}
    CORRECTION = {}   # default "x2" (see the module docstring: every gather of the product kernels uses the default cache policy)

    def hbm(key, per, units, ms_total, unit_name):
        """per-unit HBM bytes of a class from its FETCH_SIZE, by the class's correction (see the module docstring)"""
        face = per["FETCH_SIZE"] * 1024.0
        how = CORRECTION.get(key, "x2")
        if {"TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum"} <= set(per):
            # the memory-side requests of L2 by size: the bytes themselves (FETCH_SIZE = 64 B x all of them, whatever their size)
            n, n128 = per["TCC_EA0_RDREQ_sum"], per["TCC_EA0_RDREQ_128B_sum"]
            n32, n64 = per.get("TCC_EA0_RDREQ_32B_sum", 0.0), per.get("TCC_EA0_RDREQ_64B_sum")
            if n64 is None:
                n64 = max(n - n128 - n32, 0.0)
            per["hbm_bytes"] = 32.0 * n32 + 64.0 * n64 + 128.0 * n128 + 64.0 * max(n - n32 - n64 - n128, 0.0)
            per["hbm_bytes_face_value"], per["hbm_bytes_doubled"], per["hbm_correction"] = face, 2.0 * face, "request sizes"
            gbs = per["hbm_bytes"] * units / (ms_total * 1e-3) / 1e9
            lines.append(f"   HBM traffic = 32 B x TCC_EA0_RDREQ_32B + 64 B x _64B + 128 B x _128B = {per['hbm_bytes']:.1f} B per {unit_name[:-1]} "
                         f"({n128 / max(n, 1e-30):.2f} of the requests are 128 B wide) = {gbs:.0f} GB/s at the traced duration = {gbs / 8000:.3f} of the 8 TB/s peak"
                         f"   [FETCH_SIZE at face value {face:.1f} B = {gbs * face / per['hbm_bytes'] / 8000:.3f}; doubled {2 * face:.1f} B = {gbs * 2 * face / per['hbm_bytes'] / 8000:.3f}]")
            return
        twin = classes.get("sparse_scan_load_only", {}).get("per_target", {}).get("FETCH_SIZE")
        if how == "twin" and twin is not None:
            per["hbm_bytes"] = face + min(face, twin * 1024.0)     # the stream's requests are 128 B wide (counted half), the gathers' 64 B
            note = f"FETCH_SIZE + the loads-only twin's FETCH_SIZE (the stream's 128-B requests count half; the rounds' gathers are 64-B requests at face value)"
        elif how == "x1":
            per["hbm_bytes"] = face
            note = "FETCH_SIZE at face value (scattered 12-byte groups: 64-B requests, tools/fetchcal)"
        else:
            per["hbm_bytes"] = 2.0 * face
            note = "2 x FETCH_SIZE (coalesced streaming loads: 128-B requests tallied at 64 B)"
        per["hbm_bytes_face_value"], per["hbm_bytes_doubled"], per["hbm_correction"] = face, 2.0 * face, how
        gbs = per["hbm_bytes"] * units / (ms_total * 1e-3) / 1e9
        lines.append(f"   HBM traffic = {note} = {per['hbm_bytes']:.1f} B per {unit_name[:-1]} = {gbs:.0f} GB/s at the traced duration = {gbs / 8000:.3f} of the 8 TB/s peak"
                     f"   [face value {face:.1f} B = {gbs * face / per['hbm_bytes'] / 8000:.3f}; doubled {2 * face:.1f} B = {gbs * 2 * face / per['hbm_bytes'] / 8000:.3f}]")

    def add_class(key, kernel, cmd_tag, unit_name, units_per_dispatch, last=None, desc="", units_total=None):
        """units_total: the class is ALL dispatches of the kernel in the command, which together process that many units
        (launches of different sizes: per-unit figures = sums over the dispatches / units_total)."""
        tr = [(i, r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, int(r["Grid_Size_X"]))
              for i, r in enumerate(trace_rows(os.path.join(o, "trace_" + cmd_tag)))]
        sel = pick(tr, kernel, last, every=units_total is not None)
        if not sel:
            return
        if units_total is not None:
            tot_ms = sum(x[2] for x in sel)
            cls = {"kernel": sel[0][1][:120], "command": cmd_tag, "dispatches": len(sel), unit_name + "_total": units_total, "total_ms": tot_ms,
                   "per_" + unit_name[:-1]: {}}
            lines.append(f"== class {key}: {desc}")
            lines.append(f"   kernel {cls['kernel']}")
            lines.append(f"   rocprofv3 --kernel-trace: {len(sel)} dispatches (all sizes), {tot_ms:.3f} ms together for {units_total} {unit_name} -> {units_total / tot_ms / 1e3:.2f} M {unit_name}/s")
            pm = pmc_rows(glob.glob(os.path.join(o, "pmc_" + cmd_tag + "_*")))
            per = cls["per_" + unit_name[:-1]]
            for c in sorted(pm):
                sq = pick(pm[c], kernel, last, every=True)
                if not sq:
                    continue
                tot = sum(x[2] for x in sq)
                per[c] = tot / units_total
                lines.append(f"   {c:24s} sum over {len(sq):5d} dispatches {tot:20.1f}   per {unit_name[:-1]} {tot / units_total:12.4f}")
            if "FETCH_SIZE" in per:
                hbm(key, per, units_total, tot_ms, unit_name)
            if "TCC_HIT_sum" in per and "TCC_MISS_sum" in per:
                lines.append(f"   L2: {per['TCC_HIT_sum'] / max(per['TCC_HIT_sum'] + per['TCC_MISS_sum'], 1e-30):.3f} of the requests hit")
            classes[key] = cls
            return
        ms = [x[2] for x in sel]
        cls = {"kernel": sel[0][1][:120], "command": cmd_tag, "dispatches": len(sel), unit_name + "_per_dispatch": units_per_dispatch,
               "mean_ms": sum(ms) / len(ms), "min_ms": min(ms), "max_ms": max(ms), "grid_size_x": sel[0][3], "per_" + unit_name[:-1]: {}}
        lines.append(f"== class {key}: {desc}")
        lines.append(f"   kernel {cls['kernel']}")
        lines.append(f"   rocprofv3 --kernel-trace: {len(sel)} dispatches, grid {sel[0][3]}, mean {cls['mean_ms']:.4f} ms (min {min(ms):.4f}, max {max(ms):.4f}); "
                     f"{units_per_dispatch} {unit_name} per dispatch -> {units_per_dispatch / cls['mean_ms'] / 1e3:.2f} M {unit_name}/s")
        pm = pmc_rows(glob.glob(os.path.join(o, "pmc_" + cmd_tag + "_*")))
        for c in sorted(pm):
            s = pick(pm[c], kernel, last)
            if not s:
                continue
            mean = sum(x[2] for x in s) / len(s)
            cls["per_" + unit_name[:-1]][c] = mean / units_per_dispatch
            lines.append(f"   {c:24s} mean per dispatch {mean:18.1f}  (n={len(s)})   per {unit_name[:-1]} {mean / units_per_dispatch:12.3f}")
        per = cls["per_" + unit_name[:-1]]
        if "FETCH_SIZE" in per:
            hbm(key, per, units_per_dispatch, cls["mean_ms"], unit_name)
        if "GRBM_GUI_ACTIVE" in per and "SQ_ACTIVE_INST_VALU" in per:   # busy shares of the two pipes this round argues with (256 CUs, 8 XCDs)
            cyc_cu = per["GRBM_GUI_ACTIVE"] / 8.0 * 256.0          # cycles one CU spends per unit (every CU works on its own units)
            s_ = f"   per {unit_name[:-1]} and CU: {cyc_cu:.0f} cycles; vector pipes active {per['SQ_ACTIVE_INST_VALU'] * 4.0 / (4.0 * cyc_cu):.2f}"
            if "SQ_LDS_IDX_ACTIVE" in per:
                s_ += f", LDS active {per['SQ_LDS_IDX_ACTIVE'] / cyc_cu:.2f} (bank conflicts {per.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(per['SQ_LDS_IDX_ACTIVE'], 1e-30):.2f} of that)"
            lines.append(s_)
        classes[key] = cls

    # the headline kernel: the 16-bit-stream scan when the run's pilot chose it (round 4), else the 4-byte-record scan
    add_class("sparse_scan_load_only", "sparse_scan16_kernel&" + LOAD_ONLY16, "headline", "targets", a.targets, last=5,
              desc="the loads-only twin of the stream kernel (bench.py roofline.load_only): same grabs, headers, stream loads, no examination")
    add_class("sparse_scan_headline", "sparse_scan16_kernel", "headline", "targets", a.targets, last=a.steps,
              desc=f"the {a.steps} timed 1-mask scans of the resident library (bench.py `value`): cds::sparse_scan16_kernel over the 16-bit stage-1 stream")
    if "sparse_scan_headline" not in classes:
        classes.pop("sparse_scan_load_only", None)
        add_class("sparse_scan_headline", "sparse_scan_kernel", "headline", "targets", a.targets, last=a.steps,
                  desc=f"the {a.steps} timed 1-mask scans of the resident library (bench.py `value`)")
        add_class("sparse_scan_load_only", "sparse_scan_kernel&" + LOAD_ONLY, "headline", "targets", a.targets, last=5,
                  desc="the loads-only twin of the scan kernel (bench.py roofline.load_only): same grabs, headers, record loads, no examination")
    add_class("sparse_scan_records32", "sparse_scan_kernel", "headline32", "targets", a.targets, last=a.steps,
              desc=f"the same {a.steps} scans with CDS_SPARSE_SCAN32=1: cds::sparse_scan_kernel over the 4-byte records (round 3's headline kernel)")
    add_class("dense_scan", "cds::scan_kernel", "headline", "targets", a.stage, desc="the dense-layout kernel over the verification windows")
    # the overlapping regime: every dispatch of the 4-byte-record kernel in the two commands (1 warm-up + 3 timed scans of 32,768 targets;
    # the overlapping-synthetic tool: pilot rest + 3 timed + 4 forced + 1 final scans of 131,072 targets)
    add_class("real_images_em_12191", "sparse_scan_kernel", "real", "targets", 32768,
              desc="tools/real_image_rate.py --mask em_12191: the reference's 4 LM test MIPs x 8,192 copies against the EM mask that overlaps them (cds::sparse_scan_kernel, the pilot's choice)")
    add_class("overlapping_synthetic", "sparse_scan_kernel", "overlap", "targets", 131072,
              desc="tools/overlap_rate.py (= bench.py aux.overlapping_synthetic at 131,072 targets): synthetic targets that overlap the benchmark mask, 38 % of the visited records reach a round (cds::sparse_scan_kernel)")
    add_class("sparse_multi", "sparse_multi_kernel", "multi", "comparisons", 8 * 262144,
              desc="tools/bench_sparse.py --masks 64 --targets 262144: every launch of cds::sparse_multi_kernel (one group of 8 masks over 262,144 targets each) -- the "
                   "kernel of BASELINE configs[2] at a size whose counter passes fit one gpurun call; per (mask, target) comparison")
    add_class("sparse_multi_full" if "sparse_multi" in classes else "sparse_multi", "sparse_multi_kernel", "full", "comparisons", None, units_total=a.multi_masks * a.targets + a.extra_multi,
              desc=f"aux.mask_batch + the scans aux.gradient_scores' selection needs: every multi-mask launch of the command ({a.multi_masks} masks x {a.targets} targets)")
    add_class("gradscore_list", "gradscore_list_kernel<true>", "full", "pairs", None, units_total=a.pairs or (a.multi_masks * 500 + 1),
              desc="aux.gradient_scores, the default path at 2.2 pairs per target: the scorer reads the zgap bytes itself, no conversion pass (one launch per batch of selected targets) + the one fixture pair")
    add_class("gradscore_list_zslice_plane", "gradscore_list_kernel<false>", "full", "pairs", None, units_total=a.pairs or (a.multi_masks * 500),
              desc="the same pairs with CDS_SHAPE_ZSLICE=1 (round 3's path): S(z) from the converted zslice plane")
    add_class("zslice_conversion", "zslice_kernel", "full", "targets", None, units_total=230693,
              desc="... and the conversion pass that path needs first: one zgap image -> zslice plane per selected target (230,693 distinct targets)")
    for tagc in ("headline", "full"):
        for f in glob.glob(os.path.join(o, "trace_" + tagc, "**", "*kernel_stats.csv"), recursive=True):
            lines.append(f"== rocprofv3 --kernel-trace --stats (kernel_stats.csv), command `{tagc}` -- ALL dispatches of the run, mixed sizes")
            lines.append(open(f).read().strip())
    for name in ("bench_headline.json", "bench_full.json"):
        p = os.path.join(o, name)
        if os.path.exists(p):
            lines.append(f"== {name}")
            lines.append(open(p).read().strip())
    text = "\n".join(lines) + "\n"
    open(os.path.join(o, "summary.txt"), "w").write(text)
    print(text)
    if a.write_json and classes:
        j = {"tag": a.tag, "command": "python3 bench.py --no-aux --no-cpu-baseline --verify-windows 3 (class *_headline) / --no-cpu-baseline --verify-windows 3 (others)",
             "hbm_bytes": "per class: from the request-size counters (32 B x TCC_EA0_RDREQ_32B + 64 B x _64B + 128 B x _128B) when the class has them, else "
                          "2 x FETCH_SIZE x 1024 (FETCH_SIZE counts 64 B per request; default-policy loads, streams and gathers alike, are 128-B requests: "
                          "tools/fetchcal, profiles/r05_fetch_size_calibration.txt); hbm_bytes_face_value / _doubled beside it",
             "sources": {s: hashlib.sha256(open(os.path.join(ROOT, s), "rb").read()).hexdigest() for s in SOURCES},
             "classes": classes}
        json.dump(j, open(a.write_json, "w"), indent=1)
        print("wrote", a.write_json, file=sys.stderr)


if __name__ == "__main__":
    main()
