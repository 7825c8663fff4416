#!/bin/bash
# rocprofv3 evidence for the batched-mask scan (BASELINE configs[2] shape): kernel-trace stats + separate PMC passes.
set -u
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}_batch
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/tools/bench_extra.py --masks 128 --targets 16384"
python3 $ARGS > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
lines = []
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    lines.append("== rocprofv3 --kernel-trace --stats of: python3 tools/bench_extra.py --masks 128 --targets 16384")
    lines.append(open(f).read().strip())
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if "scan_kernel" not in k:
        continue
    lines.append("== PMC counters, mean per dispatch (separate --pmc passes): " + k)
    for c, v in sorted(d.items()):
        lines.append(f"   {c:28s} {sum(v)/len(v):20.1f}   n={len(v)}")
lines.append("== tools/bench_extra.py line")
lines.append(open(out + "/bench.json").read().strip())
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
