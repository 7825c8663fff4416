#!/bin/bash
# Round-end evidence: default bench.py line + rocprofv3 kernel-trace stats + PMC passes of the same command.
# usage (on the GPU box): bash tools/profile_round.sh r01
set -u
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R && timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1; echo "pass done: $?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1; echo "pass done: $?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1; echo "pass done: $?"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1; echo "pass done: $?"
timeout -k 10 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1; echo "pass done: $?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
lines = []
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    lines.append("== rocprofv3 --kernel-trace --stats (kernel_stats.csv) of: python3 bench.py --no-cpu-baseline")
    lines.append(open(f).read().strip())
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if "scan_kernel" not in k and "gradscore" not in k:
        continue
    lines.append("== PMC counters, mean per dispatch (separate --pmc passes): " + k)
    for c, v in sorted(d.items()):
        lines.append(f"   {c:28s} {sum(v)/len(v):20.1f}   n={len(v)}")
lines.append("== bench.py line")
lines.append(open(out + "/bench.json").read().strip())
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
