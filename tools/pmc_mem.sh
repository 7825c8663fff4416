#!/bin/bash
# memory-path PMC counters of the single-mask scan kernel (per target)
cd /tmp && export TMPDIR=/tmp
set -u
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
OUT=$R/gpurun_out/pmc_mem
rm -rf $OUT; mkdir -p $OUT
ARGS="$R/bench.py --chunk 16384 --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PERF_SEL_TOTAL_READ" \
           "TCC_REQ_sum TCC_HIT_sum" "TCC_MISS_sum TCC_TAG_STALL_sum" "TCC_BUSY_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i ($set) failed"
  echo "pass $i done"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "scan_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"   {c:40s} {sum(v)/len(v)/16384:14.1f} per target   (n={len(v)})")
PY
