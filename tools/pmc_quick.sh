#!/bin/bash
# quick PMC comparison of scan-kernel variants selected by CDS_SCAN_DEBUG values in $DBGS
cd /tmp && export TMPDIR=/tmp
set -u
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
for dbg in ${DBGS:-0 3}; do
  OUT=$R/gpurun_out/pmcq_$dbg
  rm -rf $OUT; mkdir -p $OUT
  export CDS_SCAN_DEBUG=$dbg
  ARGS="$R/bench.py --chunk 16384 --steps 3 --warmup 1 --no-cpu-baseline"
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- python3 $ARGS > $OUT/c.log 2>&1
  python3 - "$OUT" "$dbg" <<'PY'
import csv, glob, sys, collections
out, dbg = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "scan_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== debug", dbg)
for c, v in sorted(agg.items()):
    print(f"   {c:24s} {sum(v)/len(v)/16384:14.1f} per target")
PY
done
