#!/bin/bash
# PMC counters of the batched-mask scan (per mask x target pair)
cd /tmp && export TMPDIR=/tmp
set -u
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
M=${MASKS:-32}; T=${TARGETS:-16384}
OUT=$R/gpurun_out/pmc_multi
rm -rf $OUT; mkdir -p $OUT
ARGS="$R/tools/bench_extra.py --masks $M --targets $T --reps 1 --check 0"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT --output-format csv -d $OUT/c -- python3 $ARGS > $OUT/c.log 2>&1
python3 - "$OUT" "$M" "$T" <<'PY'
import csv, glob, sys, collections
out, M, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
agg = collections.defaultdict(list)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "scan_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"   {c:24s} {sum(v)/len(v)/(M*T):14.1f} per pair   (n={len(v)})")
PY
