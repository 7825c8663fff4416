#!/bin/bash
# PMC counters (per target) and kernel-trace stats of the sparse scan kernel: separate rocprofv3 passes of
# tools/bench_sparse.py.  TARGETS (default 65536), EXTRA (more bench_sparse.py arguments), TAG (output directory suffix).
cd /tmp && export TMPDIR=/tmp
set -u
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
T=${TARGETS:-65536}
OUT=$R/gpurun_out/pmc_sparse${TAG:-}
rm -rf $OUT; mkdir -p $OUT
ARGS="$R/tools/bench_sparse.py --targets $T --reps 2 --check 0 ${EXTRA:-}"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/p$i.log 2>&1
done
python3 - "$OUT" "$T" <<'PY'
import csv, glob, sys, collections
out, T = sys.argv[1], float(sys.argv[2])
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats")
    print(open(f).read())
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if "sparse_scan" not in k and "sparse_multi" not in k:
        continue
    print("== counters of", k, "(mean per dispatch, and per target)")
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        print(f"   {c:26s} {m:18.1f} {m / T:14.2f}   n={len(v)}")
PY
