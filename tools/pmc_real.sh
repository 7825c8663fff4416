#!/bin/bash
# PMC counters per target of the sparse scan on the REAL images (tools/real_image_rate.py --sparse-only --mask M):
#   MASK (default em_12191), COPIES (default 8192), LIB (A/B build: build_ab/libcds_NAME.so), TAG
cd /tmp && export TMPDIR=/tmp
set -u
R=${GRAFT_REPO_ROOT:?run through gpurun}
C=${COPIES:-8192}
[ -n "${LIB:-}" ] && export CDS_HIP_LIBRARY=$R/build_ab/libcds_$LIB.so
OUT=$R/gpurun_out/pmc_real${TAG:-}
rm -rf $OUT; mkdir -p $OUT
ARGS="$R/tools/real_image_rate.py --copies $C --reps 2 --sparse-only --mask ${MASK:-em_12191}"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/p$i.log 2>&1
done
python3 - "$OUT" "$((C*4))" <<'PY'
import csv, glob, sys, collections
out, T = sys.argv[1], float(sys.argv[2])
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for line in open(f).read().splitlines():
        if "sparse_scan" in line or "Name" in line:
            print(line[:60], line[-80:])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if "sparse_scan" not in k:
        continue
    print("== counters of", k, "(mean per dispatch, and per target)")
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        print(f"   {c:26s} {m:18.1f} {m / T:14.2f}   n={len(v)}")
PY
find $OUT -name '*.csv' -size +2M -delete
