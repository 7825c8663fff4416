#!/bin/bash
# SQ counters of the headline launches (bench.py --no-aux ...: every sparse_scan* dispatch covers --targets targets), per kernel name and per target.
# usage on the GPU box: bash tools/pmc_headline.sh NAME [env ...]   (CDS_HIP_LIBRARY=... for an A/B build)
cd /tmp && export TMPDIR=/tmp
set -u
R=${GRAFT_REPO_ROOT:?run through gpurun}
NAME=${1:-pmch}; shift || true
OUT=$R/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
ARGS="$R/bench.py --no-aux --no-cpu-baseline --verify-windows 3 --steps 5 --warmup 2"
for e in "$@"; do export "$e"; done
timeout -k 10 250 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
timeout -k 10 250 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "sparse_scan" in k:
            import re
            if "scan16" in k:
                m = re.search(r", (\d+), (true|false), (true|false)>\(", k)
                short = "scan16" + ("_loads_only" if m and m.group(2) == "true" else "") + ("_plane_in_L2" if m and m.group(3) == "true" else "")
            else:
                short = "scan32" + ("_loads_only" if re.search(r", \d+, true>\(", k) else "")
            agg[short][row["Counter_Name"]].append((float(row.get("Grid_Size", 0) or 0), float(row["Counter_Value"])))
for k, cs in sorted(agg.items()):
    print("==", k)
    for c, gv in sorted(cs.items()):
        g = max(x[0] for x in gv)
        v = [x[1] for x in gv if x[0] == g]   # (the pilot launch of the first search has a smaller grid)
        print(f"   {c:24s} {sum(v)/len(v)/1048576:12.1f} per target   ({len(v)} dispatches)")
PY
find $OUT -name '*.csv' -size +4M -delete
