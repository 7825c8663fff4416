#!/bin/bash
# multi-mask kernel: record blocks in flight per wave (SM_NB 1 / 2 (tree) / 3)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05r; mkdir -p $O
A="tools/bench_sparse.py --masks 64 --targets 262144 --reps 3 --warmup 1 --check 16"
for r in 1 2; do
  for v in base nb1 nb3; do
    L=""; [ $v != base ] && L=$PWD/build_ab/libcds_$v.so
    echo -n "$v: "; CDS_HIP_LIBRARY=$L python $A 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['scan_ms'],2), round(d['comparisons_per_s']/1e6))" | tee -a $O/$v.txt
  done
done
