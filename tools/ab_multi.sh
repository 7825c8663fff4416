#!/bin/bash
# multi-mask scan (64 masks x 262,144 targets) of A/B builds build_ab/libcds_NAME.so: tools/ab_multi.sh "sm0 sm1 sm3"; ENVS="A=1"
for v in ${1:-sm0}; do
  env ${ENVS:-} CDS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libcds_$v.so timeout -k 10 300 python3 tools/bench_sparse.py --masks ${MASKS:-64} --targets ${TARGETS:-262144} --reps 2 --warmup 1 --check 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v ${ENVS:-}', round(d['scan_ms'],2), 'ms', round(d['comparisons_per_s']/1e6,1), 'M/s')"
done
