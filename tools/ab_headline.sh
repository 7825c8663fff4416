#!/bin/bash
# headline scan of A/B builds (build_ab/libcds_expNAME.so): tools/ab_headline.sh "0 2 6"; extra environment through ENVS="A=1 B=2"
for v in ${1:-0}; do
  env ${ENVS:-} CDS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libcds_exp$v.so timeout -k 10 300 python3 bench.py --no-aux --no-cpu-baseline --verify-windows 3 --steps 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp$v ${ENVS:-}', round(d['roofline']['kernel_ms_per_launch'],3),'ms per 1M-target launch', 'parity' if d['value'] else 'timing-only')"
done
