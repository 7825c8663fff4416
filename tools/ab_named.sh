#!/bin/bash
# headline scan of named A/B builds (build_ab/libcds_NAME.so): tools/ab_named.sh "s16e1 s16e2"; extra environment through ENVS="A=1 B=2"
for v in ${1:-exp0}; do
  env ${ENVS:-} CDS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libcds_$v.so timeout -k 10 300 python3 bench.py --no-aux --no-cpu-baseline --verify-windows 3 --steps 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); lo=d['roofline'].get('load_only') or {}; print('$v ${ENVS:-}', round(d['roofline']['kernel_ms_per_launch'],3),'ms per 1M-target launch;', 'loads only', round(lo.get('kernel_ms_per_launch') or 0,3), 'parity' if d['value'] else 'timing-only')"
done
