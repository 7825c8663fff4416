# A/B of a cds_sparse.hip build flag over the three legs it can move: tools/r05_exp3.sh "aux0 aux2"
for v in ${1:-aux0}; do
  L=$GRAFT_REPO_ROOT/build_ab/libcds_$v.so
  echo "== $v"
  MASKS=64 TARGETS=262144 bash tools/ab_multi.sh "$v" 2>&1 | grep "M/s"
  CDS_HIP_LIBRARY=$L timeout -k 10 300 python3 tools/real_image_rate.py --copies 8192 --reps 3 --sparse-only --mask em_12191 2>&1 | grep '^{"mask' | cut -c1-120
  CDS_HIP_LIBRARY=$L timeout -k 10 300 python3 bench.py --no-aux --no-cpu-baseline --verify-windows 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('headline', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms; twin', round(r['load_only']['kernel_ms_per_launch'],3))"
done
