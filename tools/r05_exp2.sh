set -e
mkdir -p gpurun_out/r05c
timeout -k 10 600 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu > gpurun_out/r05c/tests.log 2>&1 || { tail -40 gpurun_out/r05c/tests.log; exit 1; }
tail -3 gpurun_out/r05c/tests.log
timeout -k 10 600 python bench.py --no-aux --no-cpu-baseline --verify-windows 3 > gpurun_out/r05c/bench_headline.json 2> gpurun_out/r05c/bench_headline.err || { tail -20 gpurun_out/r05c/bench_headline.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05c/bench_headline.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value',d['value'],'ms',d['ms_per_step'],'frac',r['frac'],'load_only',r['load_only']['kernel_ms_per_launch'],'ratio',r['load_only']['scan_time_over_load_only_time'], 'verif', d.get('verification',{}).get('all_equal', d.get('verification')))
PY
R="timeout -k 10 300 python3 tools/real_image_rate.py --copies 8192 --reps 3 --sparse-only"
echo "== scan16 forced, plane in L2" > gpurun_out/r05c/real.txt; CDS_SPARSE_SCAN16=1 $R 2>&1 | grep '^{"mask' | cut -c1-200 >> gpurun_out/r05c/real.txt
echo "== scan16 forced, plane in LDS" >> gpurun_out/r05c/real.txt; CDS_SPARSE_SCAN16=1 CDS_SPARSE_PLANE_LDS=1 $R 2>&1 | grep '^{"mask' | cut -c1-200 >> gpurun_out/r05c/real.txt
cat gpurun_out/r05c/real.txt
