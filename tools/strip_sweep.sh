for sw in 304 152 200; do
  echo "== strip_w $sw"
  CDS_SPARSE_STRIP_W=$sw timeout -k 10 300 python3 bench.py --no-aux --no-cpu-baseline --verify-windows 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', round(d['value']/1e6,1), 'M/s', round(d['roofline']['kernel_ms_per_launch'],3),'ms', d['config']['library_bytes_per_target'])"
  CDS_SPARSE_STRIP_W=$sw timeout -k 10 300 python3 tools/real_image_rate.py --copies 8192 --reps 3 --sparse-only 2>&1 | grep '^{"mask' | cut -c1-120
done
