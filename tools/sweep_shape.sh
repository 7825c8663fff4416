#!/bin/bash
for rows in 8 16 32; do for w in 8 16 32; do
  echo -n "rows $rows wgs $w: "; CDS_SHAPE_ROWS=$rows CDS_SHAPE_WGS_PER_CU=$w python tools/bench_extra.py --gradscore --masks 16 --top 500 --targets 8192 --check 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['pairs_per_s_kernel']/1e6,3),'M pairs/s', round(d['kernel_ms'],2),'ms', round(d['algorithmic_GBps']),'GB/s')"
done; done
