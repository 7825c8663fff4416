#!/bin/bash
# headline and loads-only twin for other partition counts of the target range (CDS_SPARSE_PARTS)
for p in ${1:-8 32 128 256 512}; do CDS_SPARSE_PARTS=$p timeout -k 10 300 python3 bench.py --no-aux --no-cpu-baseline --verify-windows 3 --steps 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('parts $p', round(d['value']/1e6,1), [round(x,3) for x in r['kernel_ms_min_median_max']], 'loads only', round(r['load_only']['kernel_ms_per_launch'],3))"; done
