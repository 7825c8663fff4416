#!/bin/bash
# usage: tools/sweep.sh  (runs on the GPU box) -- scan kernel configuration sweep
CH=${CHUNK:-16384}
for cfg in ${CFGS:-0 1 2}; do for w in ${WGS:-8 16}; do
  CDS_SCAN_CFG=$cfg CDS_SCAN_WGS_PER_CU=$w python bench.py --chunk $CH --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python - "$cfg" "$w" <<'PY'
import sys, json
d = json.loads(open('/tmp/o.json').read())
print("cfg", sys.argv[1], "wgs/cu", sys.argv[2], round(d["value"]/1e6, 3), "M/s kern_ms", round(d["roofline"]["kernel_ms_per_launch"], 3), "frac", round(d["roofline"]["frac"], 3))
PY
done; done
