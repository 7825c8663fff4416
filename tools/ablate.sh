#!/bin/bash
for dbg in 0 1 2 3; do
  CDS_SCAN_DEBUG=$dbg CDS_SCAN_CFG=${CFG:-0} CDS_SCAN_WGS_PER_CU=16 python bench.py --chunk 16384 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python - "$dbg" <<'PY'
import sys, json
d = json.loads(open('/tmp/o.json').read())
print("debug", sys.argv[1], round(d["value"]/1e6, 3), "M/s kern_ms", round(d["roofline"]["kernel_ms_per_launch"], 3))
PY
done
