#!/bin/bash
# usage: tools_sweep.sh  (runs on the GPU box) -- scan kernel configuration sweep
for cfg in 0 1; do for w in 2 4 8 16; do
  CDS_SCAN_CFG=$cfg CDS_SCAN_WGS_PER_CU=$w python bench.py --chunk 8192 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python - "$cfg" "$w" <<'PY'
import sys, json
d = json.loads(open('/tmp/o.json').read())
print("cfg", sys.argv[1], "wgs/cu", sys.argv[2], round(d["value"]/1e6, 3), "M/s kern_ms", round(d["roofline"]["kernel_ms_per_launch"], 3), "frac", round(d["roofline"]["frac"], 3))
PY
done; done
