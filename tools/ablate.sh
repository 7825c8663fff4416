#!/bin/bash
for dbg in ${DBGS:-0 4 1 2 3}; do
  CDS_SCAN_DEBUG=$dbg python bench.py --chunk ${CHUNK:-16384} --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python - "$dbg" <<'PY'
import sys, json
d = json.loads(open('/tmp/o.json').read())
print("debug", sys.argv[1], round(d["value"]/1e6, 3), "M/s kern_ms", round(d["roofline"]["kernel_ms_per_launch"], 3))
PY
done
