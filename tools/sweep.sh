#!/bin/bash
# usage: tools/sweep.sh  (runs on the GPU box) -- scan kernel launch-size sweep
CH=${CHUNK:-16384}
for w in ${WGS:-4 8 16 32 64}; do
  CDS_SCAN_WGS_PER_CU=$w python bench.py --chunk $CH --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python - "$w" <<'PY'
import sys, json
d = json.loads(open('/tmp/o.json').read())
print("wgs/cu", sys.argv[1], round(d["value"]/1e6, 3), "M/s kern_ms", round(d["roofline"]["kernel_ms_per_launch"], 3), "frac", round(d["roofline"]["frac"], 3))
PY
done
