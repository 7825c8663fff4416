#!/bin/bash
# phase split of the batched-mask scan (timing-only ablations, results are wrong with debug != 0)
for dbg in ${DBGS:-0 4 1 2 3}; do
  CDS_SCAN_DEBUG=$dbg python tools/bench_extra.py --masks ${MASKS:-32} --targets ${TARGETS:-16384} --reps 2 --check 0 2>/dev/null | tail -1 > /tmp/o.json
  python - "$dbg" <<'PY'
import sys, json
d = json.loads(open('/tmp/o.json').read())
print("multi debug", sys.argv[1], round(d["comparisons_per_s_kernels"]/1e6, 3), "M/s kern_ms", round(d["kernel_ms_avg"], 3), "launches", d["launches"])
PY
done
