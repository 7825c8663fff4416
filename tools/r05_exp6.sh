#!/bin/bash
# instruction-cache counters of the headline launch (the stream kernel is 61 KB of code: 56 inlined copies of the evaluation round)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
H_ARGS="$R/bench.py --no-aux --no-cpu-baseline --verify-windows 3"
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_icache -- python3 $H_ARGS > $O/pmc_icache.log 2>&1; echo "pass: $?"
cd $R && python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/r05k/pmc_icache/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': cnt[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0))[:6]:
    print(k, cnt[k], {c: round(x / max(cnt[k], 1)) for c, x in v.items()})
PY
find $O -name '*.csv' -size +8M -delete
