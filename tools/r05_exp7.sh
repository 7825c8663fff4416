#!/bin/bash
# the shard's device decoders with and without the SPANS code: kernel times (rocprofv3 --stats) and end-to-end rates
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in spans nospans; do
  [ $v = nospans ] && export CDS_VARIANT_NO_SPANS=1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$v -- python3 $R/tools/gradscore_files_rate.py --targets 4096 --threads 16 --batch 512 --real --shard > $O/$v.txt 2> $O/$v.err
  tail -1 $O/$v.txt | cut -c1-260; (cd $R && [ $v = spans ] && timeout -k 10 300 python3 -m pytest tests/test_mipio.py -x -q -m gpu 2>&1 | tail -2)
  f=$(find $O/trace_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v kernel stats"; head -8 $f | cut -c1-200
done
find $O -name '*.csv' -size +4M -delete
