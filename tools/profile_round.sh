#!/bin/bash
# Round evidence: bench.py lines + rocprofv3 kernel-trace stats + separate PMC passes, summarised PER LAUNCH CLASS
# (tools/summarize_profile.py) so that profiles/TAG_summary.txt alone reproduces ms per launch and bytes per target.
# usage (on the GPU box): bash tools/profile_round.sh r03a [full]
#   two profiled commands: `headline` = bench.py --no-aux --no-cpu-baseline --verify-windows 3 (only the timed 1-mask scans and
#   the dense verification windows launch scan kernels), `full` (optional 2nd argument) = the same with the aux legs.
set -u
TAG=${1:-rXX}
FULL=${2:-}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
H_ARGS="$R/bench.py --no-aux --no-cpu-baseline --verify-windows 3"
F_ARGS="$R/bench.py --no-cpu-baseline --verify-windows 3 --steps 3 --warmup 1"
run() {  # name, seconds, rocprofv3 options..., then the python args
    local name=$1 secs=$2; shift 2
    timeout -k 10 $secs rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "pass $name done: $?"
}
timeout -k 10 300 python3 $H_ARGS > $OUT/bench_headline.json 2> $OUT/bench_headline.err; echo "bench headline: $?"
run trace_headline 300 --kernel-trace --stats --output-format csv -d $OUT/trace_headline -- python3 $H_ARGS
run pmc_headline_fetch 300 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_headline_fetch -- python3 $H_ARGS
run pmc_headline_write 300 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_headline_write -- python3 $H_ARGS
run pmc_headline_sq 300 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_headline_sq -- python3 $H_ARGS
run pmc_headline_sq2 300 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_headline_sq2 -- python3 $H_ARGS
# the same headline through round 3's kernel (the 4-byte records), for the side-by-side class sparse_scan_records32
export CDS_SPARSE_SCAN32=1
run trace_headline32 300 --kernel-trace --stats --output-format csv -d $OUT/trace_headline32 -- python3 $H_ARGS
run pmc_headline32_fetch 300 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_headline32_fetch -- python3 $H_ARGS
run pmc_headline32_sq 300 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_headline32_sq -- python3 $H_ARGS
unset CDS_SPARSE_SCAN32
if [ -n "$FULL" ]; then
    run trace_full 500 --kernel-trace --stats --output-format csv -d $OUT/trace_full -- python3 $F_ARGS
    run pmc_full_fetch 500 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_full_fetch -- python3 $F_ARGS
    run pmc_full_sq 500 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_full_sq -- python3 $F_ARGS
fi
# (the full command scans the 1,024-mask set three times: aux.mask_batch and its end-to-end leg through the processor, which runs twice)
cd $R && python3 tools/summarize_profile.py $OUT --tag $TAG --extra-multi 2147483648 --write-json $OUT/scan_pmc.json
# the big per-dispatch CSVs stay on the box: only the logs, the summary and the JSON travel back
find $OUT -name '*.csv' -size +8M -delete
