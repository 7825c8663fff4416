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
# (gpurun kills a command that writes nothing for 7 minutes; a profiled bench.py prints at its end only)
( while true; do date +%T >> $OUT/heartbeat.txt; sleep 45; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp
H_ARGS="$R/bench.py --no-aux --no-cpu-baseline --verify-windows 3"
F_ARGS="$R/bench.py --no-cpu-baseline --verify-windows 3 --steps 3 --warmup 1"
run() {  # name, seconds, rocprofv3 options..., then the python args
    local name=$1 secs=$2; shift 2
    timeout -k 10 $secs rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "pass $name done: $?"
}
# FULL = "" (headline, headline32, real, overlap), "full" (those + the full command), "fullonly" (the full command alone: a second
# gpurun call of <= 20 minutes; merge the two scan_pmc.json files with tools/merge_pmc.py)
if [ "$FULL" != "fullonly" ]; then
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
# the overlapping regime: the reference's LM test MIPs against an EM mask that overlaps them (records32, the pilot's choice there),
# and bench.py's overlapping-synthetic library -- classes real_images_em_12191 / overlapping_synthetic, each with its counters
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"
R_ARGS="$R/tools/real_image_rate.py --copies 8192 --reps 3 --sparse-only --mask em_12191"
O_ARGS="$R/tools/overlap_rate.py --targets 131072 --no-cpu-baseline"
M_ARGS="$R/tools/bench_sparse.py --masks 64 --targets 262144 --reps 2 --warmup 1 --check 0"
for cmd in real overlap multi; do
    A="$R_ARGS"; [ $cmd = overlap ] && A="$O_ARGS"; [ $cmd = multi ] && A="$M_ARGS"
    run trace_$cmd 300 --kernel-trace --stats --output-format csv -d $OUT/trace_$cmd -- python3 $A
    run pmc_${cmd}_fetch 300 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${cmd}_fetch -- python3 $A
    run pmc_${cmd}_sq 300 --pmc $SQ1 --output-format csv -d $OUT/pmc_${cmd}_sq -- python3 $A
    run pmc_${cmd}_sq2 300 --pmc $SQ2 --output-format csv -d $OUT/pmc_${cmd}_sq2 -- python3 $A
    # request sizes at the memory side of L2: FETCH_SIZE counts 64 B per request whatever its size (tools/fetchcal); these give the bytes
    run pmc_${cmd}_req 300 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc_${cmd}_req -- python3 $A
    [ $cmd = multi ] && run pmc_${cmd}_l2 300 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc_${cmd}_l2 -- python3 $A
done
run pmc_headline_req 300 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc_headline_req -- python3 $H_ARGS
fi
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"
if [ -n "$FULL" ]; then   # (each pass of the full command takes 4-7 minutes under the profiler: two passes per gpurun call)
    run trace_full 560 --kernel-trace --stats --output-format csv -d $OUT/trace_full -- python3 $F_ARGS
    run pmc_full_fetch 560 --pmc FETCH_SIZE GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc_full_fetch -- python3 $F_ARGS
fi
# (the full command scans the 1,024-mask set three times: aux.mask_batch and its end-to-end leg through the processor, which runs twice)
cd $R && python3 tools/summarize_profile.py $OUT --tag $TAG --extra-multi 2147483648 --write-json $OUT/scan_pmc.json
# the big per-dispatch CSVs stay on the box: only the logs, the summary and the JSON travel back
find $OUT -name '*.csv' -size +8M -delete
