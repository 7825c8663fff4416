#!/bin/bash
# FETCH_SIZE calibration for scattered loads (tools/fetchcal/fetchcal.hip): builds the tool, runs it plainly and under separate PMC
# passes, prints per mode the factor between the bytes in the touched granules and FETCH_SIZE x 1024.
#   on the GPU box: bash tools/fetchcal/run.sh [GiB]   -> gpurun_out/fetchcal/summary.txt
set -u
R=${GRAFT_REPO_ROOT:?run through gpurun}
OUT=$R/gpurun_out/fetchcal
rm -rf $OUT; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/fetchcal/fetchcal.hip -o $OUT/fetchcal || exit 1
cd /tmp && export TMPDIR=/tmp
G=${1:-6}
$OUT/fetchcal $G > $OUT/plain.jsonl || exit 1
for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_REQ_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_HIT_sum"; do
    tag=$(echo $set | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$tag -- $OUT/fetchcal $G > $OUT/pmc_$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, json, sys, collections
out = sys.argv[1]
modes = [json.loads(l) for l in open(out + "/plain.jsonl") if l.startswith("{")]
cnt = collections.defaultdict(dict)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "cal<" in k:
            cnt[k.split("cal<")[1].split(">")[0]][r["Counter_Name"]] = float(r["Counter_Value"])
print("FETCH_SIZE calibration on gfx950 (tools/fetchcal): bytes in touched granules vs counters, per load shape")
for m in modes:
    c = cnt.get(m["kernel"].split("<")[1].split(">")[0], {})
    line = f"{m['mode']:10s} requested {m['bytes_requested']/1e9:7.2f} GB  64-B sectors {m['bytes_64B_sectors']/1e9:7.2f} GB  128-B lines {m['bytes_128B_lines']/1e9:7.2f} GB  {m['ms']:8.2f} ms"
    if "FETCH_SIZE" in c:
        fb = c["FETCH_SIZE"] * 1024.0
        line += f" | FETCH_SIZE {fb/1e9:7.2f} GB -> sectors/FETCH = {m['bytes_64B_sectors']/fb:5.2f}, lines/FETCH = {m['bytes_128B_lines']/fb:5.2f}, requested/FETCH = {m['bytes_requested']/fb:5.2f}"
    if "TCC_EA0_RDREQ_sum" in c:
        rd, rd32 = c["TCC_EA0_RDREQ_sum"], c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        rd64, rd128 = c.get("TCC_EA0_RDREQ_64B_sum"), c.get("TCC_EA0_RDREQ_128B_sum")
        if rd128 is not None:
            rd64 = rd64 if rd64 is not None else rd - rd32 - rd128
            line += f" | EA RDREQ {rd/1e6:8.1f} M = {rd32/1e6:.1f} M x 32 B + {rd64/1e6:.1f} M x 64 B + {rd128/1e6:.1f} M x 128 B = {(rd32*32 + rd64*64 + rd128*128)/1e9:7.2f} GB"
        else:
            line += f" | EA RDREQ {rd/1e6:8.1f} M ({rd32/1e6:8.1f} M of them 32 B) -> bytes if the rest are 64 B = {(rd32*32 + (rd-rd32)*64)/1e9:7.2f} GB"
    for k in ("TCC_REQ_sum", "TCC_MISS_sum", "TCC_HIT_sum", "TCP_TCC_READ_REQ_sum"):
        if k in c:
            line += f" | {k} {c[k]/1e6:8.1f} M"
    print(line)
PY
find $OUT -name '*.csv' -size +2M -delete; rm -f $OUT/fetchcal
