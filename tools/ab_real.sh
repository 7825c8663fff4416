#!/bin/bash
# real-image rate of A/B builds: tools/ab_real.sh "0 1 3" [copies]
for v in ${1:-0}; do echo "== exp$v"; CDS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libcds_exp$v.so timeout -k 10 300 python3 tools/real_image_rate.py --copies ${2:-8192} --reps 3 --sparse-only 2>&1 | grep "^{\"mask" | cut -c1-140; done
