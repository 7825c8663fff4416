#!/bin/bash
# multi-mask kernel, small A/B: LDS reciprocal table instead of v_rcp_f32 per record, signed bit-field extract for the union map's verdict,
# the class term of the hue bin as one multiply + and (build_ab/tweak: an edited copy of cds_sparse.hip; the tree's kernel is untouched)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q; mkdir -p $O
A="tools/bench_sparse.py --masks 64 --targets 262144 --reps 3 --warmup 1 --check 16"
for r in 1 2; do
  python $A 2>/dev/null | tail -1 | cut -c1-300 | tee -a $O/base.txt
  CDS_HIP_LIBRARY=$PWD/build_ab/libcds_tweak.so python $A 2>/dev/null | tail -1 | cut -c1-300 | tee -a $O/tweak.txt
done
CDS_HIP_LIBRARY=$PWD/build_ab/libcds_tweak.so timeout -k 10 400 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k "multi" 2>&1 | tail -2
