set -e
mkdir -p gpurun_out/r05a
R="timeout -k 10 300 python3 tools/real_image_rate.py --copies 8192 --reps 3 --sparse-only"
echo "== default" > gpurun_out/r05a/real.txt;  $R 2>&1 | grep '^{"mask' | cut -c1-200 >> gpurun_out/r05a/real.txt
for kb in 80 56; do echo "== LDS_KB=$kb" >> gpurun_out/r05a/real.txt; CDS_SPARSE_LDS_KB=$kb $R 2>&1 | grep '^{"mask' | cut -c1-200 >> gpurun_out/r05a/real.txt; done
echo "== WPE8 (64 vgpr, one WG)" >> gpurun_out/r05a/real.txt; CDS_SPARSE_WPE8=1 $R 2>&1 | grep '^{"mask' | cut -c1-200 >> gpurun_out/r05a/real.txt
echo "== nb8" >> gpurun_out/r05a/real.txt; CDS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libcds_nb8.so $R 2>&1 | grep '^{"mask' | cut -c1-200 >> gpurun_out/r05a/real.txt
echo "== scan16 forced" >> gpurun_out/r05a/real.txt; CDS_SPARSE_SCAN16=1 $R 2>&1 | grep '^{"mask' | cut -c1-200 >> gpurun_out/r05a/real.txt
cat gpurun_out/r05a/real.txt
