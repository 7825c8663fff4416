#!/bin/bash
# the variant shard against the files, on the reference's images (gradient ramps: two-bit code) and on synthetic variants (raw 16-bit gradients)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
python tools/gradscore_files_rate.py --targets 8192 --threads 16 --batch 512 --real --shard > $O/real_8192.txt 2> $O/real_8192.err; tail -2 $O/real_8192.txt
python tools/gradscore_files_rate.py --targets 8192 --threads 16 --batch 512 --real --shard --cold > $O/real_8192_cold.txt 2>> $O/real_8192.err; tail -1 $O/real_8192_cold.txt
python tools/gradscore_files_rate.py --targets 8192 --threads 4 --batch 512 --real --shard > $O/real_8192_t4.txt 2>> $O/real_8192.err; tail -1 $O/real_8192_t4.txt
python tools/gradscore_files_rate.py --targets 8192 --threads 1 --batch 512 --real --shard > $O/real_8192_t1.txt 2>> $O/real_8192.err; tail -1 $O/real_8192_t1.txt
python tools/gradscore_files_rate.py --targets 2048 --threads 16 --batch 256 --shard > $O/synth_2048.txt 2>> $O/real_8192.err; tail -2 $O/synth_2048.txt
timeout -k 10 300 python -m pytest tests/test_cpp_mirror.py -x -q -m gpu 2>&1 | tail -3
