#!/bin/bash
# Host-side AddressSanitizer run of the image decoder, the zip archive reader (csrc/cds_imageio.hip) and the variant shard's packer and
# reader (csrc/cds_variants.hip) -- runs on the CPU, no GPU needed.
# The library is compiled with -Xarch_host -fsanitize=address (device code untouched), linked into fuzz_main.cpp, and fed
# mutated copies (truncations, header byte flips, random byte flips) of small TIFF / PNG files in every supported variant.
#   bash tools/asan_fuzz/run.sh [iterations per file, default 20000]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=${TMPDIR:-/tmp}/cds_asan_fuzz
mkdir -p "$B" && cd "$B"
S=$ROOT/colormipsearch_amd/csrc
for f in cds_host cds_scan cds_synth cds_shape cds_sparse cds_imageio cds_variants; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -Xarch_host -fsanitize=address -fno-omit-frame-pointer \
      -ffp-contract=off -w -I"$ROOT/include" -c "$S/$f.hip" -o $f.o &
done
wait
/opt/rocm/lib/llvm/bin/clang++ -fsanitize=address -g -I"$ROOT/include" -c "$ROOT/tools/asan_fuzz/fuzz_main.cpp" -o fuzz_main.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address fuzz_main.o cds_*.o -o fuzz -lz -ldl -lpthread
python3 "$ROOT/tools/asan_fuzz/make_seeds.py"
./fuzz "${1:-20000}" seed_*.tif seed_*.png seed_archive.zip
