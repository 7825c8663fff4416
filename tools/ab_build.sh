#!/bin/bash
# A/B builds of the library: tools/ab_build.sh NAME "-DSP_EXP=3" -> build_ab/libcds_NAME.so (use with CDS_HIP_LIBRARY=...)
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/build_ab/$NAME
mkdir -p $D
cd $R/colormipsearch_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-unused-result $*"
# a shared object is reused only while it is newer than its source AND every header: round 4's `s16u` build linked a cds_scan.o that
# predated a new cds_context member and its context refused to start (profiles/README.md)
for f in cds_host cds_scan cds_synth cds_shape cds_imageio cds_variants cds_diag cds_multi; do
    o=$R/build_ab/common_$f.o
    [ -f $o ] && [ $o -nt $f.hip ] && [ $o -nt cds_internal.hpp ] && [ $o -nt cds_lut.hpp ] && [ $o -nt cds_imageio.hpp ] && [ $o -nt $R/include/cds_hip.h ] && [ $o -nt $R/include/cds_hip_debug.h ] || /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-unused-result -c $f.hip -o $R/build_ab/common_$f.o &
done
/opt/rocm/bin/hipcc $FLAGS -c cds_sparse.hip -o $D/cds_sparse.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_ab/libcds_$NAME.so $D/cds_sparse.o $R/build_ab/common_*.o -lz -ldl
echo built $R/build_ab/libcds_$NAME.so
