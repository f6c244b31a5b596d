#!/bin/bash
# Builds variants of libdcs_hip.so whose conv_split.hip is compiled WITH the SLP vectoriser (the product build uses
# -fno-slp-vectorize, csrc/Makefile) plus one source experiment each (DCS_SLP_EXP in csrc/conv_shared.h), into
# build/slp/: the bisect of the "wrong, run-to-run different BatchNorm-backward sums" defect.
#   tools/slp_variants.sh && gpurun -- 'for v in 0 1 2 3 np; do DCS_LIB=$PWD/build/slp/libdcs_slp_$v.so python tools/x3_check2.py; done'
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/doubly-contrastive-semseg_amd/dcs_amd/csrc
OUT=$ROOT/build/slp
mkdir -p $OUT
make -C $CS -j8 >/dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CS -Wall -Wno-unused-function"
OTHERS="$CS/conv_igemm.o $CS/bn_elementwise.o $CS/resize_pool.o $CS/losses.o $CS/label_boundary.o $CS/contrast_fused.o $CS/sampler_host.o $CS/config_host.o"
build() {  # name, extra flags
  /opt/rocm/bin/hipcc $FLAGS $2 -c $CS/conv_split.hip -o $OUT/conv_split_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/conv_split_$1.o $OTHERS -o $OUT/libdcs_slp_$1.so
  rm -f $OUT/conv_split_$1.o
}
build 0 "" & build 1 "-DDCS_SLP_EXP=1" & build 2 "-DDCS_SLP_EXP=2" & build 3 "-DDCS_SLP_EXP=3" & build np "-DDCS_EPI_NO_PREFETCH" &
wait
ls -la $OUT
