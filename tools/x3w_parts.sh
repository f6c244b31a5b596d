#!/bin/bash
# Timing experiments on conv3x3_x3w_kernel (results discarded): builds variants of libdcs_hip.so with parts of the kernel's
# loop switched off (DCS_X3W_EXP bits in csrc/conv_split.hip: 1 no halo replacement, 2 no weight loads, 4 no fragment reads,
# 8 no epilogue) into build/x3w/.
#   tools/x3w_parts.sh 0 1 2 4 8 7 15 && gpurun -- 'for v in 0 1 2 4 8 7 15; do DCS_LIB=$PWD/build/x3w/libdcs_x3w_$v.so python tools/conv_bench.py fwd 5 _; done'
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/doubly-contrastive-semseg_amd/dcs_amd/csrc
OUT=$ROOT/build/x3w
mkdir -p $OUT
make -C $CS -j8 >/dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CS -Wall -Wno-unused-function -fno-slp-vectorize"
OTHERS="$CS/conv_igemm.o $CS/bn_elementwise.o $CS/resize_pool.o $CS/losses.o $CS/label_boundary.o $CS/contrast_fused.o $CS/sampler_host.o $CS/config_host.o"
build() {
  /opt/rocm/bin/hipcc $FLAGS -DDCS_X3W_EXP=$1 $EXTRA -c $CS/conv_split.hip -o $OUT/conv_split_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/conv_split_$1.o $OTHERS -o $OUT/libdcs_x3w_$1.so
  rm -f $OUT/conv_split_$1.o
}
n=0
for v in "$@"; do
  build $v &
  n=$((n+1)); if [ $((n % 4)) = 0 ]; then wait; fi
done
wait
ls -la $OUT
