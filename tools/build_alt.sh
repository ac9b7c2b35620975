#!/bin/bash
# Build gan-control_amd/csrc/alt/libalt_<name>.so (git-ignored, but NOT gpurun-ignored like build/: it travels to the GPU box): the standard library with conv_bf16x3.hip recompiled under extra flags (and optionally from another
# copy of the source, e.g. `git show HEAD:gan-control_amd/csrc/conv_bf16x3.hip > /tmp/old.hip`) -- for same-box A/B runs (GANCONTROL_HIP_LIB=...).
#   tools/build_alt.sh <name> "<extra flags>" [source file]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)/gan-control_amd/csrc
name=$1; extra=$2; src=${3:-$R/conv_bf16x3.hip}
mkdir -p $R/alt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R -I$R/../../include -Wall -Wno-unused-result $extra -c -x hip $src -o $R/build/alt_$name.o 2> /dev/null
objs=$(ls $R/build/{capi,upfirdn2d,bias_act,conv,weight_layout,pointwise,warp,inception,small_gemm,style,conv_bf16}.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/alt/libalt_$name.so $R/build/alt_$name.o $objs
echo built $R/alt/libalt_$name.so
