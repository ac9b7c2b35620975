#!/bin/bash
# Build gan-control_amd/csrc/alt/libalt_<name>.so (git-ignored, but NOT gpurun-ignored like build/: it travels to the GPU box): the standard library with ONE translation
# unit recompiled under extra flags -- conv_bf16x3.hip by default, or the file named by the third argument (a csrc/ file name such as conv_s2ws.hip, or another copy of one,
# e.g. `git show HEAD:gan-control_amd/csrc/conv_bf16x3.hip > /tmp/conv_bf16x3.hip`: the base name selects the object it replaces) -- for same-box A/B runs (GANCONTROL_HIP_LIB=...).
#   tools/build_alt.sh <name> "<extra flags>" [source file]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)/gan-control_amd/csrc
name=$1; extra=$2; src=${3:-$R/conv_bf16x3.hip}
[ -f "$src" ] || src=$R/$src
unit=$(basename "$src" .hip)
mkdir -p $R/alt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R -I$R/../../include -Wall -Wno-unused-result $extra -c -x hip $src -o $R/build/alt_$name.o 2> /dev/null
# the experiment build carries its own stamp (gc_source_hash): counters collected on it are never quoted for the in-tree library or another experiment
stamp="alt:$name:$( (cat $src; echo "$extra") | sha256sum | cut -c1-12)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R -I$R/../../include -Wall -Wno-unused-result -DGC_SOURCE_HASH="\"$stamp\"" -c $R/capi.hip -o $R/build/alt_capi_$name.o 2> /dev/null
objs="$R/build/alt_capi_$name.o"
for u in upfirdn2d bias_act conv conv_bf16x3 conv_s2ws weight_layout pointwise warp inception small_gemm style conv_bf16; do
    [ "$u" = "$unit" ] || objs="$objs $R/build/$u.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/alt/libalt_$name.so $R/build/alt_$name.o $objs
echo built $R/alt/libalt_$name.so
