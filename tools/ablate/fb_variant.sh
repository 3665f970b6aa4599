#!/bin/bash
# Builds tools/ablate/_build/libt2s_fbv_<name>.so: the product library with its fused-backward source replaced by the given file
#   tools/ablate/fb_variant.sh NAME SOURCE.hip          (FB_VARIANT_FLAGS="-DFB_DQ_DEPTH=6": extra compiler flags, e.g. a macro of the product source itself)
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/tools/ablate/_build
src=$root/vitxt_gqa_amd/csrc
mkdir -p $out
flags="--offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -shared -w -I$src $FB_VARIANT_FLAGS"
/opt/rocm/bin/hipcc $flags -o $out/libt2s_fbv_$1.so $(ls $src/*.hip $src/*.cpp | grep -v attn_bwd_fused_bf16.hip) $2
ls -la $out/libt2s_fbv_$1.so
