#!/bin/bash
# Builds tools/ablate/_build/libt2s_fwd_<name>.so: the product library with its forward-attention source replaced by the given file
#   tools/ablate/fwd_variant.sh NAME SOURCE.hip         (FWD_VARIANT_FLAGS="-DFWD_PRIO_MODE=1": extra compiler flags, e.g. a macro of the product source)
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/tools/ablate/_build
src=$root/vitxt_gqa_amd/csrc
mkdir -p $out
flags="--offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -shared -w -I$src $FWD_VARIANT_FLAGS"
/opt/rocm/bin/hipcc $flags -o $out/libt2s_fwd_$1.so $(ls $src/*.hip $src/*.cpp | grep -v attn_fwd_bf16.hip) $2
ls -la $out/libt2s_fwd_$1.so
