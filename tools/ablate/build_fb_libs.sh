#!/bin/bash
# Builds variant libraries of the product with ONLY the fused-backward source replaced, HERE (hipcc cross-compiles without a GPU; the .so
# files under tools/ablate/_build/ travel to the GPU box with the snapshot), so that no GPU-minute is spent compiling:
#   tools/ablate/build_fb_libs.sh NAME=SOURCE.hip [NAME=SOURCE.hip ...]   ->  tools/ablate/_build/libt2s_fb_NAME.so
# The other 15 sources are compiled once into tools/ablate/_build/obj/ (rebuilt when a product source is newer).
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/tools/ablate/_build
src=$root/vitxt_gqa_amd/csrc
mkdir -p $out/obj
flags="--offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -w -I$src -I$root/include"
objs=""
for f in $(ls $src/*.hip $src/*.cpp | grep -v attn_bwd_fused_bf16.hip); do
  o=$out/obj/$(basename $f).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find $src -name '*.h' -newer $o)" ] || [ -n "$(find $src -name '*.inc' -newer $o)" ]; then
    /opt/rocm/bin/hipcc $flags -c -o $o $f &
  fi
  objs="$objs $o"
done
wait
for spec in "$@"; do
  name=${spec%%=*}; file=${spec#*=}
  ( /opt/rocm/bin/hipcc $flags -c -o $out/obj/fb_$name.o $file && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libt2s_fb_$name.so $objs $out/obj/fb_$name.o && ls -la $out/libt2s_fb_$name.so ) &
done
wait
