#!/bin/bash
# Builds tools/ablate/_build/libt2s_gemm_<name>.so: the product library with its GEMM source (gemm_bf16.hip) replaced by the given file
#   tools/ablate/gemm_variant.sh NAME SOURCE.hip     (reuses the objects of tools/ablate/build_fb_libs.sh where they exist)
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/tools/ablate/_build
src=$root/vitxt_gqa_amd/csrc
mkdir -p $out/obj
flags="--offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -w -I$src -I$root/include"
objs=""
for f in $(ls $src/*.hip $src/*.cpp | grep -v gemm_bf16.hip); do
  o=$out/obj/$(basename $f).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find $src -name '*.h' -newer $o)" ] || [ -n "$(find $src -name '*.inc' -newer $o)" ]; then
    /opt/rocm/bin/hipcc $flags -c -o $o $f &
  fi
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc $flags -c -o $out/obj/gemm_$1.o $2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libt2s_gemm_$1.so $objs $out/obj/gemm_$1.o
ls -la $out/libt2s_gemm_$1.so
