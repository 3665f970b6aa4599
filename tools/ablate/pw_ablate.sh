#!/bin/bash
# Builds libt2s_hip variants whose one-wave-per-SIMD forward kernel lacks one ingredient (PW_ABL bit mask, see
# tools/ablate/attn_fwd_pw_bf16.hip; the forward source linked with it must be tools/ablate/attn_fwd_bf16_diag.hip, which still routes T2S_ATTN_FWD_PW=1 to it) under tools/ablate/_build/, for launch-time experiments:
#   tools/ablate/pw_ablate.sh build 0 1 2 4 8 16 ...      (here, no GPU needed)
#   tools/ablate/pw_ablate.sh run   0 1 2 4 8 16 ...      (on the GPU box: one attn_probe line per variant and dropout setting)
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/tools/ablate/_build
src=$root/vitxt_gqa_amd/csrc
flags="--offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -Wno-unused-function"
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p $out/obj
  for f in $src/*.hip $src/*.cpp; do
    b=$(basename $f); [ $b = attn_fwd_pw_bf16.hip ] && continue
    [ $out/obj/$b.o -nt $f ] || echo $f
  done | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc $flags -c {} -o $out/obj/\$(basename {}).o"
  for v in "$@"; do echo $v; done | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc $flags -DPW_ABL={} -I$src -c $root/tools/ablate/attn_fwd_pw_bf16.hip -o $out/obj/pw_abl{}.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libt2s_pw_abl{}.so \$(ls $out/obj/*.o | grep -v pw_abl) $out/obj/pw_abl{}.o"
  ls -la $out/libt2s_pw_abl*.so
else
  for v in "$@"; do
    for d in 0.1 0.0; do
      echo "abl=$v drop=$d $(T2S_HIP_LIB=$out/libt2s_pw_abl$v.so timeout -k 10 120 python $root/tools/attn_probe.py 32 10120 0.7 12 5 $d 2>&1 | grep -o 'fwd [0-9.]* ms')"
    done
  done
fi
