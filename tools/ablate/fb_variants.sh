#!/bin/bash
# Builds libt2s_hip variants whose fused backward is the DIAGNOSTIC source tools/ablate/attn_bwd_fused_bf16_diag.hip (cycle stamps, workgroup
# timeline, timing-only ablation switches, the register-staged / drop-word / LDS-prefetch forms: none of that is in the product source
# any more) compiled with the given -D flags, under tools/ablate/_build/:
#   tools/ablate/fb_variants.sh NAME1=FLAGS1 NAME2=FLAGS2 ...     e.g.  dbg1="-DOVL_DBG=1"
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/tools/ablate/_build
src=$root/vitxt_gqa_amd/csrc
flags="--offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -Wno-unused-function"
mkdir -p $out/obj
for f in $src/*.hip $src/*.cpp; do
  b=$(basename $f); [ $b = attn_bwd_fused_bf16.hip ] && continue
  [ $out/obj/$b.o -nt $f ] || echo $f
done | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc $flags -c {} -o $out/obj/\$(basename {}).o"
for spec in "$@"; do echo "$spec"; done | xargs -P 8 -I{} sh -c 'spec="{}"; name=${spec%%=*}; fl=${spec#*=}; /opt/rocm/bin/hipcc '"$flags"' $fl -I'"$src"' -c '"$root"'/tools/ablate/attn_bwd_fused_bf16_${FB_SRC:-diag}.hip -o '"$out"'/obj/fbv_$name.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o '"$out"'/libt2s_fb_$name.so $(ls '"$out"'/obj/*.o | grep -v "fbv_\|pw_abl\|attn_bwd_fused_bf16") '"$out"'/obj/fbv_$name.o'
ls -la $out/libt2s_fb_*.so
