#!/bin/bash
# round 4, GPU call AF: what the exponentials cost in the fused backward: timing-only build with a plain multiply in place of each v_exp_f32
# (-DFB_ABL=16, results wrong) against the same source without the switch, with and without dropout
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4af
mkdir -p $OUT
set -e
cd $REPO
rm -f $OUT/ab.txt
for rep in 1 2; do
  for d in 0.1 0.0; do
    for v in base abl16; do
      echo "== $v, dropout $d" >> $OUT/ab.txt
      T2S_PROBE_FORMS=shipped T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fb_$v.so timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $d 2>&1 | grep "bwd fused" >> $OUT/ab.txt
    done
  done
done
cat $OUT/ab.txt | cut -c1-150
