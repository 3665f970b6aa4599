#!/bin/bash
# round 4, GPU call G: where the dQ hand-off's extra time goes - timing-only ablation builds (tools/ablate/fb_variants.sh, FB_HO_ABL)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4g
mkdir -p $OUT
set -e
cd $REPO
for v in base hoa1 hoa2 hoa4 hoa8 hoa11; do
  echo "== $v" >> $OUT/ho_ablation.txt
  if [ $v = base ]; then
    timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 7 0.1 2>&1 | grep "fused/" >> $OUT/ho_ablation.txt
  else
    T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fb_$v.so timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 7 0.1 2>&1 | grep "fused/" >> $OUT/ho_ablation.txt
  fi
done
cat $OUT/ho_ablation.txt
