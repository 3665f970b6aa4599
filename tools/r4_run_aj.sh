#!/bin/bash
# round 4, GPU call AJ: fused backward: a tile's dropout row keys hashed during phase B two tiles ahead instead of inside phase A
# (FB_RK_AHEAD): tests, then the same-box A/B against -DFB_RK_AHEAD=0
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4aj
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py -m gpu -x -q -k "fused or bwd or backward or handoff or dropout" > $OUT/pytest_fused.log 2>&1 || { tail -40 $OUT/pytest_fused.log; exit 1; }
tail -2 $OUT/pytest_fused.log
rm -f $OUT/ab.txt
for rep in 1 2 3; do
  echo "== row keys hashed in phase B, two tiles ahead (product build)" >> $OUT/ab.txt
  T2S_PROBE_FORMS=shipped timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 0.1 2>&1 | grep "bwd fused" >> $OUT/ab.txt
  echo "== inside phase A (-DFB_RK_AHEAD=0)" >> $OUT/ab.txt
  T2S_PROBE_FORMS=shipped T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fb_rk0.so timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 0.1 2>&1 | grep "bwd fused" >> $OUT/ab.txt
done
cat $OUT/ab.txt | cut -c1-150
