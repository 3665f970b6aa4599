#!/bin/bash
# round 4, GPU call Z: row keys / row constants of a slot fetched from LDS one slot ahead (FB_LDS_EARLY): tests, then the same-box A/B
# against -DFB_LDS_EARLY=0, then the stamps of the new build
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4z
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py -m gpu -x -q -k "fused or bwd or backward or handoff or dropout" > $OUT/pytest_fused.log 2>&1 || { tail -40 $OUT/pytest_fused.log; exit 1; }
tail -2 $OUT/pytest_fused.log
rm -f $OUT/ab.txt
for rep in 1 2 3; do
  echo "== LDS loads one slot ahead (product build)" >> $OUT/ab.txt
  T2S_PROBE_FORMS=shipped timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 0.1 2>&1 | grep "bwd fused" >> $OUT/ab.txt
  echo "== at the head of their own slot (-DFB_LDS_EARLY=0)" >> $OUT/ab.txt
  T2S_PROBE_FORMS=shipped T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fb_early0.so timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 0.1 2>&1 | grep "bwd fused" >> $OUT/ab.txt
done
cat $OUT/ab.txt | cut -c1-150
