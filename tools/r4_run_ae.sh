#!/bin/bash
# round 4, GPU call AE: a fence behind the first MFMA of a tile (the compiler hoisted 20 LDS loads above it; lgkmcnt counts 15 at most, so
# the MFMA's long-arrived operands could only be waited for by draining six of the new loads): tests, then the same-box A/B against -DFB_SLOT0=0
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4ae
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py -m gpu -x -q -k "fused or bwd or backward or handoff or dropout" > $OUT/pytest_fused.log 2>&1 || { tail -40 $OUT/pytest_fused.log; exit 1; }
tail -2 $OUT/pytest_fused.log
rm -f $OUT/ab.txt
for rep in 1 2 3; do
  for d in 0.1 0.0; do
    echo "== fence behind the tile's first MFMA (product build), dropout $d" >> $OUT/ab.txt
    T2S_PROBE_FORMS=shipped timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $d 2>&1 | grep "bwd fused" >> $OUT/ab.txt
    echo "== without (-DFB_SLOT0=0), dropout $d" >> $OUT/ab.txt
    T2S_PROBE_FORMS=shipped T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fb_slot00.so timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $d 2>&1 | grep "bwd fused" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt | cut -c1-150
