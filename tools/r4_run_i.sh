#!/bin/bash
# round 4, GPU call I: running-sum prefetch by LDS-DMA (shipped build) vs register loads (-DFB_HO_PREFETCH=0): tests, A/B at L = 10 132 and 3 000
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4i
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py tests/test_model_gpu.py tests/test_crosscheck_gpu.py tests/test_vit_gpu.py -m gpu -x -q -k "fused or bwd or attention or pruned or gradients or shared_prefix or crosscheck or vit or dropout" > $OUT/pytest_attn.log 2>&1 || { tail -60 $OUT/pytest_attn.log; exit 1; }
tail -2 $OUT/pytest_attn.log
for L1 in 10120 3000; do
 for d in 0.1 0.0; do
  echo "== prefetch by LDS-DMA (shipped), L1=$L1 dropout $d" >> $OUT/attn_probe.txt
  timeout -k 10 300 python3 tools/attn_probe.py 32 $L1 0.7 12 7 $d 2>&1 | grep "fused/\|status" >> $OUT/attn_probe.txt
  echo "== register loads (-DFB_HO_PREFETCH=0), L1=$L1 dropout $d" >> $OUT/attn_probe.txt
  T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fb_hp0.so timeout -k 10 300 python3 tools/attn_probe.py 32 $L1 0.7 12 7 $d 2>&1 | grep "fused/\|status" >> $OUT/attn_probe.txt
 done
done
cat $OUT/attn_probe.txt
