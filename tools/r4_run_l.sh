#!/bin/bash
# round 4, GPU call L: the forward's K / V tiles by LDS-DMA (-DT2S_FWD_DMA=1 build) vs register staging (shipped): forward tests on the variant
# library, then the forward alone, same box back to back, with and without dropout
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4l
mkdir -p $OUT
set -e
cd $REPO
T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fwd_dma1.so timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py -m gpu -x -q -k "attention_fwd or forward_full_length or dropout_fwd or repair or shared_prefix or rows" > $OUT/pytest_fwd_dma.log 2>&1 || { tail -40 $OUT/pytest_fwd_dma.log; exit 1; }
tail -2 $OUT/pytest_fwd_dma.log
for rep in 1 2; do
 for d in 0.1 0.0; do
  echo "== shipped (register staging), dropout $d" >> $OUT/fwd_dma_ab.txt
  T2S_PROBE_FORMS=shipped timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $d 2>&1 | grep "fwd " >> $OUT/fwd_dma_ab.txt
  echo "== K / V by LDS-DMA (-DT2S_FWD_DMA=1), dropout $d" >> $OUT/fwd_dma_ab.txt
  T2S_PROBE_FORMS=shipped T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fwd_dma1.so timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $d 2>&1 | grep "fwd " >> $OUT/fwd_dma_ab.txt
 done
done
cat $OUT/fwd_dma_ab.txt | cut -c1-120
