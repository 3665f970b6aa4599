#!/bin/bash
# round 4, GPU call AI: forward: the tile's column-key hashing goes round the four waves instead of always wave 0 (+ 24-bit row-offset multiplies):
# attention tests, then the same-box A/B against -DT2S_FWD_CK_ROT=0
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4ai
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py -m gpu -x -q > $OUT/pytest_attn.log 2>&1 || { tail -40 $OUT/pytest_attn.log; exit 1; }
tail -2 $OUT/pytest_attn.log
rm -f $OUT/ab.txt
for rep in 1 2 3; do
  echo "== hashing wave rotates (product build)" >> $OUT/ab.txt
  T2S_PROBE_FORMS=shipped timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 0.1 2>&1 | grep "fwd " | sed -E 's/ \| bwd.*$//' >> $OUT/ab.txt
  echo "== always wave 0 (-DT2S_FWD_CK_ROT=0)" >> $OUT/ab.txt
  T2S_PROBE_FORMS=shipped T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fwd_rot0.so timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 0.1 2>&1 | grep "fwd " | sed -E 's/ \| bwd.*$//' >> $OUT/ab.txt
done
cat $OUT/ab.txt | cut -c1-150
