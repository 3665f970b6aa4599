#!/bin/bash
# forward attention with 32 rows per wave (three workgroups per CU) against the product's 64 rows per wave, same box, interleaved
OUT=gpurun_out/${1:-qb1}; mkdir -p $OUT
T2S_ATTN_FWD_QB1=1 timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_dropout_gpu.py -m gpu -x -q -k "fwd or forward or dropout or attention" > $OUT/pytest_qb1.log 2>&1 || { tail -30 $OUT/pytest_qb1.log; exit 1; }
tail -1 $OUT/pytest_qb1.log
for dp in 0.1 0.0; do for rep in 1 2 3; do
  echo "== product (QB = 2), dropout $dp"; T2S_PROBE_FORMS=shipped timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $dp 2>&1 | grep "fwd " | sed -E 's/ \| bwd.*$//'
  echo "== QB = 1, dropout $dp"; T2S_ATTN_FWD_QB1=1 T2S_PROBE_FORMS=shipped timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $dp 2>&1 | grep "fwd " | sed -E 's/ \| bwd.*$//'
done; done | tee $OUT/fwd_qb1_ab.txt
