#!/bin/bash
# round 4, GPU call X: empty key list in the hand-off form; the attention test files
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4x
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_dropout_gpu.py -m gpu -x -q > $OUT/pytest_attn.log 2>&1 || { tail -60 $OUT/pytest_attn.log; exit 1; }
tail -2 $OUT/pytest_attn.log
