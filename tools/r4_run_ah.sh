#!/bin/bash
# round 4, GPU call AH: the opt-out switches still work end to end on the final code: model / training / full-size tests with the atomic dQ form
# and with the two-kernel backward
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4ah
mkdir -p $OUT
set -e
cd $REPO
T2S_ATTN_BWD_DQ=atomic timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py tests/test_training_gpu.py tests/test_fullsize_gpu.py tests/test_crosscheck_gpu.py -m gpu -x -q > $OUT/pytest_atomic.log 2>&1 || { tail -40 $OUT/pytest_atomic.log; exit 1; }
tail -2 $OUT/pytest_atomic.log
T2S_ATTN_BWD_FUSED=0 timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py tests/test_training_gpu.py tests/test_crosscheck_gpu.py -m gpu -x -q > $OUT/pytest_two_kernel.log 2>&1 || { tail -40 $OUT/pytest_two_kernel.log; exit 1; }
tail -2 $OUT/pytest_two_kernel.log
