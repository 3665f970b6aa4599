#!/bin/bash
# round 4, GPU call AD: the same 110 training steps with the two-kernel backward (T2S_ATTN_BWD_FUSED=0: no fused kernel, no hand-off) -
# is the gradient-norm spike around step 90-100 of the fixed-batch run a property of the optimisation or of the new kernels?
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4ad
mkdir -p $OUT
set -e
cd $REPO
T2S_ATTN_BWD_FUSED=0 timeout -k 10 600 python3 tools/train_curve.py 110 64 2>&1 | grep -v "amdgpu.ids" > $OUT/train_curve_two_kernel.txt
awk 'NR<=3 || NR%10==2 || /fused/' $OUT/train_curve_two_kernel.txt
