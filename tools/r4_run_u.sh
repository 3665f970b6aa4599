#!/bin/bash
# round 4, GPU call U: 150-step loss / gradient-norm curve of the benchmark configuration on the round's final code
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4u
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 600 python3 tools/train_curve.py 150 64 2>&1 | grep -v "amdgpu.ids" > $OUT/train_curve.txt
cat $OUT/train_curve.txt
