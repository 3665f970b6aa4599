#!/bin/bash
# round 4, GPU call M: workgroup timeline of the fused backward (tools/fused_timeline.py, -DFB_TIMELINE build)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4m
mkdir -p $OUT
set -e
cd $REPO
for args in "32 0.7 0.1 1" "32 0.7 0.1 0" "32 0.7 0.0 1" "64 0.7 0.1 1"; do
  echo "== fused_timeline.py $args" >> $OUT/timeline.txt
  timeout -k 10 300 python3 tools/fused_timeline.py $args 2>&1 | grep -v "amdgpu.ids" >> $OUT/timeline.txt
done
cat $OUT/timeline.txt | cut -c1-250
