#!/bin/bash
# round 4, GPU call M: the ViT producer on the wide add + LayerNorm kernel; workgroup timeline of the fused backward
# (tools/fused_timeline.py, -DFB_TIMELINE build)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4m
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 600 python3 -m pytest tests/test_vit_gpu.py -m gpu -x -q > $OUT/pytest_vit.log 2>&1 || { tail -40 $OUT/pytest_vit.log; exit 1; }
tail -2 $OUT/pytest_vit.log
rm -f $OUT/timeline.txt
for args in "32 0.7 0.1 1" "32 0.7 0.1 0"; do
  echo "== fused_timeline.py $args" >> $OUT/timeline.txt
  timeout -k 10 300 python3 tools/fused_timeline.py $args 2>&1 | grep -v "amdgpu.ids" >> $OUT/timeline.txt
done
cat $OUT/timeline.txt | cut -c1-250
