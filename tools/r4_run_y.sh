#!/bin/bash
# round 4, GPU call Y: cycle stamps of the final fused backward: tile segments and the slot classes of phase A, with and without dropout
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4y
mkdir -p $OUT
set -e
cd $REPO
rm -f $OUT/stamps.txt
for d in 0.1 0.0; do
  echo "== fused_stamps.py 32 0.7 $d 1 (tile segments)" >> $OUT/stamps.txt
  timeout -k 10 300 python3 tools/fused_stamps.py 32 0.7 $d 1 2>&1 | grep -v "amdgpu.ids" >> $OUT/stamps.txt
  echo "== FB_SLOTS=1 fused_stamps.py 32 0.7 $d 1 (slot classes of phase A)" >> $OUT/stamps.txt
  FB_SLOTS=1 timeout -k 10 300 python3 tools/fused_stamps.py 32 0.7 $d 1 2>&1 | grep -v "amdgpu.ids" >> $OUT/stamps.txt
done
cat $OUT/stamps.txt | cut -c1-200
