#!/bin/bash
# round 4, GPU call AC: workgroup timeline of the attention forward (tools/fwd_timeline.py, -DT2S_FWD_TIMELINE build)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4ac
mkdir -p $OUT
set -e
cd $REPO
rm -f $OUT/fwd_timeline.txt
for args in "32 0.7 0.1" "64 0.7 0.1" "64 0.053 0.1"; do
  echo "== fwd_timeline.py $args" >> $OUT/fwd_timeline.txt
  timeout -k 10 300 python3 tools/fwd_timeline.py $args 2>&1 | grep -v "amdgpu.ids" >> $OUT/fwd_timeline.txt
done
cat $OUT/fwd_timeline.txt | cut -c1-220
