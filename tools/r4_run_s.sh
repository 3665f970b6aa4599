#!/bin/bash
# round 4, GPU call S: where the fused hand-off form overtakes the two-kernel backward for SHORT key lists (the pos / neg MMT passes:
# <= 549 and <= 74 keys for 10 132 query rows), B = 64
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4s
mkdir -p $OUT
set -e
cd $REPO
rm -f $OUT/light.txt
for keep in 0.006 0.02 0.053 0.1 0.19; do
  for d in 0.1 0.0; do
    echo "== B=64 L1=10120 keep $keep dropout $d" >> $OUT/light.txt
    timeout -k 10 300 python3 tools/attn_probe.py 64 10120 $keep 12 10 $d 2>&1 | grep "bwd \|fwd " | cut -c1-150 >> $OUT/light.txt
  done
done
cat $OUT/light.txt
