#!/bin/bash
# round 4, GPU call N: compact hand-off tickets (no leave-at-once workgroups between two pairs): fused-backward tests, the workgroup
# timeline, and the five backward forms side by side
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4n
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py -m gpu -x -q -k "fused or bwd or backward or handoff or hand_off" > $OUT/pytest_fused.log 2>&1 || { tail -40 $OUT/pytest_fused.log; exit 1; }
tail -2 $OUT/pytest_fused.log
rm -f $OUT/timeline.txt $OUT/attn_probe.txt
for args in "32 0.7 0.1 1" "32 0.7 0.1 0" "64 0.7 0.1 1"; do
  echo "== fused_timeline.py $args" >> $OUT/timeline.txt
  FB_TL_SAVE=$OUT/tl_$(echo $args | tr ' ' '_').npz timeout -k 10 300 python3 tools/fused_timeline.py $args 2>&1 | grep -v "amdgpu.ids" >> $OUT/timeline.txt
done
cat $OUT/timeline.txt | cut -c1-250
for d in 0.1 0.0; do
  echo "== L1=10120 dropout $d" >> $OUT/attn_probe.txt
  timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $d 2>&1 | grep "bwd \|status" >> $OUT/attn_probe.txt
done
echo "== L1=3000 dropout 0.1" >> $OUT/attn_probe.txt
timeout -k 10 300 python3 tools/attn_probe.py 32 3000 0.7 12 10 0.1 2>&1 | grep "bwd \|status" >> $OUT/attn_probe.txt
cat $OUT/attn_probe.txt | cut -c1-200
