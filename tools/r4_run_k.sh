#!/bin/bash
# round 4, GPU call K: column keys per 256-row window - the whole GPU suite, the attention probe, then the final core + busy profiles
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4k
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log; grep "dropout mask at" $OUT/pytest.log || true
for d in 0.1 0.0; do
  timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 7 $d 2>&1 | grep -v amdgpu >> $OUT/attn_probe_b32.txt
done
cat $OUT/attn_probe_b32.txt
bash tools/profile_round.sh r04 core
bash tools/profile_round.sh r04 busy
