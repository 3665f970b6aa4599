#!/bin/bash
# round 4, GPU call AA: TunableOp over the regrouped (G = 16) FFN weight-gradient GEMMs only
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4aa
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 900 python3 tools/tune_wgrad_groups.py 2>&1 | grep -v "amdgpu.ids" > $OUT/tune.txt
cat $OUT/tune.txt | cut -c1-220
cp gpurun_out/gemm_gfx950_b64_100x100_with_g16.csv $OUT/
