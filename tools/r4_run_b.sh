#!/bin/bash
# round 4, GPU call B: the dQ hand-off - kernel tests, A/B against the atomic form (same process), then the rest of call A
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4b
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py -m gpu -x -q -s -k "fused or bwd or attention or pruned" > $OUT/pytest_attn.log 2>&1 || { tail -60 $OUT/pytest_attn.log; exit 1; }
tail -3 $OUT/pytest_attn.log
timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 7 0.1 > $OUT/attn_probe_b32.txt 2>&1
timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 7 0.0 >> $OUT/attn_probe_b32.txt 2>&1
cat $OUT/attn_probe_b32.txt
timeout -k 10 300 python3 tools/gemm_epilogue_probe.py > $OUT/gemm_epilogue_probe.txt 2>&1 || { tail -30 $OUT/gemm_epilogue_probe.txt; exit 1; }
cat $OUT/gemm_epilogue_probe.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
timeout -k 10 600 python3 bench.py --no-cpu-baseline > $OUT/bench_handoff.json 2> $OUT/bench.err
T2S_ATTN_BWD_DQ=atomic timeout -k 10 600 python3 bench.py --no-cpu-baseline > $OUT/bench_atomic.json 2>> $OUT/bench.err
python3 -c "
import json
for n in ('handoff','atomic'):
    d=json.load(open('$OUT/bench_%s.json'%n)); r=d['roofline']
    print(n, 'ms/step %.1f'%d['ms_per_step'], 'drop0 %.1f'%d['dropout_0']['ms_per_step'], 'fused_avg %.2f'%r['fused_avg_launch_ms'], 'frac %.3f'%r['frac'], 'loss', d['loss'])
"
