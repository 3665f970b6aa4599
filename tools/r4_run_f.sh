#!/bin/bash
# round 4, GPU call F: full / edge key blocks in ONE launch (MODE 3) - tests, A/B against the split form, bench
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4f
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py tests/test_model_gpu.py tests/test_crosscheck_gpu.py tests/test_vit_gpu.py -m gpu -x -q -s -k "fused or bwd or attention or pruned or gradients or shared_prefix or crosscheck or vit or dropout" > $OUT/pytest_attn.log 2>&1 || { tail -60 $OUT/pytest_attn.log; exit 1; }
tail -2 $OUT/pytest_attn.log; grep "dropout mask at L=10132" $OUT/pytest_attn.log || true
for d in 0.1 0.0; do
  timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 7 $d >> $OUT/attn_probe_b32.txt 2>&1
done
grep -v amdgpu.ids $OUT/attn_probe_b32.txt
timeout -k 10 600 python3 bench.py --no-cpu-baseline > $OUT/bench_handoff.json 2> $OUT/bench.err
T2S_ATTN_BWD_DQ=atomic timeout -k 10 600 python3 bench.py --no-cpu-baseline > $OUT/bench_atomic.json 2>> $OUT/bench.err
T2S_ATTN_BWD_DQ=atomic T2S_FB_SPLIT_EDGE=1 timeout -k 10 600 python3 bench.py --no-cpu-baseline > $OUT/bench_atomic_split.json 2>> $OUT/bench.err
python3 -c "
import json
for n in ('handoff','atomic','atomic_split'):
    d=json.load(open('$OUT/bench_%s.json'%n)); r=d['roofline']
    print(n, 'ms/step %.1f'%d['ms_per_step'], 'drop0 %.1f'%d['dropout_0']['ms_per_step'], 'fused_avg %.2f'%r['fused_avg_launch_ms'], 'frac %.3f'%r['frac'], 'loss', d['loss'], 'mem %.0f'%d['peak_mem_gb'])
"
