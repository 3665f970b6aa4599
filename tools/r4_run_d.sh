#!/bin/bash
# round 4, GPU call D: LDS-DMA staging of the fused backward (A/B against the register-staged build), hand-off vs atomics, the
# 300 x 200 geometry sweep, the wgrad layout probe, bench
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4d
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "fused or bwd or attention or pruned or gradients or shared_prefix" > $OUT/pytest_attn.log 2>&1 || { tail -60 $OUT/pytest_attn.log; exit 1; }
tail -2 $OUT/pytest_attn.log
for d in 0.1 0.0; do
  echo "== LDS-DMA staging (shipped build), dropout $d" >> $OUT/attn_probe_b32.txt
  timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 7 $d >> $OUT/attn_probe_b32.txt 2>&1
  echo "== register staging (-DFB_DMA=0), dropout $d" >> $OUT/attn_probe_b32.txt
  T2S_HIP_LIB=$REPO/tools/ablate/_build/libt2s_fb_dma0.so timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 7 $d >> $OUT/attn_probe_b32.txt 2>&1
done
grep -v amdgpu.ids $OUT/attn_probe_b32.txt
timeout -k 10 600 python3 tools/stress_sweep.py > $OUT/stress_sweep.txt 2>&1 || { tail -30 $OUT/stress_sweep.txt; exit 1; }
grep -v amdgpu.ids $OUT/stress_sweep.txt
timeout -k 10 300 python3 tools/wgrad_layout_probe.py > $OUT/wgrad_layout.txt 2>&1
grep -v amdgpu.ids $OUT/wgrad_layout.txt
timeout -k 10 600 python3 bench.py --no-cpu-baseline > $OUT/bench_handoff.json 2> $OUT/bench.err
T2S_ATTN_BWD_DQ=atomic timeout -k 10 600 python3 bench.py --no-cpu-baseline > $OUT/bench_atomic.json 2>> $OUT/bench.err
python3 -c "
import json
for n in ('handoff','atomic'):
    d=json.load(open('$OUT/bench_%s.json'%n)); r=d['roofline']
    print(n, 'ms/step %.1f'%d['ms_per_step'], 'drop0 %.1f'%d['dropout_0']['ms_per_step'], 'fused_avg %.2f'%r['fused_avg_launch_ms'], 'frac %.3f'%r['frac'], 'loss', d['loss'], 'mem %.0f'%d['peak_mem_gb'])
"
