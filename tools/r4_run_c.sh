#!/bin/bash
# round 4, GPU call C: full GPU suite, bench with the dQ hand-off vs the atomic form, RCCL overlap trace, 300x200 batch sweep
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4c
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
grep "B=64 100x100" $OUT/pytest.log || true
timeout -k 10 600 python3 bench.py --no-cpu-baseline > $OUT/bench_handoff.json 2> $OUT/bench.err
T2S_ATTN_BWD_DQ=atomic timeout -k 10 600 python3 bench.py --no-cpu-baseline > $OUT/bench_atomic.json 2>> $OUT/bench.err
python3 -c "
import json
for n in ('handoff','atomic'):
    d=json.load(open('$OUT/bench_%s.json'%n)); r=d['roofline']
    print(n, 'ms/step %.1f'%d['ms_per_step'], 'drop0 %.1f'%d['dropout_0']['ms_per_step'], 'fused_avg %.2f'%r['fused_avg_launch_ms'], 'frac %.3f'%r['frac'], 'loss', d['loss'], 'mem %.0f'%d['peak_mem_gb'])
"
cd /tmp && export TMPDIR=/tmp
T2S_BENCH_FORCE_DIST=1 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/dist_trace -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dropout0 > $OUT/dist_trace.log 2>&1
python3 $REPO/tools/overlap_from_trace.py $OUT/dist_trace > $OUT/overlap.json || true
tail -c 1200 $OUT/overlap.json
rm -rf $OUT/dist_trace
cd $REPO
for b in 1 2 4 8; do
  timeout -k 10 600 python3 bench.py --batch $b --frames 300 --ocr 200 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stress_b${b}_300x200.json 2>> $OUT/stress.err
  python3 -c "import json,sys; d=json.load(open('$OUT/stress_b${b}_300x200.json')); print('B=$b', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['fused_avg_launch_ms'], d['roofline_fwd']['frac'], d['peak_mem_gb'])"
done
