#!/bin/bash
# round 4, GPU call A: the GPU test suite (with the new B=64 test), the single-rank RCCL overlap trace, the 300x200 batch sweep
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4a
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
T2S_BENCH_FORCE_DIST=1 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/dist_trace -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dropout0 > $OUT/dist_trace.log 2>&1
python3 $REPO/tools/overlap_from_trace.py $OUT/dist_trace > $OUT/overlap.json || true
tail -c 1500 $OUT/overlap.json
rm -rf $OUT/dist_trace
for b in 1 2 4 8; do
  timeout -k 10 600 python3 $REPO/bench.py --batch $b --frames 300 --ocr 200 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stress_b${b}_300x200.json 2>> $OUT/stress.err
  python3 -c "import json,sys; d=json.load(open('$OUT/stress_b${b}_300x200.json')); print('B=$b', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['fused_avg_launch_ms'], d['roofline_fwd']['frac'], d['peak_mem_gb'])"
done
