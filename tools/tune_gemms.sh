#!/bin/bash
# Records vitxt_gqa_amd/tuned/gemm_gfx950_b64_100x100.csv: one benchmark step with PyTorch's TunableOp timing the library's GEMM
# solutions for every shape it meets (about 12 minutes on an MI355X), then an A/B of bench.py without / with the recorded file.
# Run on the GPU box:  gpurun --timeout 1500 -- 'tools/tune_gemms.sh'
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/gpurun_out
export PYTORCH_TUNABLEOP_FILENAME=$root/gpurun_out/tunableop_results.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 python $root/bench.py --steps 1 --warmup 1 > $root/gpurun_out/tune_run.json
cp $root/gpurun_out/tunableop_results0.csv $root/vitxt_gqa_amd/tuned/gemm_gfx950_b64_100x100.csv
unset PYTORCH_TUNABLEOP_FILENAME
T2S_TUNED_GEMMS=0 python $root/bench.py --steps 5 --warmup 2 > $root/gpurun_out/tune_b0.json
python $root/bench.py --steps 5 --warmup 2 > $root/gpurun_out/tune_b1.json
grep -o '"ms_per_step": [0-9.]*' $root/gpurun_out/tune_b0.json $root/gpurun_out/tune_b1.json
