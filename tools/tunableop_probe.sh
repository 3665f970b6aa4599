#!/bin/bash
# Does PyTorch's TunableOp (per-shape search over the rocBLAS / hipBLASLt solutions) find faster library GEMMs for the step's plain NT shapes?
#   1. the step with tuning ON (bench, few steps): every library GEMM shape is benchmarked once, results -> $OUT/tunableop_results0.csv
#   2. the step with the file loaded, tuning OFF, against the default heuristic, same box, interleaved
OUT=gpurun_out/${1:-tunable}; mkdir -p $OUT
export PYTORCH_TUNABLEOP_FILENAME=$PWD/$OUT/tunableop_results.csv
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=60 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=10 PYTORCH_TUNABLEOP_VERBOSE=0 \
  timeout -k 10 900 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropout0 > $OUT/tune.json 2> $OUT/tune.err || { tail -20 $OUT/tune.err; exit 1; }
ls -la $OUT/; wc -l $OUT/tunableop_results*.csv; head -30 $OUT/tunableop_results0.csv
for rep in 1 2; do
  PYTORCH_TUNABLEOP_ENABLED=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-dropout0 > $OUT/default_$rep.json 2> $OUT/default_$rep.err || exit 1
  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-dropout0 > $OUT/tuned_$rep.json 2> $OUT/tuned_$rep.err || exit 1
done
python3 - <<PY
import json
for n in ("default_1","tuned_1","default_2","tuned_2"):
    d=json.loads(open("$OUT/%s.json"%n).read().strip().splitlines()[-1])
    print("%-10s %8.2f ms/step %7.2f samples/s  loss %s" % (n, d["ms_per_step"], d["value"], d.get("loss")))
PY
