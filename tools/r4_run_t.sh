#!/bin/bash
# round 4, GPU call T: short key lists (pos / neg passes, compact K | V) through the fused hand-off form: whole GPU suite, then the bench
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4t
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
timeout -k 10 600 python3 bench.py --steps 6 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4t/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"].get("frac"), d["roofline"].get("fused_avg_launch_ms"), d.get("dropout_0", {}).get("ms_per_step"))
PY
