#!/bin/bash
# round 4, GPU call AB: the whole GPU suite and the bench on the round's last build
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4ab
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4ab/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"].get("frac"), d["roofline"].get("long_list_avg_launch_ms"), d.get("dropout_0", {}).get("ms_per_step"), d["cpu_baseline"]["value"])
PY
