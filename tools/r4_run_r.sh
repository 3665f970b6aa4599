#!/bin/bash
# round 4, GPU call R: keep-word masks in the two-kernel backward as well: attention tests, the five backward forms, the bench
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4r
mkdir -p $OUT
set -e
cd $REPO
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_dropout_gpu.py -m gpu -x -q > $OUT/pytest_attn.log 2>&1 || { tail -40 $OUT/pytest_attn.log; exit 1; }
tail -2 $OUT/pytest_attn.log
rm -f $OUT/attn_probe.txt
for d in 0.1 0.0; do
  echo "== L1=10120 dropout $d" >> $OUT/attn_probe.txt
  timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.7 12 10 $d 2>&1 | grep "bwd \|status\|fwd " >> $OUT/attn_probe.txt
done
echo "== L1=10120, 5 % keys (the light launches' regime), dropout 0.1" >> $OUT/attn_probe.txt
timeout -k 10 300 python3 tools/attn_probe.py 32 10120 0.05 12 10 0.1 2>&1 | grep "bwd \|status\|fwd " >> $OUT/attn_probe.txt
cat $OUT/attn_probe.txt | cut -c1-200
timeout -k 10 600 python3 bench.py --steps 6 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4r/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"].get("frac"), d["roofline"].get("fused_avg_launch_ms"), d.get("dropout_0"))
PY
