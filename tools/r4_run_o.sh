#!/bin/bash
# round 4, GPU call O: cycle stamps of the fused backward cross-checked against the workgroup timeline of the same launch; the
# timeline with the last wave's end recorded; the bench on the compact-ticket build
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4o
mkdir -p $OUT
set -e
cd $REPO
rm -f $OUT/stamps.txt $OUT/timeline.txt
for args in "32 0.7 0.1 1" "32 0.7 0.0 1" "32 0.7 0.1 0"; do
  echo "== fused_stamps.py $args" >> $OUT/stamps.txt
  timeout -k 10 300 python3 tools/fused_stamps.py $args 2>&1 | grep -v "amdgpu.ids" >> $OUT/stamps.txt
done
cat $OUT/stamps.txt | cut -c1-200
for args in "32 0.7 0.1 1" "32 0.7 0.1 0"; do
  echo "== fused_timeline.py $args" >> $OUT/timeline.txt
  timeout -k 10 300 python3 tools/fused_timeline.py $args 2>&1 | grep -v "amdgpu.ids" >> $OUT/timeline.txt
done
cat $OUT/timeline.txt | cut -c1-230
timeout -k 10 600 python3 bench.py --steps 6 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4o/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"].get("frac"), d.get("attn_bwd_dq"))
PY
