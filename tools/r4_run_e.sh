#!/bin/bash
# round 4, GPU call E: per-kernel times of the fused backward in both dQ forms (rocprofv3 kernel trace of tools/attn_probe.py)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4e
mkdir -p $OUT
set -e
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/probe -- python3 $REPO/tools/attn_probe.py 32 10120 0.7 12 6 0.1 > $OUT/probe.log 2>&1
cp $(find $OUT/probe -name '*kernel_stats.csv' | head -1) $OUT/attn_probe_b32_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/attn_probe_b32_kernel_stats.csv')))
for r in rows[:14]:
    print("%-110s n=%4s avg %9.3f ms total %9.1f ms" % (r['Name'][:110], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
PY
rm -rf $OUT/probe
