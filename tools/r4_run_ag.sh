#!/bin/bash
# round 4, GPU call AG: BASELINE configs[4] (300 frames x 200 OCR) batch sweep on the round's final code
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4ag
mkdir -p $OUT
set -e
cd $REPO
rm -f $OUT/sweep.txt
for b in 1 2 4 8; do
  timeout -k 10 600 python3 bench.py --batch $b --frames 300 --ocr 200 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stress_b${b}_300x200.json 2>> $OUT/stress.err
  python3 -c "
import json
d=json.loads([l for l in open('$OUT/stress_b${b}_300x200.json') if l.startswith('{')][-1]); r=d['roofline']; f=d['roofline_fwd']
print('%3d  %8.1f  %8.3f   %.3f  %11.2f  %10.1f     %.3f  %9.1f  %8.1f  %9.1f' % ($b, d['ms_per_step'], d['value'], r['frac'], r.get('long_list_avg_launch_ms') or 0, r['achieved'], f['frac'], f['achieved'], d['peak_mem_gb'], d['dropout_0']['ms_per_step']))" >> $OUT/sweep.txt
done
cat $OUT/sweep.txt
