#!/bin/bash
# round 4, GPU call H: bucket launch points, then the 'extra' part of the profile round
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04
mkdir -p $OUT
set -e
cd $REPO
if [ ! -s $OUT/bucket_launch_points.json ]; then
timeout -k 10 600 python3 tools/bucket_timing.py 3 > $OUT/bucket_launch_points.json 2> $OUT/bucket.err || { tail -20 $OUT/bucket.err; exit 1; }
fi
python3 -c "
import json; t=open('$OUT/bucket_launch_points.json').read(); d=json.loads(t[t.index('{'):]); s=d['steps'][-1]; print('backward %.1f ms'%s['backward_ms']); [print(b) for b in s['buckets']]; print('hidden', d['buckets_with_more_than_2ms_of_backward_behind_them'], 'of', d['n_buckets'])"
bash tools/profile_round.sh r04 extra
