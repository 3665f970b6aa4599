#!/bin/bash
# HBM traffic of the attention kernels via PMC (separate passes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).
# usage: tools/pmc_traffic.sh <tag> [probe args...]
TAG=$1; shift
ARGS=${@:-8 10120 1.0 12 2}
OUT=/root/repo/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 /root/repo/tools/attn_probe.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 /root/repo/tools/attn_probe.py $ARGS > $OUT/write.log 2>&1
python3 /root/repo/tools/pmc_summary.py $OUT
