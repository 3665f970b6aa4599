#!/bin/bash
# Regenerate the judged profile artifacts of a round on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02 core'   and   ... 'bash tools/profile_round.sh r02 extra'
# writes gpurun_out/<tag>/{bench.json, kernel_stats.csv, pmc_fetch_write.txt, traffic.json}; copy them into profiles/.
# The PMC passes use --kernel-trace only (never combined with sys/hip/hsa traces) and run the program itself after "--".
TAG=${1:-r02}
PART=${2:-all}          # core = bench line + kernel stats + PMC traffic; extra = the other BASELINE configurations; all
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
set -e               # a failed GPU step ends the run: no further GPU work behind it
cd /tmp && export TMPDIR=/tmp
if [ "$PART" != "extra" ] && [ "$PART" != "busy" ]; then
# 1. the bench line (defaults: B=64 train step, with the bounded CPU baseline)
timeout -k 10 900 python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
fi
if [ "$PART" != "core" ] && [ "$PART" != "busy" ]; then
# 1b. the other BASELINE configurations: forward-only (configs[1]) and the 300 x 200 long-sequence stress (configs[4])
timeout -k 10 900 python3 $REPO/bench.py --forward-only --no-cpu-baseline > $OUT/forward_only.json 2>> $OUT/bench.err
timeout -k 10 900 python3 $REPO/bench.py --batch 2 --frames 300 --ocr 200 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stress_b2_300x200.json 2>> $OUT/bench.err
timeout -k 10 900 python3 $REPO/bench.py --no-cpu-baseline --no-dropout0 --host-inputs --compact-wire > $OUT/host_inputs_compact.json 2>> $OUT/bench.err
# 1c. measured ceilings of this box (library GEMM rates, HBM copy / read / fill rates) and the attention kernels alone
timeout -k 10 300 python3 $REPO/tools/ceilings.py > $OUT/ceilings.json 2>> $OUT/bench.err
timeout -k 10 300 python3 $REPO/tools/attn_probe.py 32 10120 0.7 12 5 0.1 > $OUT/attn_probe_b32.txt 2>> $OUT/bench.err
timeout -k 10 300 python3 $REPO/tools/attn_probe.py 32 10120 0.7 12 5 0.0 >> $OUT/attn_probe_b32.txt 2>> $OUT/bench.err
T2S_BENCH_FORCE_DIST=1 timeout -k 10 600 python3 $REPO/bench.py --no-cpu-baseline --no-dropout0 > $OUT/single_rank_rccl.json 2>> $OUT/bench.err
fi
if [ "$PART" != "extra" ] && [ "$PART" != "busy" ]; then
# 2. kernel trace + stats of the same command (fewer steps, no CPU baseline: identical GPU work per step)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dropout0 > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python3 $REPO/tools/kstats.py $OUT/stats 3 14
# 3. HBM traffic counters, one pass each
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropout0 > $OUT/fetch.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropout0 > $OUT/write.log 2>&1
python3 $REPO/tools/pmc_summary.py $OUT > $OUT/pmc_fetch_write.txt
python3 $REPO/tools/traffic_from_pmc.py $OUT > $OUT/traffic.json
cat $OUT/traffic.json
fi
if [ "$PART" == "busy" ]; then
# 4. SQ counters of the attention kernels alone (three --pmc passes each, with and without dropout) -> MFMA busy
bash $REPO/tools/pmc_attn.sh ${TAG}_drop 8 10120 0.7 12 2 0.1 > $OUT/attn_probe_b8_sq_pmc_drop.txt 2>&1
bash $REPO/tools/pmc_attn.sh ${TAG}_nodrop 8 10120 0.7 12 2 0.0 > $OUT/attn_probe_b8_sq_pmc_nodrop.txt 2>&1
python3 $REPO/tools/mfma_busy_from_pmc.py $REPO/gpurun_out/pmc_${TAG}_drop drop0.1 > $OUT/mfma_busy_drop.json
python3 $REPO/tools/mfma_busy_from_pmc.py $REPO/gpurun_out/pmc_${TAG}_nodrop drop0 > $OUT/mfma_busy_nodrop.json
cat $OUT/mfma_busy_drop.json $OUT/mfma_busy_nodrop.json
fi
