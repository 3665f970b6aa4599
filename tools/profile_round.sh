#!/bin/bash
# Regenerate the judged profile artifacts of a round on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1800 -- 'bash tools/profile_round.sh r01'
# writes gpurun_out/<tag>/{bench.json, kernel_stats.csv, pmc_fetch_write.txt, traffic.json}; copy them into profiles/.
# The PMC passes use --kernel-trace only (never combined with sys/hip/hsa traces) and run the program itself after "--".
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench line (defaults: B=64 train step, with the bounded CPU baseline)
python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
# 1b. the other BASELINE configurations: forward-only (configs[1]) and the 300 x 200 long-sequence stress (configs[4])
python3 $REPO/bench.py --forward-only --no-cpu-baseline > $OUT/forward_only.json 2>> $OUT/bench.err
python3 $REPO/bench.py --batch 2 --frames 300 --ocr 200 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stress_b2_300x200.json 2>> $OUT/bench.err
python3 $REPO/bench.py --no-cpu-baseline --no-dropout0 --host-inputs --compact-wire > $OUT/host_inputs_compact.json 2>> $OUT/bench.err
# 2. kernel trace + stats of the same command (fewer steps, no CPU baseline: identical GPU work per step)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dropout0 > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python3 $REPO/tools/kstats.py $OUT/stats 3 14
# 3. HBM traffic counters, one pass each
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropout0 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropout0 > $OUT/write.log 2>&1
python3 $REPO/tools/pmc_summary.py $OUT > $OUT/pmc_fetch_write.txt
python3 - "$OUT" <<'EOF'
import json, re, sys
out = sys.argv[1]
txt = open(out + "/pmc_fetch_write.txt").read()
vals = {}
sec = (re.search(r"attn_fwd_bf16_kernel<true, 2, true, false>[^\n]*\n((?:\s+\w+\s+n=[^\n]*\n)+)", txt)       # dropout on (bench default)
       or re.search(r"attn_fwd_bf16_kernel<true, 2, false, false>[^\n]*\n((?:\s+\w+\s+n=[^\n]*\n)+)", txt))
for m in re.finditer(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+ mean=([0-9.e+]+)", sec.group(1) if sec else ""):
    vals[m.group(1)] = float(m.group(2))
if len(vals) == 2:
    b = 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024
    json.dump({"_comment": "HBM bytes per launch of the 64-row-per-wave attn_fwd_bf16_kernel (steady-state launch) in a B=64 train step: rocprofv3 PMC, separate "
               "FETCH_SIZE / WRITE_SIZE passes over `bench.py --steps 1 --warmup 1`; FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md "
               "(gfx950 reports half of a wide coalesced stream), WRITE_SIZE (KB) as is: 2*%.4g*1024 + %.4g*1024" % (vals["FETCH_SIZE"], vals["WRITE_SIZE"]),
               "attn_fwd_bf16_kernel": b}, open(out + "/traffic.json", "w"), indent=1)
    print("traffic bytes/launch", b)
else:
    print("traffic: kernel not found in", out + "/pmc_fetch_write.txt")
EOF
