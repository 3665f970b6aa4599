#!/bin/bash
# SQ counter passes over the own GEMM kernels (separate passes; --kernel-trace only).  usage: tools/pmc_gemm.sh <tag>
TAG=$1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_gemm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pm in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
          "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_WAVES" \
          "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pm --output-format csv -d $OUT/p$i -- python3 $REPO/tools/gemm_pmc_probe.py > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PYEOF
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if not ("gemm_" in k or "Cijk" in k):
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    if "GRBM_GUI_ACTIVE" not in m or "SQ_VALU_MFMA_BUSY_CYCLES" not in m:
        continue
    simd_cycles = m["GRBM_GUI_ACTIVE"] / 8 * 1024
    wc = m.get("SQ_WAVE_CYCLES", 0) * 4            # quad-cycles -> cycles, summed over waves
    print(k[:100])
    print("   mfma_busy %.3f   per wave-cycle: wait_any %.3f  wait_inst_any %.3f  active_inst_any %.3f   lds conflict frac %.4f   insts per MFMA: valu %.2f lds %.2f salu %.2f vmem %.3f"
          % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
             m.get("SQ_ACTIVE_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1),
             m.get("SQ_INSTS_VALU", 0) / max(m.get("SQ_INSTS_MFMA", 1), 1), m.get("SQ_INSTS_LDS", 0) / max(m.get("SQ_INSTS_MFMA", 1), 1), m.get("SQ_INSTS_SALU", 0) / max(m.get("SQ_INSTS_MFMA", 1), 1),
             m.get("SQ_INSTS_VMEM", 0) / max(m.get("SQ_INSTS_MFMA", 1), 1)))
PYEOF
