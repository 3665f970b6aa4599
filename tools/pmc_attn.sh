#!/bin/bash
# PMC counter passes over the attention probe (separate passes; --kernel-trace only, per the gpurun rules).
# usage: tools/pmc_attn.sh <tag> [probe args...]
TAG=$1; shift
ARGS=${@:-8 10120 1.0 12 2}
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export T2S_PROBE_FORMS=shipped      # tools/attn_probe.py: only the backward forms the product runs (two-kernel + fused with the dQ hand-off)
i=0
for pm in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
          "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_WAVES" \
          "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pm --output-format csv -d $OUT/p$i -- python3 /root/repo/tools/attn_probe.py $ARGS > $OUT/p$i.log 2>&1
done
python3 /root/repo/tools/pmc_summary.py $OUT
