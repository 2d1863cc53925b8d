#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1)); d=$OUT/attnpmc_$i
  rocprofv3 --pmc $c --kernel-include-regex "k_attention" --output-format csv -d $d -- python3 $R/tools/attn_time.py > $d.txt 2>&1 || { echo "pass failed"; tail -3 $d.txt; }
done
TOP=12 python3 $R/tools/pmc_summary.py $OUT/attnpmc_1 $OUT/attnpmc_2
