#!/bin/bash
# PMC passes over the GEMM micro A/B (.variants/gemm_ab*): HBM-side traffic and L2 hit rate of the hand-written kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; BIN=${1:-gemm_ab}; TAG=${2:-gemmpmc}
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  d=$OUT/${TAG}_$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-include-regex "k_gemm_f16x3t" --output-format csv -d $d -- $R/.variants/$BIN 1 2 > $d.txt 2>&1 || { echo "pass $c failed"; tail -3 $d.txt; }
done
TOP=40 python3 $R/tools/pmc_summary.py $OUT/${TAG}_FETCH_SIZE $OUT/${TAG}_WRITE_SIZE $OUT/${TAG}_TCC_HIT_sum $OUT/${TAG}_SQ_VALU_MFMA_BUSY_CYCLES
