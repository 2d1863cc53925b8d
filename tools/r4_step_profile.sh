#!/bin/bash
# headline step: value, kernel trace, MFMA-busy / clock of the encoder kernels (program directly after `--`)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-r4a}; shift
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
B="--steps 2 --warmup 1 --no_cpu_baseline --no_knn_1m --no_f32_gemm_check $*"
python3 $R/bench.py $B > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err || exit 1
python3 -c "import json,sys; d=json.load(open('$OUT/bench_$TAG.json')); print('value', d['value'], 'ms', d['ms_per_step'], 'enc frac', d['encoder']['frac'], d['stages_s'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $R/bench.py $B > /dev/null 2> $OUT/prof_$TAG.err || exit 2
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-include-regex "Cijk|k_attention|k_gemm_f16x3t|k_layernorm" --output-format csv -d $OUT/pmc_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_knn_1m --no_f32_gemm_check $* > /dev/null 2> $OUT/pmc_$TAG.err || exit 3
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); head -25 $f | cut -c1-200
TOP=16 python3 $R/tools/pmc_summary.py $OUT/pmc_$TAG
