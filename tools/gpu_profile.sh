#!/bin/bash
# Runs on the GPU box (via gpurun): headline bench, rocprofv3 kernel trace of the same command, and the PMC passes
# (separate runs, kernel-trace only: never combined with --sys-trace etc.) the roofline `traffic` figures come from.
# The program is put directly after `--` (no env / bash -c hop: the profiler preloads before the program starts).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp
K1M="--workload knn --knn_n 1000000 --knn_d 768 --steps 1 --warmup 0 --no_cpu_baseline"
python3 $R/bench.py --steps 2 --warmup 1 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_knn_1m > $OUT/prof_bench_$TAG.json 2> $OUT/prof_bench_$TAG.err || exit 2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_knn_$TAG -- python3 $R/bench.py $K1M > $OUT/prof_knn_$TAG.json 2> $OUT/prof_knn_$TAG.err || exit 3
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_scan|k_bf16_final|k_neighbors|k_merge" --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $R/bench.py $K1M > $OUT/pmc_fetch_$TAG.json 2> $OUT/pmc_fetch_$TAG.err || exit 4
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_scan|k_bf16_final|k_neighbors|k_merge" --output-format csv -d $OUT/pmc_write_$TAG -- python3 $R/bench.py $K1M > $OUT/pmc_write_$TAG.json 2> $OUT/pmc_write_$TAG.err || exit 5
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "k_scan|k_bf16_final|k_neighbors|k_merge" --output-format csv -d $OUT/pmc_mfma_$TAG -- python3 $R/bench.py $K1M > $OUT/pmc_mfma_$TAG.json 2> $OUT/pmc_mfma_$TAG.err || exit 6
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_scan|k_bf16_final|k_neighbors|k_merge" --output-format csv -d $OUT/pmc_bench_fetch_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_knn_1m > /dev/null 2> $OUT/pmc_bench_fetch_$TAG.err || exit 7
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_scan|k_bf16_final|k_neighbors|k_merge" --output-format csv -d $OUT/pmc_bench_write_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_knn_1m > /dev/null 2> $OUT/pmc_bench_write_$TAG.err || exit 8
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "k_scan|k_bf16_final|k_neighbors|k_merge|k_attention|k_layernorm|k_vision" --output-format csv -d $OUT/pmc_bench_mfma_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_knn_1m > /dev/null 2> $OUT/pmc_bench_mfma_$TAG.err || exit 9
# MFMA utilisation of the tower GEMMs (hipBLASLt `Cijk_...` kernels) and the attention kernels on the FULL headline shape
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "Cijk|k_attention|k_gemm_f16x3t" --output-format csv -d $OUT/pmc_encoder_mfma_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_knn_1m > /dev/null 2> $OUT/pmc_encoder_mfma_$TAG.err || exit 10
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
ls $OUT/prof_bench_$TAG/*/ | head -20
