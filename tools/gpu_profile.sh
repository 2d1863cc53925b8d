#!/bin/bash
# Runs on the GPU box (via gpurun): headline bench, rocprofv3 kernel trace of the same command, and the PMC passes
# (separate runs, kernel-trace only: never combined with --sys-trace etc.) the roofline figures come from.
# The program is put directly after `--` (no env / bash -c hop: the profiler preloads before the program starts).
# Every profiled pass runs with --no_f32_gemm_check: the untimed all-fp32 cross-check step of the plain bench run must not
# contaminate kernel statistics or counters (round-3 advisor finding).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
TAG=${1:-r5}
cd /tmp && export TMPDIR=/tmp
K1M="--workload knn --knn_n 1000000 --knn_d 768 --steps 1 --warmup 0 --no_cpu_baseline"
P="--no_cpu_baseline --no_knn_1m --no_mscoco --no_f32_gemm_check"
SCAN="k_scan|k_bf16_final|k_neighbors|k_merge"
ENC="Cijk|k_attention|k_gemm_f16x3t|k_layernorm|k_vision|k_preprocess"
PART=${PART:-abc}     # a: headline + 1M kNN passes, b: the ViT-B/16 and ViT-L/14 passes, c: the full-size mscoco workload (one gpurun call each: inside the 20-minute limit)
if [[ $PART == *a* ]]; then
python3 $R/bench.py --steps 2 --warmup 1 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 $P > $OUT/prof_bench_$TAG.json 2> $OUT/prof_bench_$TAG.err || exit 2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_knn_$TAG -- python3 $R/bench.py $K1M > $OUT/prof_knn_$TAG.json 2> $OUT/prof_knn_$TAG.err || exit 3
n=3
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$((n+1)); t=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-include-regex "$SCAN" --output-format csv -d $OUT/pmc_knn_${t}_$TAG -- python3 $R/bench.py $K1M > /dev/null 2> $OUT/pmc_knn_${t}_$TAG.err || exit $n
done
# the headline step: scan kernels and the encoder kernels (hand-written GEMM, library GEMMs, attention, LayerNorm), one pass per counter set
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  n=$((n+1)); t=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-include-regex "$SCAN|$ENC" --output-format csv -d $OUT/pmc_bench_${t}_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 $P > /dev/null 2> $OUT/pmc_bench_${t}_$TAG.err || exit $n
done
fi
n=20
# configs[2] / configs[4] encoders: ViT-B/16 (L = 197) and ViT-L/14 (L = 257) at 4 000 + 500 + 500 samples
[[ $PART == *b* ]] && for a in vit-b-16 vit-l-14; do
  eb=664; [ $a = vit-l-14 ] && eb=510      # 1 022 / 1 024 row tiles of 128 token rows
  A="--arch $a --n_train 4000 --n_val 500 --n_test 500 --encoder_batch $eb $P"
  n=$((n+1))
  python3 $R/bench.py --steps 2 --warmup 1 $A > $OUT/bench_${a}_$TAG.json 2> $OUT/bench_${a}_$TAG.err || exit $n
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${a}_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 $A > /dev/null 2> $OUT/prof_${a}_$TAG.err || exit $n
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
    t=$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --kernel-include-regex "$ENC" --output-format csv -d $OUT/pmc_${a}_${t}_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 $A > /dev/null 2> $OUT/pmc_${a}_${t}_$TAG.err || exit $n
  done
done
# configs[2] at full size: ViT-B/16, 82 783 + 5 000 + 5 000 image / caption pairs, DB = 50 000-row subset (round 5)
if [[ $PART == *c* ]]; then
  C="--workload mscoco --steps 2 --warmup 1"
  python3 $R/bench.py $C > $OUT/bench_mscoco_$TAG.json 2> $OUT/bench_mscoco_$TAG.err || exit 31
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mscoco_$TAG -- python3 $R/bench.py --workload mscoco --steps 1 --warmup 1 --no_cpu_baseline > /dev/null 2> $OUT/prof_mscoco_$TAG.err || exit 32
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
    t=$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --kernel-include-regex "$ENC" --output-format csv -d $OUT/pmc_mscoco_${t}_$TAG -- python3 $R/bench.py --workload mscoco --n_train 20000 --db_limit 12000 --steps 1 --warmup 0 --no_cpu_baseline > /dev/null 2> $OUT/pmc_mscoco_${t}_$TAG.err || exit 33
  done
fi
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
ls $OUT | head -60
