#!/bin/bash
# same-box A/B of the headline step under environment variants: tools/r4_ab_step.sh "NAME=ENV..." ...   (alternating, 2 rounds)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
B="--steps 3 --warmup 1 --no_cpu_baseline --no_knn_1m --no_f32_gemm_check ${BENCH_ARGS:-}"
for round in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    v=$(env $envs python3 $R/bench.py $B 2>$OUT/ab_$name.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f scores/s  %.1f ms  embed %.3f s' % (d['value'], d['ms_per_step'], d['stages_s']['embed_s']))")
    echo "round $round  $name  [$envs]: $v"
  done
done
