#!/bin/bash
# repeat the end-to-end CLI test in fresh processes with the finite-check diagnostic (flake hunt)
mkdir -p gpurun_out; : > gpurun_out/cli_flake.txt
for i in $(seq 1 ${1:-12}); do
  LEMON_DEBUG_FINITE=1 timeout -k 10 200 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > gpurun_out/cli_flake_$i.log 2>&1
  echo "run $i rc=$? $(tail -1 gpurun_out/cli_flake_$i.log)" >> gpurun_out/cli_flake.txt
  grep -h "FloatingPointError\|non-finite" gpurun_out/cli_flake_$i.log | head -3 >> gpurun_out/cli_flake.txt
done
cat gpurun_out/cli_flake.txt
