#!/bin/bash
# Sampler on a second stream next to the update of the previous chunk: epoch time for a few chunk lengths.
set -e
mkdir -p gpurun_out
B="--no-variants --cpu-steps 0 --cpu-eval-users 0 --cpu-ref-users 0 --cpu-ref-eval-users 0 --no-eval --steps 5 --warmup 1"
for cfg in "--overlap 0" "--overlap 1 --chunk 33554432" "--overlap 1 --chunk 16777216" "--overlap 1 --chunk 8388608" "--overlap 1 --chunk 4194304"; do
  echo "== $cfg" >> gpurun_out/overlap.log
  timeout -k 10 200 python bench.py $B $cfg 2>/dev/null >> gpurun_out/overlap.log
done
cat gpurun_out/overlap.log
