#!/bin/bash
# bench.py for the three history models, GPU part only (no CPU baselines, no variants)
for M in ${@:-REBUS Fossil FossilSimple}; do
  timeout -k 10 200 python bench.py --model $M --steps 3 --warmup 1 --cpu-steps 0 --cpu-eval-users 0 --cpu-ref-users 0 --cpu-ref-eval-users 0 --no-variants --no-top50 > gpurun_out/bench_$M.json 2> gpurun_out/bench_$M.err || exit 1
done
