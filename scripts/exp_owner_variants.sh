set -e
B="timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-steps 0 --cpu-eval-users 0 --cpu-ref-users 0 --cpu-ref-eval-users 0 --no-eval --no-variants"
run() { name=$1; shift; $B "$@" > gpurun_out/x_$name.json 2> gpurun_out/x_$name.err; python -c "
import json,sys; d=json.load(open('gpurun_out/x_$name.json')); r=d['roofline']; print('$name', d['ms_per_step'], r['avg_launch_ms'], r['partition_ms_per_launch'], r['sampler_ms_per_launch'], r.get('owner_slices'), r.get('negatives_resident'), r.get('atomic'))"; }
run owner_default
run owner_tile4096 --part-tile 4096
run owner_flush_inf --owner-flush 60000
run owner_flush16 --owner-flush 16
run owner_plainpos --owner 1 --atomic-mask 4
