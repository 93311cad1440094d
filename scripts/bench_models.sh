#!/bin/bash
# the bench line (GPU part only: no CPU baselines, no variants) of any list of models: bench_models.sh "<MODEL>[:extra bench flags]" ...
for mf in "$@"; do
  M=${mf%%:*}; F=""; [ "$mf" != "$M" ] && F=${mf#*:}
  T=${mf//[: =-]/_}
  timeout -k 10 300 python bench.py --model $M $F --steps 3 --warmup 1 --cpu-steps 0 --cpu-eval-users 0 --cpu-ref-users 0 --cpu-ref-eval-users 0 --no-variants > gpurun_out/bm_$T.json 2> gpurun_out/bm_$T.err || exit 1
  python3 -c "
import json;j=json.loads(open('gpurun_out/bm_$T.json').read().strip().splitlines()[-1]);r=j['roofline'];print('$mf',round(j['value']/1e9,3),'G triples/s',j['ms_per_step'],'ms frac',r['frac'],'atomic',r.get('atomic',{}).get('frac'),'eval',round(j['eval']['users_per_sec']/1e6,2),'M users/s top50 ms',j['eval'].get('top50',{}).get('device_ms'))"
done
