#!/bin/bash
# update-kernel time of the history models for other numbers of tuples per walk and engine options:
# bench_hist_g.sh "<MODEL>:<G>[:name=value[,name=value...]]" ...
B="--steps 3 --warmup 1 --cpu-steps 0 --cpu-eval-users 0 --cpu-ref-users 0 --cpu-ref-eval-users 0 --no-variants --no-top50 --no-eval"
for mg in "$@"; do
  IFS=: read M G O <<< "$mg"
  OPTS=""; for kv in ${O//,/ }; do OPTS="$OPTS --opt $kv"; done
  T=${mg//[:=,]/_}
  timeout -k 10 200 python bench.py --model $M --user-batch $G $OPTS $B > gpurun_out/bg_$T.json 2> gpurun_out/bg_$T.err || exit 1
  python3 -c "
import json;j=json.loads(open('gpurun_out/bg_$T.json').read().strip().splitlines()[-1]);print('$mg',round(j['value']/1e6,1),'M triples/s',j['ms_per_step'],'ms')"
done
