# history kernels at the full shape: plain stores instead of atomics, more tuples per walk (what bounds the kernel?)
set -e
cd $GRAFT_REPO_ROOT
for M in Fossil REBUS; do for OPT in "" "--policy 2" "--user-batch 4" "--user-batch 1"; do
timeout -k 10 280 python bench.py --model $M $OPT --steps 2 --warmup 1 --no-variants --cpu-steps 0 --cpu-eval-users 0 --no-top50 --no-eval > gpurun_out/hm.json 2> gpurun_out/hm.err
python - <<PY
import json; d=json.load(open("gpurun_out/hm.json")); print("$M [$OPT]", d["value"], d["ms_per_step"], d["roofline"]["frac"], flush=True)
PY
done; done
