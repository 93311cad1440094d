set -e
cd $GRAFT_REPO_ROOT
for M in Fossil REBUS; do for P in 1 0; do
timeout -k 10 280 python bench.py --model $M --policy $P --steps 2 --warmup 1 --no-variants --cpu-steps 0 --cpu-eval-users 0 --no-top50 > gpurun_out/r3_hist_${M}_p$P.json 2> gpurun_out/r3_hist_${M}_p$P.err
python - <<PY
import json; d=json.load(open("gpurun_out/r3_hist_${M}_p$P.json")); print("$M policy $P", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["eval"]["auc_test"], d["eval"]["hit50_test"], flush=True)
PY
done; done
