# hot-row threshold of the hybrid scatter for Fossil: accuracy against the sequential restatement on the 60k-user corpus,
# time per epoch at the full shape
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u scripts/exp_seq_vs_default.py --models Fossil --epochs 2 --evals-per-epoch 4 --oracle-seeds 0 --engine-seeds 3 --only-variants 1 --with-atomics 0 \
  --variants "hot_x2:hot_count=11484;hot_x4:hot_count=22968;hot_x8:hot_count=45936;hot_x16:hot_count=91872;hot_x64:hot_count=367000" > gpurun_out/hot_thr_acc.log 2>&1
grep -A20 "steps per epoch" gpurun_out/hot_thr_acc.log
for H in 191000 382000 764000 1528000 6000000; do
timeout -k 10 280 python bench.py --model Fossil --hot-count $H --steps 2 --warmup 1 --no-variants --cpu-steps 0 --cpu-eval-users 0 --no-top50 --no-eval > gpurun_out/hot_thr_$H.json 2> gpurun_out/hot_thr_$H.err
python - <<PY
import json; d=json.load(open("gpurun_out/hot_thr_$H.json")); print("Fossil C2 hot_count $H", d["value"], d["ms_per_step"], flush=True)
PY
done
