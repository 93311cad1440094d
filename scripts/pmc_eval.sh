#!/bin/bash
# One counter pass over a short bench run with its evaluation (GPU box), summed per scoring kernel:
# pmc_eval.sh <tag> "<counters>" [bench flags...]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; CTRS=$2; shift; shift
O=$R/gpurun_out/pmc_$TAG
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc $CTRS -d "$O" -o p --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 0 --eval-calls 1 --cpu-steps 0 --cpu-eval-users 0 --cpu-ref-users 0 --cpu-ref-eval-users 0 --no-variants "$@" > "$O/bench.json" 2> "$O/err.txt"
python3 - "$O" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
f = glob.glob(o + "/**/*counter_collection.csv", recursive=True)[0]
first = None
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    first = first or r["Counter_Name"]
    if r["Counter_Name"] == first: cnt[k] += 1
with open(o + "/summary.txt", "w") as out:
    for k, c in acc.items():
        if "score_" not in k and "correct_clicked" not in k and "topk_" not in k: continue
        n = max(cnt[k], 1)
        line = k + " launches=%d " % n + " ".join("%s=%.5g" % (a, b / n) for a, b in sorted(c.items()))
        print(line); out.write(line + "\n")
PY
rm -rf "$O"/*/ 2>/dev/null || true
