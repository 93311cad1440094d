#!/bin/bash
# HBM-side traffic and L2 hit rate of the evaluation kernels (GPU box).  usage: bash scripts/pmc_eval_mem.sh <tag> [exp_eval flags...]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/pmc_evalmem_$TAG
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
ARGS="--users 1000000 --items 200000 --clicks 100000000 --reps 1 $*"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$O/a" -o p --output-format csv -- python3 "$R/scripts/exp_eval.py" $ARGS > "$O/a.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d "$O/b" -o p --output-format csv -- python3 "$R/scripts/exp_eval.py" $ARGS > "$O/b.log" 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(o + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open(o + "/summary.txt", "w") as out:
    for k, c in acc.items():
        if "score_" not in k: continue
        line = k + " " + " ".join("%s=%.4g" % (a, b / max(cnt[k][a], 1)) for a, b in sorted(c.items()))
        print(line); out.write(line + "\n")
PY
rm -rf "$O"/a "$O"/b 2>/dev/null || true
