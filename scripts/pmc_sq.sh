#!/bin/bash
# Issue / wait breakdown of the update kernel: one rocprofv3 counter pass over a short bench run (GPU box).
# usage: bash scripts/pmc_sq.sh <tag> [bench flags...]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/pmc_$TAG
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d "$O" -o p --output-format csv -- python3 "$R/bench.py" --steps 2 --warmup 1 --cpu-steps 0 --cpu-eval-users 0 --cpu-ref-users 0 --cpu-ref-eval-users 0 --no-eval --no-variants "$@" > "$O/bench.json" 2> "$O/err.txt"
python3 - "$O" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
f = glob.glob(o + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
with open(o + "/summary.txt", "w") as out:
    for k, c in acc.items():
        if "sgd_" not in k and "sampler" not in k and "part_" not in k: continue
        n = max(cnt[k], 1)
        line = k + " launches=%d " % n + " ".join("%s=%.4g" % (a, b / n) for a, b in sorted(c.items()))
        print(line); out.write(line + "\n")
PY
rm -rf "$O"/*/ 2>/dev/null || true
