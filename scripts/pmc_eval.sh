#!/bin/bash
# Issue / wait breakdown of the evaluation kernels: rocprofv3 counter passes over scripts/exp_eval.py (GPU box).
# usage: bash scripts/pmc_eval.sh <tag> [exp_eval flags...]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/pmc_eval_$TAG
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
ARGS="--users 1000000 --items 200000 --clicks 100000000 --reps 1 $*"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d "$O/a" -o p --output-format csv -- python3 "$R/scripts/exp_eval.py" $ARGS > "$O/a.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC -d "$O/b" -o p --output-format csv -- python3 "$R/scripts/exp_eval.py" $ARGS > "$O/b.log" 2>&1
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
        if "score_" not in k and "correct_" not in k: continue
        line = k + " " + " ".join("%s=%.4g" % (a, b / max(cnt[k][a], 1)) for a, b in sorted(c.items()))
        print(line); out.write(line + "\n")
PY
rm -rf "$O"/a "$O"/b 2>/dev/null || true
