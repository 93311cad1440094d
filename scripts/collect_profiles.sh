#!/bin/bash
# Runs on the GPU box (gpurun): for one model (default BPRMF) the bench line, the rocprofv3 kernel statistics of the same
# command and three counter passes, all written under gpurun_out/prof/<tag>/.  scripts/summarize_pmc.py turns the
# counter passes into pmc_summary.json.  Every step has its own time limit; the script stops at the first failure.
# usage: collect_profiles.sh [MODEL [extra bench flags ...]]
set -e
MODEL=${1:-BPRMF}
shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof/$MODEL
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
CMD="$R/bench.py --model $MODEL --cpu-steps 0 --cpu-eval-users 0 --cpu-ref-users 0 --cpu-ref-eval-users 0 --no-variants --steps 3 --warmup 1 $*"
echo "$CMD" > "$O/command.txt"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o s -- python3 $CMD > "$O/bench_under_rocprof.json" 2> "$O/stats.err"
echo "$MODEL stats done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d "$O/pmc1" -o p --output-format csv -- python3 $CMD > /dev/null 2> "$O/pmc1.err"
echo "$MODEL pmc1 done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$O/pmc2" -o p --output-format csv -- python3 $CMD > /dev/null 2> "$O/pmc2.err"
echo "$MODEL pmc2 done"
timeout -s KILL 300 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES -d "$O/pmc3" -o p --output-format csv -- python3 $CMD > /dev/null 2> "$O/pmc3.err"
echo "$MODEL pmc3 done"
python3 "$R/scripts/summarize_pmc.py" "$O/pmc_summary.json" "$O/pmc1" "$O/pmc2" "$O/pmc3" > "$O/pmc_summary.txt"
# the counter CSVs are large: keep the summaries only
rm -rf "$O/pmc1" "$O/pmc2" "$O/pmc3"
find "$O/stats" -name '*kernel_trace*' -delete
echo "$MODEL all done"
