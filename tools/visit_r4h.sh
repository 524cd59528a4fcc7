#!/bin/bash
# Round 4, full measurement visit on the final code: GPU suite, smoke, driver-form bench line, rocprofv3 kernel trace (single-stream), FETCH / WRITE PMC
# passes, SQ counter passes, one-pair latency; summaries via tools/summarize_prof.py.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[visit] $1 at +$(( $(date +%s) - T0 )) s"; }
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log; stamp smoke
bash tools/gpu_round.sh bench prof pmc pmcsq > $OUT/round.log 2>&1; grep -E "pairs/s on|aligner [0-9]" $OUT/bench.log | tail -3; stamp bench-prof-pmc
timeout 300 python tools/latency_probe.py forward-only > $OUT/latency.log 2>&1; grep -E "eager|pairs" $OUT/latency.log | tail -6; stamp latency
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log; stamp suite
# (tools/gpu_round.sh has written prof_summary.txt / pmc_latest.json and trimmed the big per-dispatch csvs)
