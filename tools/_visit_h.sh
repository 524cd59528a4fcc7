#!/bin/bash
# full visit: suite, smoke, driver-form bench, rocprofv3 kernel trace, PMC passes
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export D3R_VISIT=profiles/r05_h
bash tools/gpu_round.sh testsall bench prof pmc pmcsq 2>&1 | tail -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
