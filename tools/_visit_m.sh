#!/bin/bash
# full visit on the round's code: suite, smoke, driver-form bench, rocprofv3 kernel trace, PMC passes, workloads c3 / c5, end to end
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export D3R_VISIT=profiles/r05_m
bash tools/gpu_round.sh testsall bench prof pmc pmcsq 2>&1 | tail -5
OUT=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py --workload c3 --steps 3 --warmup 1 > $OUT/bench_c3.json 2> $OUT/bench_c3.log; cut -c1-200 $OUT/bench_c3.json
timeout 900 python bench.py --workload c5 --steps 2 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.log; cut -c1-300 $OUT/bench_c5.json
timeout 600 python tools/e2e_pipeline.py > $OUT/e2e_full.log 2>&1; grep -E "inference|global_aligner" $OUT/e2e_full.log
D3R_LN_FOLD=0 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -k "not bench_multi_rank and not c5_100 and not c3_190" > $OUT/pytest_gpu_nofold.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_nofold.log; tail -3 $OUT/pytest_gpu_nofold.log
