#!/bin/bash
# scratch (round 5 visit A): multi-rank bench self-tests, host-path A/B HEAD vs round-3 tree, driver-form bench
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $OUT/device.txt; nproc >> $OUT/device.txt
timeout 900 python -m pytest tests/test_timed_configs_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "bench_multi_rank or c3_190" > $OUT/pytest_multirank.log 2>&1; echo "rc=$?" >> $OUT/pytest_multirank.log; tail -5 $OUT/pytest_multirank.log
for rep in 1 2; do
  (cd ab_old && timeout 300 python tools/e2e_pipeline.py) > $OUT/e2e_old_$rep.log 2>&1; grep -E "inference|global_aligner|init|getters|iterations" $OUT/e2e_old_$rep.log | head -12
  timeout 300 python tools/e2e_pipeline.py > $OUT/e2e_head_$rep.log 2>&1; grep -E "inference|global_aligner|init|getters|iterations" $OUT/e2e_head_$rep.log | head -12
done
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.log; tail -4 $OUT/bench.log; cut -c1-600 $OUT/bench.json
