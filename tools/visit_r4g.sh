#!/bin/bash
# Round 4, visit G: the 2.5-unit mode on the outlier-weight sets and the error-distribution seeds; full bench line with the fast modes.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_forward_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -s -k "outlier_channels" > $OUT/pytest_x2f8_outlier.log 2>&1; echo "rc=$?" >> $OUT/pytest_x2f8_outlier.log; grep -E "sharp attention|passed|failed|rc=" $OUT/pytest_x2f8_outlier.log | tail -20
timeout 900 python bench.py --no-cpu-baseline --no-aligner > $OUT/bench_fast.json 2> $OUT/bench_fast.log; grep -E "mode |pairs/s on" $OUT/bench_fast.log
