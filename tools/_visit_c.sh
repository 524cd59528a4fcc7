#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $OUT/device.txt; nproc >> $OUT/device.txt
timeout 900 python -m pytest tests/test_forward_gpu.py -m gpu -q -x -s --timeout 600 -p no:cacheprovider -k "layernorm_fold and not full_size" > $OUT/pytest_fold.log 2>&1; echo "rc=$?" >> $OUT/pytest_fold.log; grep -E "folded vs|passed|failed|rc=|Error|error" $OUT/pytest_fold.log | tail -30
timeout 600 python tools/fold_probe.py --reps=3 > $OUT/fold_probe.log 2>&1; grep -E "MEAN|folded vs|Error|error" $OUT/fold_probe.log | tail
timeout 900 python -m pytest tests/test_forward_gpu.py -m gpu -q -x -s --timeout 600 -p no:cacheprovider -k "layernorm_fold_full_size" > $OUT/pytest_fold_full.log 2>&1; echo "rc=$?" >> $OUT/pytest_fold_full.log; grep -E "vs CPU|folded vs|passed|failed|rc=|Error" $OUT/pytest_fold_full.log | tail
