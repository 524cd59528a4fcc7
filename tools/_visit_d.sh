#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $OUT/device.txt; nproc >> $OUT/device.txt
timeout 600 python tools/fold_probe.py --reps=3 > $OUT/fold_probe.log 2>&1; grep -E "MEAN|folded vs|Error|error|->" $OUT/fold_probe.log | tail -40
D3R_GEMM_T384=0 timeout 600 python tools/fold_probe.py --reps=2 > $OUT/fold_probe_no384.log 2>&1; grep -E "MEAN|Error|error" $OUT/fold_probe_no384.log | tail
D3R_LN_FOLD=1 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu_fold.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_fold.log; grep -E "^FAILED|^ERROR|passed|failed|rc=" $OUT/pytest_gpu_fold.log | tail -40
