#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/fold_probe.py --reps=3 > $OUT/fold_probe.log 2>&1; grep -E "MEAN|folded vs|Error|error|cfg7|other " $OUT/fold_probe.log | tail -12
D3R_GEMM_R=0 timeout 600 python tools/fold_probe.py --reps=3 > $OUT/fold_probe_r0.log 2>&1; grep -E "MEAN|Error|error|N= 1024 K= 1024" $OUT/fold_probe_r0.log | tail
timeout 900 python -m pytest tests/test_forward_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "kernel_variants or layernorm_fold" > $OUT/pytest_variants.log 2>&1; echo "rc=$?" >> $OUT/pytest_variants.log; tail -4 $OUT/pytest_variants.log
