#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/ab_probe.py D3R_ATTN_NW=4,8d,8e > $OUT/ab_attn_nw.log 2>&1; grep -E "MEAN|outputs|==" $OUT/ab_attn_nw.log
timeout 600 python tools/e2e_pipeline.py > $OUT/e2e_full.log 2>&1; grep -E "inference|global_aligner" $OUT/e2e_full.log
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "attention" > $OUT/pytest_attn.log 2>&1; tail -2 $OUT/pytest_attn.log
