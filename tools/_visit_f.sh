#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
D3R_LN_FOLD=1 timeout 300 python tools/ab_probe.py D3R_GEMM_X3NT=0,1 > $OUT/ab_x3nt.log 2>&1; grep -E "MEAN|outputs|==" $OUT/ab_x3nt.log
