#!/bin/bash
# Round 4, visit N: the GPU suite on the alternative routes of the round's switches (register-staged attention, no 384 x 192 tile), from a fresh full build.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
D3R_ATTN_DMA=0 D3R_GEMM_T384=0 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -k "not bench_multi_rank and not c5_100 and not c3_190" > $OUT/pytest_gpu_alt.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_alt.log; tail -3 $OUT/pytest_gpu_alt.log
