#!/bin/bash
# Round 4, visit K: which launches should take tile configuration 9? D3R_GEMM_T384 = 0 (never) / 1 (rule) / 2 (rule + the fp32-residual projections at K <= 1024),
# on the two-stream schedule (what is timed) and on one stream (what the per-kernel tables show).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do for t in 0 1 2; do for ss in "" "--single-stream"; do echo "D3R_GEMM_T384=$t $ss"; D3R_GEMM_T384=$t timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast --no-profile --no-aligner --no-parity $ss 2>&1 | grep "pairs/s on"; done; done; done > $OUT/ab_t384_modes.txt 2>&1; cat $OUT/ab_t384_modes.txt
