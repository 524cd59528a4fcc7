#!/bin/bash
# Round 4, visit E: the 2.5-unit K loop with its DMA pieces interleaved with the MFMA rows (D3R_GEMM_X2IL): parity, A/B against the burst form and fp16x3.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for il in 1 0; do D3R_GEMM_X2IL=$il timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "2p5_unit" 2>&1 | tail -2; done > $OUT/pytest_x2il_kernel.log 2>&1; cat $OUT/pytest_x2il_kernel.log
timeout 600 python -m pytest tests/test_forward_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "fp16x2f8" 2>&1 | tail -2
for m in "fp16x3 1" "fp16x2f8 1" "fp16x2f8 0" "fp16x3 1" "fp16x2f8 1" "fp16x2f8 0"; do set -- $m; echo "precision=$1 D3R_GEMM_X2IL=$2"; D3R_GEMM_X2IL=$2 timeout 200 python bench.py --precision $1 --steps 6 --warmup 2 --no-cpu-baseline --no-fast --no-profile --no-aligner --no-parity 2>&1 | grep "pairs/s on"; done > $OUT/ab_x2il.txt 2>&1; cat $OUT/ab_x2il.txt
