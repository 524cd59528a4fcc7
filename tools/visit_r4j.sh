#!/bin/bash
# Round 4, visit J: tile configuration 9 (384 x 192) as chosen by the heuristic: kernel-level and forward parity with the tile pinned, the 32-pair batch against
# one-pair calls, A/B on the forward.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_forward_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "(test_linear_split_fp16 and 9) or (pinned_gemm_tile and 9) or (kernel_variants and 9)" > $OUT/pytest_cfg9.log 2>&1; echo "rc=$?" >> $OUT/pytest_cfg9.log; tail -4 $OUT/pytest_cfg9.log
timeout 900 python -m pytest tests/test_timed_configs_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -k "c2_batch or c3_190" > $OUT/pytest_cfg9_c2.log 2>&1; echo "rc=$?" >> $OUT/pytest_cfg9_c2.log; tail -3 $OUT/pytest_cfg9_c2.log
for t in 0 1 0 1; do echo "D3R_GEMM_T384=$t"; D3R_GEMM_T384=$t timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast --no-profile --no-aligner --no-parity 2>&1 | grep "pairs/s on"; done > $OUT/ab_t384_rule.txt 2>&1; cat $OUT/ab_t384_rule.txt
