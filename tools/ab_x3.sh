#!/bin/bash
# A/B of the split-fp16 GEMM variants on the whole forward (B = 32, single process per arm): software-pipelined vs plain K loop,
# LDS-staged vs direct epilogue. Usage (GPU box): bash tools/ab_x3.sh
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
ARMS=("$@"); [ ${#ARMS[@]} -eq 0 ] && ARMS=("1 0" "0 0" "1 1" "0 1")
for arm in "${ARMS[@]}"; do
  set -- $arm
  echo "== D3R_GEMM_X3SW=$1 D3R_GEMM_NOWIDE=$2" >> gpurun_out/ab_x3.log
  D3R_GEMM_X3SW=$1 D3R_GEMM_NOWIDE=$2 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast --no-aligner 2>&1 >/dev/null | grep -E "pairs/s|per-kernel|linear cfg|conv cfg" | head -24 >> gpurun_out/ab_x3.log
done
grep -E "==|pairs/s" gpurun_out/ab_x3.log
