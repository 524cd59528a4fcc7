#!/bin/bash
# Generic A/B over environment settings on the whole forward (B = 32): bash tools/ab_env.sh "VAR=1 VAR2=0" "VAR=0" ...
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
: > gpurun_out/ab_env.log
for arm in "$@"; do
  echo "== $arm" >> gpurun_out/ab_env.log
  env $arm timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast --no-aligner 2>&1 >/dev/null | grep -E "pairs/s|per-kernel|linear cfg|conv cfg|attention|other" | head -18 >> gpurun_out/ab_env.log
done
grep -E "==|pairs/s" gpurun_out/ab_env.log
