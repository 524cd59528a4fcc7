#!/bin/bash
# Round 4, visit C: what would a 2.5-unit arithmetic buy? (perf proxy: the fp16 + fp8 K loop with the MFMA mix of hi.hi + hi.w_lo on f16 and a_lo.w_hi
# on e4m3 -- results invalid, timing only) next to fp16x3 and fp16f8 on the same box.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for m in "fp16x3 0" "fp16f8 0" "fp16f8 1" "fp16x3 0" "fp16f8 1" "fp16f8 0"; do set -- $m; echo "precision=$1 D3R_F8_PROXY=$2"; D3R_F8_PROXY=$2 timeout 200 python bench.py --precision $1 --steps 6 --warmup 2 --no-cpu-baseline --no-fast --no-profile --no-aligner --no-parity 2>&1 | grep "pairs/s on"; done > $OUT/ab_f8_proxy.txt 2>&1; cat $OUT/ab_f8_proxy.txt
