#!/bin/bash
# Round 3, visit H: the DPT head tail fused into the last 3x3 convolution (EPI_HEAD4): parity tests, A/B on the forward and the one-pair call.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[visit] $1 at +$(( $(date +%s) - T0 )) s"; }
timeout 900 python -m pytest tests/test_forward_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -s -k "fused_head or (pinned_gemm_tile and fp16x3) or full_size_fp32_pair or batch" > $OUT/pytest_head4.log 2>&1; echo "rc=$?" >> $OUT/pytest_head4.log; grep -E "fused head|512_dpt fp16x3|passed|failed|rc=" $OUT/pytest_head4.log | tail -14; stamp tests
for f in 1 0 1 0; do echo "D3R_HEAD_FUSE=$f"; D3R_HEAD_FUSE=$f timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast --no-profile --no-aligner 2>&1 | grep "pairs/s on"; done; stamp bench
for f in 1 0; do echo "D3R_HEAD_FUSE=$f"; D3R_HEAD_FUSE=$f timeout 200 python tools/latency_probe.py forward-only 2>&1 | grep "eager"; done; stamp latency
