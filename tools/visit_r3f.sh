#!/bin/bash
# Round 3, visit F: the small-problem GEMM tiles (tile configuration 8 = 64 x 64 on a three-slot ring): kernel and network tests with the tile pinned, bit-identity of the batch sizes, crossover sweeps, per-launch table of the one-pair forward.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[visit] $1 at +$(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "test_linear_split_fp16 and 8-" > $OUT/pytest_cfg8.log 2>&1; echo "rc=$?" >> $OUT/pytest_cfg8.log; tail -5 $OUT/pytest_cfg8.log; stamp kernel-tests
timeout 600 python -m pytest tests/test_forward_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "(pinned_gemm_tile and 8) or (kernel_variants and 8) or batch or bit_identical" >> $OUT/pytest_cfg8.log 2>&1; echo "rc=$?" >> $OUT/pytest_cfg8.log; tail -5 $OUT/pytest_cfg8.log; stamp forward-tests
timeout 600 python tools/latency_probe.py small-tiles > $OUT/latency_small_tiles2.log 2>&1; grep -E "T64|T128|one pair" $OUT/latency_small_tiles2.log; stamp probe
