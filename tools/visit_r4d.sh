#!/bin/bash
# Round 4, visit D: the 2.5-unit mode (D3R_DTYPE_F16X2F8): kernel-level parity, forward parity on tiny and full-size models, six weight seeds against
# the CPU oracle, throughput next to fp16x3 / fp16f8 on the same box.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[visit] $1 at +$(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "2p5_unit" > $OUT/pytest_x2f8_kernel.log 2>&1; echo "rc=$?" >> $OUT/pytest_x2f8_kernel.log; tail -12 $OUT/pytest_x2f8_kernel.log; stamp kernel-tests
timeout 900 python -m pytest tests/test_forward_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -s -k "fp16x2f8 or full_size_fp32_pair" > $OUT/pytest_x2f8_forward.log 2>&1; echo "rc=$?" >> $OUT/pytest_x2f8_forward.log; grep -E "fp16x2f8\]|512_dpt fp16|passed|failed|rc=" $OUT/pytest_x2f8_forward.log | tail -30; stamp forward-tests
timeout 900 python -m pytest tests/test_timed_configs_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -s -k "weight_seeds" > $OUT/pytest_x2f8_seeds.log 2>&1; echo "rc=$?" >> $OUT/pytest_x2f8_seeds.log; grep -E "vs CPU oracle|passed|failed|rc=" $OUT/pytest_x2f8_seeds.log | tail -24; stamp seeds
for m in fp16x3 fp16x2f8 fp16f8 fp16x3 fp16x2f8 fp16f8; do echo "precision=$m"; timeout 200 python bench.py --precision $m --steps 6 --warmup 2 --no-cpu-baseline --no-fast --no-profile --no-aligner --no-parity 2>&1 | grep "pairs/s on"; done > $OUT/ab_x2f8.txt 2>&1; cat $OUT/ab_x2f8.txt; stamp ab
