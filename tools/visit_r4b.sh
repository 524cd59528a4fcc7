#!/bin/bash
# Round 4, visit B: DMA-staged split-fp16 attention (parity, bit-identity with the register-staged kernel, A/B), c5 workload and self-tests after the
# consistent-scene change.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[visit] $1 at +$(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "attention" > $OUT/pytest_attn_dma.log 2>&1; echo "rc=$?" >> $OUT/pytest_attn_dma.log; tail -6 $OUT/pytest_attn_dma.log; stamp attention-tests
timeout 300 python tools/gpu_probe.py attndma > $OUT/attndma.log 2>&1; tail -10 $OUT/attndma.log; stamp attndma
for d in 0 1 0 1; do echo "D3R_ATTN_DMA=$d"; D3R_ATTN_DMA=$d timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast --no-profile --no-aligner --no-parity 2>&1 | grep "pairs/s on"; done > $OUT/ab_attn_dma.txt 2>&1; cat $OUT/ab_attn_dma.txt; stamp ab
D3R_ATTN_DMA=1 timeout 600 python -m pytest tests/test_forward_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "full_size_fp32_pair or (matches_oracle and fp16x3) or pairs_of_two_image_sizes" > $OUT/pytest_fwd_dma.log 2>&1; echo "rc=$?" >> $OUT/pytest_fwd_dma.log; tail -4 $OUT/pytest_fwd_dma.log; stamp forward-tests-dma
timeout 900 python -m pytest tests/test_timed_configs_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -k "c5" > $OUT/pytest_c5.log 2>&1; echo "rc=$?" >> $OUT/pytest_c5.log; tail -6 $OUT/pytest_c5.log; stamp c5-tests
timeout 900 python bench.py --workload c5 --steps 3 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.log; tail -4 $OUT/bench_c5.log; stamp bench-c5
