#!/bin/bash
# Round 3, GPU visit C: the two-blocks-per-CU split-fp16 GEMM shape (cfg 7: weights of a K step in registers) -- parity, phase trace, A/B on the
# forward; a longer run of the C4 stress harness.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[visit] $1 at +$(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_forward_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "split_fp16 or pinned_gemm_tile or kernel_variants" > $OUT/pytest_cfg7.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_cfg7.log; tail -6 $OUT/pytest_cfg7.log; stamp tests_cfg7
for r in 0 1; do
  D3R_GEMM_R=$r timeout 300 python bench.py --no-cpu-baseline --no-aligner --no-fast --steps 8 > $OUT/bench_r$r.json 2> $OUT/bench_r$r.log; grep -E "pairs/s|per-kernel" $OUT/bench_r$r.log | tail -3
done; stamp bench_ab
for r in 1 0; do
  D3R_GEMM_R=$r timeout 300 python bench.py --no-cpu-baseline --no-aligner --no-fast --no-profile --steps 8 2>&1 | grep -E "pairs/s" | tail -1
done; stamp bench_ab2
D3R_PROBE_EXTRA=0 timeout 300 python tools/gpu_probe.py gemmtrace > $OUT/gemmtrace_cfg7.log 2>&1; tail -40 $OUT/gemmtrace_cfg7.log; stamp gemmtrace
timeout 600 python tools/c4_stress.py run 12 > $OUT/c4_stress_long.log 2>&1; tail -3 $OUT/c4_stress_long.log; grep -c "ok=1" $OUT/c4_stress_long.log; stamp c4_stress
find $OUT -type f -size +6M -delete
du -sh $OUT
