#!/bin/bash
# Round 3, GPU visit A: kernel-level attention first (decides whether the rest runs on the new attention kernel), C4 stress harness,
# the GPU suite, the bench line (default precision fp16x3), rocprofv3 kernel trace + PMC passes, small-batch latency.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $OUT/device.txt; nproc >> $OUT/device.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/device.txt
T0=$(date +%s)
stamp() { echo "[visit] $1 at +$(( $(date +%s) - T0 )) s"; }
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 200 -p no:cacheprovider -k "attention" -s > $OUT/pytest_attention.log 2>&1
ARC=$?; echo "pytest rc=$ARC" >> $OUT/pytest_attention.log; tail -4 $OUT/pytest_attention.log; stamp attention
if [ $ARC -ne 0 ]; then export D3R_ATTN_V1=1; echo "[visit] new attention kernel FAILED its test: the rest of the visit runs with D3R_ATTN_V1=1"; fi
timeout 400 python tools/c4_stress.py run 3 > $OUT/c4_stress.log 2>&1; tail -8 $OUT/c4_stress.log; stamp c4_stress
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log; stamp tests
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.log; tail -25 $OUT/bench.log; stamp bench
D3R_ATTN_V1=1 timeout 300 python bench.py --no-cpu-baseline --no-aligner --no-fast --steps 6 > $OUT/bench_attn_v1.json 2> $OUT/bench_attn_v1.log; grep -E "pairs/s|attention" $OUT/bench_attn_v1.log | tail -4; stamp bench_attn_v1
D3R_GEMM_T256=250 timeout 300 python bench.py --no-cpu-baseline --no-aligner --no-fast --steps 6 > $OUT/bench_t256_250.json 2> $OUT/bench_t256_250.log; grep -E "pairs/s" $OUT/bench_t256_250.log | tail -2; stamp bench_t256
timeout 300 python tools/latency_probe.py forward-only > $OUT/latency.log 2>&1; tail -9 $OUT/latency.log; stamp latency
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast --single-stream > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.log); stamp prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_fetch -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fast --single-stream > /dev/null 2> $OLDPWD/$OUT/pmc_fetch.log)
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_write -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fast --single-stream > /dev/null 2> $OLDPWD/$OUT/pmc_write.log); stamp pmc
python tools/summarize_prof.py $OUT > $OUT/prof_summary.txt 2>&1
find $OUT -type f -size +6M -delete
du -sh $OUT
