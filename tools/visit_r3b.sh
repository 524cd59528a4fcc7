#!/bin/bash
# Round 3, GPU visit B: the pipelined attention kernel after the permlane fix, packed GELU / split epilogues, GEMM phase-trace diagnostics
# (per-round K loop / epilogue, no-store, zero operands, stagger), the one-pair-per-call profile.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[visit] $1 at +$(( $(date +%s) - T0 )) s"; }
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 200 -p no:cacheprovider -k "attention" -s > $OUT/pytest_attention.log 2>&1
ARC=$?; echo "pytest rc=$ARC" >> $OUT/pytest_attention.log; grep -E "attention x3|passed|failed" $OUT/pytest_attention.log | tail -20; stamp attention
if [ $ARC -ne 0 ]; then export D3R_ATTN_V1=1; echo "[visit] new attention kernel FAILED its test: the rest of the visit runs with D3R_ATTN_V1=1"; fi
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -12 $OUT/pytest_gpu.log; stamp tests
timeout 600 python bench.py --no-cpu-baseline --no-aligner > $OUT/bench.json 2> $OUT/bench.log; grep -E "pairs/s|per-kernel|attention" $OUT/bench.log | tail -12; stamp bench
D3R_ATTN_V1=1 timeout 300 python bench.py --no-cpu-baseline --no-aligner --no-fast --steps 6 > $OUT/bench_attn_v1.json 2> $OUT/bench_attn_v1.log; grep -E "pairs/s|attention" $OUT/bench_attn_v1.log | tail -4; stamp bench_attn_v1
timeout 300 python bench.py --pairs 1 --steps 30 --warmup 5 --no-cpu-baseline --no-aligner --no-fast > $OUT/bench_b1.json 2> $OUT/bench_b1.log; grep -E "pairs/s|per-kernel|TF/s" $OUT/bench_b1.log | tail -45; stamp bench_b1
timeout 400 python tools/gpu_probe.py gemmtrace > $OUT/gemmtrace.log 2>&1; tail -60 $OUT/gemmtrace.log; stamp gemmtrace
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aligner --no-fast --single-stream > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.log); stamp prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OLDPWD/$OUT/pmc_sq1 -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aligner --no-profile --no-fast --single-stream > /dev/null 2> $OLDPWD/$OUT/pmc_sq1.log); stamp pmcsq
rm -rf $OUT/pmc_fetch $OUT/pmc_write
python tools/summarize_prof.py $OUT > $OUT/prof_summary.txt 2>&1
find $OUT -type f -size +6M -delete
du -sh $OUT
