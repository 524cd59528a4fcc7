#!/bin/bash
# Round 3, measurement visit G (after the small-problem GEMM tile): GPU suite, smoke, the bench line as the driver runs it, rocprofv3 kernel trace + PMC passes (FETCH / WRITE / SQ),
# end-to-end inference() + alignment, aligner prefetch A/B.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $OUT/device.txt; nproc >> $OUT/device.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/device.txt
T0=$(date +%s)
stamp() { echo "[visit] $1 at +$(( $(date +%s) - T0 )) s"; }
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log; stamp tests
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -8 $OUT/smoke.log; stamp smoke
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.log; grep -E "pairs/s|per-kernel|aligner|cpu oracle" $OUT/bench.log | tail -12; stamp bench
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast --single-stream > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.log); stamp prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_fetch -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fast --single-stream > /dev/null 2> $OLDPWD/$OUT/pmc_fetch.log)
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_write -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fast --single-stream > /dev/null 2> $OLDPWD/$OUT/pmc_write.log); stamp pmc
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OLDPWD/$OUT/pmc_sq1 -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fast --single-stream > /dev/null 2> $OLDPWD/$OUT/pmc_sq1.log); stamp pmcsq
timeout 400 python tools/e2e_pipeline.py 100 swin-3 > $OUT/e2e_full.log 2>&1; tail -14 $OUT/e2e_full.log; stamp e2e
python tools/summarize_prof.py $OUT > $OUT/prof_summary.txt 2>&1
find $OUT -type f -size +6M -delete
du -sh $OUT
timeout 300 python tools/latency_probe.py forward-only > $OUT/latency.log 2>&1; grep forward $OUT/latency.log; stamp latency
