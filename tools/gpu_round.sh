#!/bin/bash
# One gpurun visit: parity tests, kernel probes, bench line and a rocprofv3 kernel trace. Everything lands in gpurun_out/.
# Usage (from the repo root, on the GPU box): bash tools/gpu_round.sh [tests] [probe] [bench] [prof] [pmc] [pmcsq] [f8]
# (tools/f8_probe.bin: hipcc --offload-arch=gfx950 -O2 tools/f8_probe.hip -o tools/f8_probe.bin, built here, travels with the snapshot)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
WHAT="${*:-tests probe bench prof}"
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $OUT/device.txt
nproc >> $OUT/device.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/device.txt
for w in $WHAT; do
  case $w in
    testsfast) timeout 200 python -m pytest tests -m gpu -q -x --timeout 120 -p no:cacheprovider -k "not full_size and not config1" > $OUT/pytest_gpu_fast.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_fast.log; tail -8 $OUT/pytest_gpu_fast.log ;;
    tests) timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log ;;
    testsall) timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log ;;
    probe) timeout 900 python tools/gpu_probe.py gemm conv attn forward aligner > $OUT/probe.log 2>&1; tail -60 $OUT/probe.log ;;
    tune) timeout 600 python tools/gpu_probe.py tune > $OUT/probe_tune.log 2>&1; tail -40 $OUT/probe_tune.log ;;
    f8) # the fp16 + fp8 mode's side measurements: hardware probe (instruction semantics + MFMA issue rates), error distribution over weight
        # seeds, per-block phase trace of the block linears
        [ -x tools/f8_probe.bin ] && timeout 120 ./tools/f8_probe.bin > $OUT/f8_probe.log 2>&1; tail -5 $OUT/f8_probe.log
        timeout 300 python tools/margin_survey.py 6 4 > $OUT/margin_survey.log 2>&1; tail -3 $OUT/margin_survey.log
        D3R_PROBE_DT=fp16f8 timeout 300 python tools/gpu_probe.py gemmtrace > $OUT/gemmtrace_f8.log 2>&1; tail -12 $OUT/gemmtrace_f8.log ;;
    benchq) timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.log; grep -E "bench\]" $OUT/bench_quick.log | tail -60 ;;
    bench) timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.log; tail -12 $OUT/bench.log; cat $OUT/bench.json ;;
    prof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast --no-parity --no-latency --no-aligner-380 --single-stream > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.log); find $OUT/prof -name "*stats*" | head; f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" ;;
    pmc) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_fetch -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fast --no-parity --no-latency --no-aligner-380 --single-stream > /dev/null 2> $OLDPWD/$OUT/pmc_fetch.log);
         (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_write -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fast --no-parity --no-latency --no-aligner-380 --single-stream > /dev/null 2> $OLDPWD/$OUT/pmc_write.log); ls $OUT/pmc_fetch $OUT/pmc_write ;;
    pmcsq) # issue / MFMA-busy / wait breakdown per kernel (SQ has 8 slots per pass, GRBM is separate); two passes, kernel-trace only
         (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OLDPWD/$OUT/pmc_sq1 -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fast --no-parity --no-latency --no-aligner-380 --single-stream > /dev/null 2> $OLDPWD/$OUT/pmc_sq1.log);
         (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OLDPWD/$OUT/pmc_sq2 -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fast --no-parity --no-latency --no-aligner-380 --single-stream > /dev/null 2> $OLDPWD/$OUT/pmc_sq2.log); tail -3 $OUT/pmc_sq1.log $OUT/pmc_sq2.log; ls $OUT/pmc_sq1 $OUT/pmc_sq2 ;;
  esac
done
# summarise the per-dispatch PMC csvs (sum / mean per kernel) and keep the merged-back payload small
python tools/summarize_prof.py $OUT > $OUT/prof_summary.txt 2>&1
find $OUT -type f -size +6M -delete
du -sh $OUT
