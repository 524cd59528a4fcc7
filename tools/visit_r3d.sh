#!/bin/bash
# Round 3, GPU visit D: 2 x 2-block upsampling kernel, selective cfg 7 rule, full suite, bench, the N = 2 self-test of bench.py on one device.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[visit] $1 at +$(( $(date +%s) - T0 )) s"; }
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 200 -p no:cacheprovider -k "upsample" > $OUT/pytest_upsample.log 2>&1
URC=$?; echo "pytest rc=$URC" >> $OUT/pytest_upsample.log; tail -4 $OUT/pytest_upsample.log; stamp upsample
if [ $URC -ne 0 ]; then export D3R_UPSAMPLE_V1=1; echo "[visit] new upsampling kernel FAILED: rest of the visit with D3R_UPSAMPLE_V1=1"; fi
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -8 $OUT/pytest_gpu.log; stamp tests
timeout 600 python bench.py --no-cpu-baseline --no-aligner --no-fast --steps 8 > $OUT/bench_new.json 2> $OUT/bench_new.log; grep -E "pairs/s|per-kernel" $OUT/bench_new.log | tail -3
D3R_UPSAMPLE_V1=1 D3R_GEMM_R=0 timeout 300 python bench.py --no-cpu-baseline --no-aligner --no-fast --steps 8 > $OUT/bench_old.json 2> $OUT/bench_old.log; grep -E "pairs/s|per-kernel" $OUT/bench_old.log | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-aligner --no-fast --no-profile --steps 8 2>&1 | grep -E "pairs/s" | tail -1
D3R_UPSAMPLE_V1=1 D3R_GEMM_R=0 timeout 300 python bench.py --no-cpu-baseline --no-aligner --no-fast --no-profile --steps 8 2>&1 | grep -E "pairs/s" | tail -1; stamp bench_ab
D3R_BENCH_ONE_DEVICE=1 D3R_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_gpus2_selftest.json 2> $OUT/bench_gpus2_selftest.log; echo "rc=$?"; tail -3 $OUT/bench_gpus2_selftest.log; cut -c1-600 $OUT/bench_gpus2_selftest.json; stamp gpus2_selftest
find $OUT -type f -size +6M -delete
du -sh $OUT
