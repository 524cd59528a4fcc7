#!/bin/bash
# Round 4, visit A: parity on the timed configurations (tests/test_timed_configs_gpu.py), the head postprocess modes, bench.py with its
# parity_check block (default line) and the c3 / c5 workloads at N = 1.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[visit] $1 at +$(( $(date +%s) - T0 )) s"; }
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $OUT/device.txt; nproc >> $OUT/device.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/device.txt
timeout 1500 python -m pytest tests/test_timed_configs_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -s > $OUT/pytest_timed_configs.log 2>&1; echo "rc=$?" >> $OUT/pytest_timed_configs.log
grep -E "vs CPU oracle|linear.*rel err|passed|failed|rc=|Error|assert" $OUT/pytest_timed_configs.log | tail -40; stamp timed-config-tests
timeout 600 python -m pytest tests/test_forward_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "postprocess_modes or fused_head or graph_replay" > $OUT/pytest_post_modes.log 2>&1; echo "rc=$?" >> $OUT/pytest_post_modes.log; tail -5 $OUT/pytest_post_modes.log; stamp post-modes
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.log; grep -E "parity_check|pairs/s on|aligner|cpu oracle" $OUT/bench.log | tail -12; stamp bench
timeout 600 python bench.py --workload c3 --steps 5 --warmup 1 > $OUT/bench_c3.json 2> $OUT/bench_c3.log; tail -3 $OUT/bench_c3.log; stamp bench-c3
timeout 900 python bench.py --workload c5 --steps 3 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.log; tail -4 $OUT/bench_c5.log; stamp bench-c5
