#!/bin/bash
# ONE parametrised visit script (replaces the per-visit tools/visit_r*.sh of rounds 3-4).
#   on the GPU box (through gpurun):   bash tools/visit.sh run <name> <gpu_round.sh words...>      e.g.  run r05_a tests bench prof pmc
#                                      bash tools/visit.sh ab  <name> <ENVVAR> <valueA> <valueB> [reps] [bench.py args...]
#   here, after the call returned:     bash tools/visit.sh collect <name>     copies the summaries from gpurun_out/ into profiles/<name>/
# `run` names the visit in the PMC digest (D3R_VISIT) so that bench.py's roofline.traffic_source says where its figure was taken.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mode="${1:?run|ab|collect}"; name="${2:?visit name}"; shift 2
OUT=gpurun_out
mkdir -p $OUT
case $mode in
  run)
    D3R_VISIT="profiles/$name" bash tools/gpu_round.sh "$@" ;;
  ab)
    # same-box A/B of one environment switch on the driver-form forward: alternating runs, every line kept
    var="${1:?env var}"; a="${2:?value A}"; b="${3:?value B}"; reps="${4:-3}"; shift 4 2>/dev/null || shift $#
    export TMPDIR=/tmp
    for r in $(seq 1 "$reps"); do for v in "$a" "$b"; do
      env "$var=$v" timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fast --no-aligner --no-parity --no-profile "$@" > $OUT/ab_${var}_${v}_$r.json 2> $OUT/ab_${var}_${v}_$r.log
      echo "$var=$v rep $r: $(grep -o 'pairs/s on.*' $OUT/ab_${var}_${v}_$r.log | head -1) $(python -c "import json;print(json.load(open('$OUT/ab_${var}_${v}_$r.json'))['value'])" 2>/dev/null)" | tee -a $OUT/ab_${var}.txt
    done; done ;;
  collect)
    dst=profiles/$name; mkdir -p $dst
    for f in device.txt prof_summary.txt prof_bench.json bench.json bench.log pytest_gpu.log pytest_gpu_fast.log smoke.log latency.log e2e_full.log probe.log; do [ -f $OUT/$f ] && cp $OUT/$f $dst/; done
    cp $OUT/ab_*.txt $OUT/*selftest*.json $OUT/bench_*.json $dst/ 2>/dev/null
    f=$(find $OUT/prof -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $dst/bench_kernel_stats.csv
    if [ -f $OUT/pmc_latest.json ]; then cp $OUT/pmc_latest.json $dst/; cp $OUT/pmc_latest.json profiles/pmc_latest.json; fi
    ls $dst ;;
esac
