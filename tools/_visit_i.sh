#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in "" "--no-latency" "--no-fast" "--no-latency --no-fast" "--no-profile --no-parity --no-latency --no-fast"; do
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline $v > $OUT/bis.json 2> $OUT/bis.log
  echo "[$v] $(python -c "import json;d=json.load(open('$OUT/bis.json'));print(round(d['value'],1), round(d['aligner']['value'],1), round(d['aligner']['ms_total'],1))")"
done
D3R_LN_FOLD=0 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bis.json 2> $OUT/bis.log
echo "[fold=0 full] $(python -c "import json;d=json.load(open('$OUT/bis.json'));print(round(d['value'],1), round(d['aligner']['value'],1), round(d['aligner']['ms_total'],1))")"
