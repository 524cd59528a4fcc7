#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $OUT/device.txt; nproc >> $OUT/device.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "layernorm or attention_split or upsample" > $OUT/pytest_kernels.log 2>&1; echo "rc=$?" >> $OUT/pytest_kernels.log; tail -4 $OUT/pytest_kernels.log
timeout 300 python tools/ab_probe.py D3R_LN_PAIR=0,1 D3R_UPSAMPLE_XCD=0,1 D3R_ATTN_SC=0,1 > $OUT/ab_probe.log 2>&1; grep -E "MEAN|outputs|==" $OUT/ab_probe.log
timeout 900 python -m pytest tests/test_timed_configs_gpu.py -m gpu -q -s --timeout 600 -p no:cacheprovider -k "released or 224_linear_batch or handover" > $OUT/pytest_parity.log 2>&1; echo "rc=$?" >> $OUT/pytest_parity.log; grep -E "vs CPU oracle|passed|failed|rc=" $OUT/pytest_parity.log | tail -40
