#!/bin/bash
# Round 4, GPU call 34: version-10 pointwise kernel (full-line pixel staging through wave-private LDS) -- parity, plan tables against versions 9 and 1
cd "$GRAFT_REPO_ROOT"
set -o pipefail
export MI355_PLAN_CACHE=0 MI355_PLAN_DIR=$PWD/gpurun_out/empty_dir; mkdir -p gpurun_out/empty_dir
timeout -k 10 400 python -m pytest tests/test_gpu_half.py -x -q -m gpu -k "every_plan_against_float64 or bits_do_not_depend" > gpurun_out/r04_c34_tests.log 2>&1 || { tail -40 gpurun_out/r04_c34_tests.log; exit 1; }
tail -2 gpurun_out/r04_c34_tests.log
export MI355_BENCH_HALF=1 MB_TOP=2
for SH in "16 160 160 576 192" "16 320 320 192 96" "16 80 80 1152 384" "16 160 160 384 192" "16 80 80 768 384" "16 40 40 1152 576" "16 80 80 384 384"; do
  MB_FILTER=v10 timeout -k 10 120 python tools/conv_microbench.py $SH 1 1 1 0 2>&1 | grep "v10\|conv " | head -3
  MB_FILTER=v9 timeout -k 10 120 python tools/conv_microbench.py $SH 1 1 1 0 2>&1 | grep "v9" | head -1
  MB_FILTER=v1 timeout -k 10 200 python tools/conv_microbench.py $SH 1 1 1 0 2>&1 | grep "v1 " | head -1
done
