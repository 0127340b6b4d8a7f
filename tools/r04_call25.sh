#!/bin/bash
# Round 4, GPU call 25: version-8 kernel (512-thread blocks, weights by LDS-DMA into a double-buffered region) -- parity of every plan, plan tables
cd "$GRAFT_REPO_ROOT"
set -o pipefail
export MI355_PLAN_CACHE=0 MI355_PLAN_DIR=$PWD/gpurun_out/empty_dir; mkdir -p gpurun_out/empty_dir
timeout -k 10 400 python -m pytest tests/test_gpu_half.py -x -q -m gpu -k "every_plan_against_float64 or bits_do_not_depend" > gpurun_out/r04_c25_tests.log 2>&1 || { tail -40 gpurun_out/r04_c25_tests.log; exit 1; }
tail -2 gpurun_out/r04_c25_tests.log
export MI355_BENCH_HALF=1 MB_TOP=4
for R in 1 2; do
for SH in "16 80 80 192 192" "16 160 160 96 96" "16 320 320 48 48" "16 160 160 192 256" "16 160 160 64 64"; do
  MB_FILTER=v8 timeout -k 10 120 python tools/conv_microbench.py $SH 3 1 1 0 2>&1 | grep "v8\|conv " | head -4
  MB_FILTER=v7 timeout -k 10 120 python tools/conv_microbench.py $SH 3 1 1 0 2>&1 | grep "v7" | head -1
done
done
