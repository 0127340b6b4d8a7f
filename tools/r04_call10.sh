#!/bin/bash
# Round 4, GPU call 10: version-7 kernel with the loop re-ordered (global prefetch truly in flight during the MFMAs) -- parity, plan tables
cd "$GRAFT_REPO_ROOT"
set -o pipefail
export MI355_PLAN_CACHE=0 MI355_PLAN_DIR=$PWD/gpurun_out/empty_dir; mkdir -p gpurun_out/empty_dir
timeout -k 10 600 python -m pytest tests/test_gpu_half.py -x -q -m gpu -k "every_plan_against_float64 or bits_do_not_depend" > gpurun_out/r04_c10_tests.log 2>&1 || { tail -40 gpurun_out/r04_c10_tests.log; exit 1; }
tail -3 gpurun_out/r04_c10_tests.log
for SH in "16 80 80 192 192" "16 160 160 96 96" "16 320 320 48 48" "16 40 40 288 288" "16 160 160 192 256" "16 160 160 64 64"; do
  MI355_BENCH_HALF=1 MB_TOP=2 timeout -k 10 200 python tools/conv_microbench.py $SH 3 1 1 0 2>&1 | grep -v amdgpu.ids
done
