#!/bin/bash
# Round 4, GPU call 32: launch plans re-timed on the final kernels (version 9 added to the half-mode pointwise candidates), half suite
cd "$GRAFT_REPO_ROOT"
set -o pipefail
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_tmp; mkdir -p $MI355_PLAN_CACHE
timeout -k 10 400 python -m pytest tests/test_gpu_half.py -x -q -m gpu > gpurun_out/r04_c32_tests.log 2>&1 || { tail -40 gpurun_out/r04_c32_tests.log; exit 1; }
tail -2 gpurun_out/r04_c32_tests.log
bash tools/make_plans.sh
