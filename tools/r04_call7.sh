#!/bin/bash
# Round 4, GPU call 7 (first call after the container was re-created): the whole GPU suite on the restored tree, then the launch plans of
# every bench workload timed into gpurun_out/plans_new/ (tools/make_plans.sh).
cd "$GRAFT_REPO_ROOT"
set -o pipefail
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_tmp; mkdir -p $MI355_PLAN_CACHE
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r04_c7_tests.log 2>&1 || { tail -40 gpurun_out/r04_c7_tests.log; exit 1; }
tail -3 gpurun_out/r04_c7_tests.log
bash tools/make_plans.sh
