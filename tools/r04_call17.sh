#!/bin/bash
# Round 4, GPU call 17: launch plans of every bench workload re-timed on the current kernels (tools/make_plans.sh), stage times of the track loop
cd "$GRAFT_REPO_ROOT"
set -o pipefail
bash tools/make_plans.sh
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_tmp; mkdir -p $MI355_PLAN_CACHE
timeout -k 10 300 python tools/track_stages.py yolov8n 300 > gpurun_out/r04_c17_stages.log 2>&1; tail -12 gpurun_out/r04_c17_stages.log
