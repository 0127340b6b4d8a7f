#!/bin/bash
# A/B on the GPU box: stream priorities of the detector (highest) and of the tracker's motion compensation (lowest) in the frame-by-frame loop
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_tmp; mkdir -p $MI355_PLAN_CACHE
for cfg in "1 1" "0 0" "1 0" "0 1" "1 1" "0 0"; do
  set -- $cfg
  echo "== MI355_ENGINE_PRIO=$1 MI355_GMC_PRIO=$2"
  MI355_ENGINE_PRIO=$1 MI355_GMC_PRIO=$2 timeout -k 10 200 python tools/track_stages.py yolov8n 300 2>&1 | grep -v amdgpu.ids | tail -9
done
