#!/bin/bash
# Round 4, GPU call 19a: rocprofv3 passes behind profiles/r04_v1_* (headline) and r04_b1_* (batch 1), shipped launch plans
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_fallback; mkdir -p $MI355_PLAN_CACHE
MFMA=1 timeout -k 10 700 tools/collect_profiles.sh r04_v1 > gpurun_out/r04_v1_collect.log 2>&1; tail -2 gpurun_out/r04_v1_collect.log
timeout -k 10 300 tools/trace_layers.sh r04_b1 yolov8n 1 --steps 50 --warmup 10 > gpurun_out/r04_b1_collect.log 2>&1; tail -3 gpurun_out/r04_b1_layer_report.txt
rm -f gpurun_out/r04_b1_trace/*/*_kernel_trace.csv
du -sh gpurun_out | tail -1
