#!/bin/bash
# Round 4, GPU call 42: per-layer table of the headline workload in the opt-in fast_act mode (what keeps it at 0.67 of the fp32 matrix peak)
cd "$GRAFT_REPO_ROOT"
export MI355_FAST_ACT=1 MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_fallback; mkdir -p $MI355_PLAN_CACHE
timeout -k 10 500 tools/trace_layers.sh r04_fast yolov8n 512 --steps 5 --warmup 2 > gpurun_out/r04_fast_collect.log 2>&1; tail -3 gpurun_out/r04_fast_layer_report.txt
rm -f gpurun_out/r04_fast_trace/*/*_kernel_trace.csv
