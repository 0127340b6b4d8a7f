#!/bin/bash
# Run ON THE GPU BOX: every rocprofv3 pass behind profiles/r03_* (final round-3 build).
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=$PWD/gpurun_out/r03_plans; mkdir -p $MI355_PLAN_CACHE     # every pass of one workload runs the same launch plans
MFMA=1 tools/collect_profiles.sh r03_v1 > gpurun_out/r03_v1_collect.log 2>&1
MFMA=1 tools/collect_profiles.sh r03_cfg5 --half --model yolov8m --size 1280 --batch 16 --chunk 16 > gpurun_out/r03_cfg5_collect.log 2>&1
tools/trace_layers.sh r03_b1 yolov8n 1 --steps 50 --warmup 10 > gpurun_out/r03_b1_collect.log 2>&1
tools/trace_layers.sh r03_np32 yolov8n-pose 32 --steps 30 --warmup 5 > gpurun_out/r03_np32_collect.log 2>&1
tools/trace_layers.sh r03_v5mu_b1 yolov5mu 1 --steps 50 --warmup 10 > gpurun_out/r03_v5mu_b1_collect.log 2>&1
# the track loop (detector at batch 1 + motion compensation on the GPU): kernel stats of csrc/gmc_kernels.hip beside the detector's
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r03_track_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_track_trace -- python3 tools/track_pipeline_bench.py 150 > gpurun_out/r03_track_trace.log 2>&1
# gpurun merges at most 64 MiB back: the raw traces of the layer reports (already summarised above) and of the track loop stay on the box
rm -f gpurun_out/r03_b1_trace/*/*_kernel_trace.csv gpurun_out/r03_np32_trace/*/*_kernel_trace.csv gpurun_out/r03_v5mu_b1_trace/*/*_kernel_trace.csv gpurun_out/r03_track_trace/*/*_kernel_trace.csv
tail -3 gpurun_out/r03_b1_layer_report.txt; ls gpurun_out | grep r03_ | head -40
