#!/bin/bash
# Run ON THE GPU BOX: every rocprofv3 pass behind profiles/r04_* (final round-4 build, SHIPPED launch plans: every pass, the driver-style
# bench run and the driver's own run execute the same launch sequence -- plan_source "file", one plan hash).
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_fallback; mkdir -p $MI355_PLAN_CACHE     # only shapes without a shipped file land here
MFMA=1 tools/collect_profiles.sh r04_v1 > gpurun_out/r04_v1_collect.log 2>&1
MFMA=1 tools/collect_profiles.sh r04_cfg5 --half --model yolov8m --size 1280 --batch 16 --chunk 16 > gpurun_out/r04_cfg5_collect.log 2>&1
tools/trace_layers.sh r04_b1 yolov8n 1 --steps 50 --warmup 10 > gpurun_out/r04_b1_collect.log 2>&1
SIZE=1280 ES=2 tools/trace_layers.sh r04_cfg5_b2 yolov8m 2 --half --size 1280 --steps 20 --warmup 5 > gpurun_out/r04_cfg5_b2_collect.log 2>&1
# the track loop (detector at batch 1 + motion compensation on the GPU) and the batched sweep: kernel stats of csrc/gmc_kernels.hip beside the detector's
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r04_track_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_track_trace -- python3 tools/track_pipeline_bench.py 150 > gpurun_out/r04_track_trace.log 2>&1
# gpurun merges at most 64 MiB back: the raw traces of the layer reports (already summarised above) and of the track loop stay on the box
rm -f gpurun_out/r04_b1_trace/*/*_kernel_trace.csv gpurun_out/r04_cfg5_b2_trace/*/*_kernel_trace.csv gpurun_out/r04_track_trace/*/*_kernel_trace.csv
tail -3 gpurun_out/r04_b1_layer_report.txt; ls gpurun_out | grep r04_ | head -60
