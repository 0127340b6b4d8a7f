#!/bin/bash
# Run ON THE GPU BOX: every rocprofv3 pass behind profiles/r02_* (final round-2 build).
cd "$GRAFT_REPO_ROOT"
MFMA=1 tools/collect_profiles.sh r02_v1 > gpurun_out/r02_v1_collect.log 2>&1
MFMA=1 tools/collect_profiles.sh r02_cfg5 --half --model yolov8m --size 1280 --batch 16 --chunk 16 > gpurun_out/r02_cfg5_collect.log 2>&1
tools/trace_layers.sh r02_b1 yolov8n 1 --steps 50 --warmup 10 > gpurun_out/r02_b1_collect.log 2>&1
tools/trace_layers.sh r02_np32 yolov8n-pose 32 --steps 30 --warmup 5 > gpurun_out/r02_np32_collect.log 2>&1
tail -2 gpurun_out/r02_b1_layer_report.txt; ls gpurun_out | grep r02_ | head -40
