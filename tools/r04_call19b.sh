#!/bin/bash
# Round 4, GPU call 19b: rocprofv3 passes behind profiles/r04_cfg5_* (YOLOv8m 1280x1280 half, batch 16 and 2) and the track loop's kernel stats
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_fallback; mkdir -p $MI355_PLAN_CACHE
MFMA=1 timeout -k 10 600 tools/collect_profiles.sh r04_cfg5 --half --model yolov8m --size 1280 --batch 16 --chunk 16 > gpurun_out/r04_cfg5_collect.log 2>&1; tail -2 gpurun_out/r04_cfg5_collect.log
SIZE=1280 ES=2 timeout -k 10 300 tools/trace_layers.sh r04_cfg5_b2 yolov8m 2 --half --size 1280 --steps 20 --warmup 5 > gpurun_out/r04_cfg5_b2_collect.log 2>&1; tail -3 gpurun_out/r04_cfg5_b2_layer_report.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r04_track_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_track_trace -- python3 tools/track_pipeline_bench.py 150 > gpurun_out/r04_track_trace.log 2>&1
rm -f gpurun_out/r04_cfg5_b2_trace/*/*_kernel_trace.csv gpurun_out/r04_track_trace/*/*_kernel_trace.csv
du -sh gpurun_out | tail -1
