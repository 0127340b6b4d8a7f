#!/bin/bash
# Round 4, GPU call 43: the k3 s2 half stem -- tests, then per-layer tables of config 5 at batch 16 and 2 with it (MI355_STEM_LEAN=0 run beside for the A/B)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests/test_gpu_half.py -x -q -m gpu -k "stem or config5 or raw_head or reproducible" > gpurun_out/r04_c43_tests.log 2>&1; tail -3 gpurun_out/r04_c43_tests.log
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_fallback; mkdir -p $MI355_PLAN_CACHE
SIZE=1280 ES=2 timeout -k 10 300 tools/trace_layers.sh r04_lean_b16 yolov8m 16 --half --size 1280 --steps 10 --warmup 3 > gpurun_out/r04_lean_b16.log 2>&1; grep -E "^model.0 |stem" gpurun_out/r04_lean_b16_layer_report.txt | cut -c1-180
MI355_STEM_LEAN=0 SIZE=1280 ES=2 timeout -k 10 300 tools/trace_layers.sh r04_gen_b16 yolov8m 16 --half --size 1280 --steps 10 --warmup 3 > gpurun_out/r04_gen_b16.log 2>&1; grep -E "^model.0 |stem" gpurun_out/r04_gen_b16_layer_report.txt | cut -c1-180
SIZE=1280 ES=2 timeout -k 10 300 tools/trace_layers.sh r04_lean_b2 yolov8m 2 --half --size 1280 --steps 20 --warmup 5 > gpurun_out/r04_lean_b2.log 2>&1; grep -E "^model.0 |stem" gpurun_out/r04_lean_b2_layer_report.txt | cut -c1-180
rm -f gpurun_out/r04_lean_b16_trace/*/*_kernel_trace.csv gpurun_out/r04_gen_b16_trace/*/*_kernel_trace.csv gpurun_out/r04_lean_b2_trace/*/*_kernel_trace.csv
