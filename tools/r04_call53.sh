#!/bin/bash
# Round 4, GPU call 53: residual reads of the half-mode epilogue issued ahead (store_tiles_f16_v2): half tests, then config 5's per-layer table (batch 16) and rates
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_half.py -x -q -m gpu > gpurun_out/r04_c53_tests.log 2>&1; tail -2 gpurun_out/r04_c53_tests.log
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_fallback; mkdir -p $MI355_PLAN_CACHE
SIZE=1280 ES=2 timeout -k 10 300 tools/trace_layers.sh r04_res_b16 yolov8m 16 --half --size 1280 --steps 10 --warmup 3 > gpurun_out/r04_res_b16.log 2>&1
grep -E "\.m\.[0-9]\.cv2 " gpurun_out/r04_res_b16_layer_report.txt | cut -c1-42,100-190; tail -1 gpurun_out/r04_res_b16_layer_report.txt | cut -c1-200
rm -f gpurun_out/r04_res_b16_trace/*/*_kernel_trace.csv
for b in 16 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch $b --chunk $b --steps 30 --warmup 5 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b$b', d['value'], 'fps', d['ms_per_step'], 'ms', d['roofline']['frac'])"
done
