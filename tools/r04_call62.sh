#!/bin/bash
# Round 4, GPU call 62: greedy NMS pass with the next chunk's candidates prefetched: NMS / e2e tests, then config 5 (batch 2, 16), headline and batch 1
cd "$GRAFT_REPO_ROOT"
timeout -k 10 700 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_gpu_half.py -x -q -m gpu -k "nms or bit_exact or config5 or reference_call" > gpurun_out/r04_c62_tests.log 2>&1; tail -2 gpurun_out/r04_c62_tests.log
for b in 2 16; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch $b --chunk $b --steps 40 --warmup 8 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 b$b', d['value'], 'fps', d['ms_per_step'], 'ms', d['roofline']['frac'], 'nms_ms', d['device_ms_per_step']['nms_ms'])"
done
timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --steps 6 --warmup 2 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], 'fps', d['ms_per_step'], 'ms', 'nms_ms', d['device_ms_per_step']['nms_ms'])"
timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --batch 1 --chunk 1 --steps 400 --warmup 50 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch1', d['value'], 'fps', d['ms_per_step'], 'ms')"
