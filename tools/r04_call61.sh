#!/bin/bash
# Round 4, GPU call 61: one-launch NMS for big maps (nms_prefix_kernel): NMS tests (both paths, fallbacks), half e2e tests, config 5 at batch 2 / 16 with and without
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_half.py -x -q -m gpu -k "nms or config5 or predict_half or reference_call" > gpurun_out/r04_c61_tests.log 2>&1; tail -3 gpurun_out/r04_c61_tests.log
for p in 1 0; do for b in 2 16; do
  MI355_NMS_PREFIX=$p timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch $b --chunk $b --steps 40 --warmup 8 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prefix=$p b$b', d['value'], 'fps', d['ms_per_step'], 'ms', d['roofline']['frac'], 'nms_ms', d['device_ms_per_step']['nms_ms'])"
done; done
