#!/bin/bash
# Round 4, GPU call 16: streaming pointwise kernel with the fused upsample -- parity of every plan, then batch 1 with and without it
cd "$GRAFT_REPO_ROOT"
set -o pipefail
export MI355_PLAN_CACHE=0 MI355_PLAN_DIR=$PWD/gpurun_out/empty_dir; mkdir -p gpurun_out/empty_dir
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "upsample or every_launch_plan" > gpurun_out/r04_c16_tests.log 2>&1 || { tail -40 gpurun_out/r04_c16_tests.log; exit 1; }
tail -2 gpurun_out/r04_c16_tests.log
for M in "yolov8n 1" "yolov8n 1" "yolov5mu 1" "yolov8s-pose 8"; do
  set -- $M
  timeout -k 10 300 python bench.py --no-configs --no-cpu-baseline --model $1 --batch $2 --chunk $2 --steps 200 --warmup 20 > gpurun_out/r04_c16_$1_$2.json 2> gpurun_out/r04_c16_$1_$2.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_c16_$1_$2.json").read().strip().splitlines()[-1])
print("$1 b$2", d["value"], d["roofline"]["frac"], d["roofline"]["launches_per_step"], d["device_ms_per_step"])
PY
done
MI355_TUNE_LOG=1 timeout -k 10 300 python bench.py --no-configs --no-cpu-baseline --model yolov8n --batch 1 --chunk 1 --steps 20 --warmup 5 > /dev/null 2> gpurun_out/r04_c16_tune.log
grep "model.12.cv1\|model.15.cv1" gpurun_out/r04_c16_tune.log | head -30
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu > gpurun_out/r04_c16_e2e.log 2>&1 || { tail -40 gpurun_out/r04_c16_e2e.log; exit 1; }
tail -2 gpurun_out/r04_c16_e2e.log
