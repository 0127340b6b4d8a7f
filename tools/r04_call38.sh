#!/bin/bash
# Round 4, GPU call 38: version-10 pointwise kernel with the upsample fused into its read side (half mode) -- parity, then config 5 tuned afresh
cd "$GRAFT_REPO_ROOT"
set -o pipefail
export MI355_PLAN_CACHE=0 MI355_PLAN_DIR=$PWD/gpurun_out/empty_dir; mkdir -p gpurun_out/empty_dir
timeout -k 10 600 python -m pytest tests/test_gpu_half.py -x -q -m gpu > gpurun_out/r04_c38_tests.log 2>&1 || { tail -40 gpurun_out/r04_c38_tests.log; exit 1; }
tail -2 gpurun_out/r04_c38_tests.log
for B in 16 2 16; do
  timeout -k 10 400 python bench.py --no-configs --no-cpu-baseline --half --model yolov8m --size 1280 --batch $B --chunk $B --steps 30 --warmup 5 > gpurun_out/r04_c38_b$B.json 2> gpurun_out/r04_c38_b$B.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_c38_b$B.json").read().strip().splitlines()[-1])
print("cfg5 b$B", d["value"], d["roofline"]["frac"], d["roofline"]["launches_per_step"], d["device_ms_per_step"])
PY
done
MI355_TUNE_LOG=1 timeout -k 10 400 python bench.py --no-configs --no-cpu-baseline --half --model yolov8m --size 1280 --batch 16 --chunk 16 --steps 3 --warmup 1 > /dev/null 2> gpurun_out/r04_c38_tune.log
grep "model.12.cv1\|model.15.cv1" gpurun_out/r04_c38_tune.log | sort -t: -k2 -n | head -12
