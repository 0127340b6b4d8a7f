#!/bin/bash
# Round 4, GPU call 15: version-7 plans in the engine -- the half=True suite, then config 5 at batch 16 / 2 with and without them (MI355_CONV_V7)
cd "$GRAFT_REPO_ROOT"
set -o pipefail
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_c15; mkdir -p $MI355_PLAN_CACHE
timeout -k 10 900 python -m pytest tests/test_gpu_half.py -x -q -m gpu > gpurun_out/r04_c15_tests.log 2>&1 || { tail -40 gpurun_out/r04_c15_tests.log; exit 1; }
tail -2 gpurun_out/r04_c15_tests.log
export MI355_PLAN_CACHE=0 MI355_PLAN_DIR=$PWD/gpurun_out/empty_dir; mkdir -p gpurun_out/empty_dir
for V7 in 0 1 0 1; do
  for B in 16 2; do
  MI355_CONV_V7=$V7 timeout -k 10 400 python bench.py --no-configs --no-cpu-baseline --half --model yolov8m --size 1280 --batch $B --chunk $B --steps 30 --warmup 5 \
      > gpurun_out/r04_c15_cfg5_b${B}_v$V7.json 2> gpurun_out/r04_c15_cfg5_b${B}_v$V7.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_c15_cfg5_b${B}_v$V7.json").read().strip().splitlines()[-1])
print("cfg5 b$B v7=$V7", d["value"], d["roofline"]["frac"], d["roofline"]["launches_per_step"], d["device_ms_per_step"]["conv_ms"])
PY
  done
done
MI355_TUNE_LOG=1 timeout -k 10 400 python bench.py --no-configs --no-cpu-baseline --half --model yolov8m --size 1280 --batch 16 --chunk 16 --steps 5 --warmup 2 > /dev/null 2> gpurun_out/r04_c15_tune.log
grep -c "v7" gpurun_out/r04_c15_tune.log
