#!/bin/bash
# Round 4, GPU call 6: small wave tiles for thin half=True launches -- parity of the new instances, then config 5 at batch 2 with and without them.
cd "$GRAFT_REPO_ROOT"
set -o pipefail
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_c6; mkdir -p $MI355_PLAN_CACHE
timeout -k 10 900 python -m pytest tests/test_gpu_half.py -x -q -m gpu > gpurun_out/r04_c6_tests.log 2>&1 || { tail -40 gpurun_out/r04_c6_tests.log; exit 1; }
tail -3 gpurun_out/r04_c6_tests.log
for SP in 0 1 0 1; do
  MI355_PLAN_CACHE=0 MI355_F16_SMALL_PT=$SP timeout -k 10 300 python bench.py --no-configs --no-cpu-baseline --half --model yolov8m --size 1280 --batch 2 --chunk 2 --steps 40 --warmup 5 \
      > gpurun_out/r04_c6_cfg5_b2_sp$SP.json 2> gpurun_out/r04_c6_cfg5_b2_sp$SP.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_c6_cfg5_b2_sp$SP.json").read().strip().splitlines()[-1])
print("cfg5 b2 small_pt=$SP", d["value"], d["roofline"]["frac"], d["roofline"]["launches_per_step"], d["device_ms_per_step"]["conv_ms"])
PY
done
MI355_PLAN_CACHE=0 MI355_TUNE_LOG=1 timeout -k 10 300 python bench.py --no-configs --no-cpu-baseline --half --model yolov8m --size 1280 --batch 2 --chunk 2 --steps 10 --warmup 3 \
      > gpurun_out/r04_c6_cfg5_b2_log.json 2> gpurun_out/r04_c6_cfg5_b2.tune.log
grep -c "PT[12] " gpurun_out/r04_c6_cfg5_b2.tune.log
