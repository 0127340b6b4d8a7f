#!/bin/bash
# Round 4, GPU call 18: the whole GPU suite on the final kernels (shipped launch plans), then the cost of the two wait states behind every
# 16-byte buffer store (common.h:buffer_store_b128; library built with -DMI355_STORE_WAIT=0 beside the product one), alternating runs
cd "$GRAFT_REPO_ROOT"
set -o pipefail
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/r04_c18_tests.log 2>&1 || { tail -40 gpurun_out/r04_c18_tests.log; exit 1; }
tail -3 gpurun_out/r04_c18_tests.log
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_c18; mkdir -p $MI355_PLAN_CACHE
B="timeout -k 10 300 python bench.py --no-configs --no-cpu-baseline --steps 12 --warmup 3"
for R in 1 2; do
  $B > gpurun_out/r04_c18_wait_$R.json 2> gpurun_out/r04_c18_wait_$R.err
  MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_nowait.so $B > gpurun_out/r04_c18_nowait_$R.json 2> gpurun_out/r04_c18_nowait_$R.err
done
for f in wait_1 nowait_1 wait_2 nowait_2; do python - <<PY
import json
d=json.loads(open("gpurun_out/r04_c18_$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["roofline"]["frac"], d["roofline"]["launches_per_step"], d["roofline"]["plan_source"], d["roofline"]["plan_hash"])
PY
done
