#!/bin/bash
# Round 4, first GPU call: the split engine + mi355_opts + fast_act against the GPU suite's core, then headline A/Bs.
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_tmp; mkdir -p $MI355_PLAN_CACHE
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_precision.py tests/test_gpu_e2e.py tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r04_c1_tests.log 2>&1 || { tail -30 gpurun_out/r04_c1_tests.log; exit 1; }
tail -3 gpurun_out/r04_c1_tests.log
B="timeout -k 10 300 python bench.py --no-configs --no-cpu-baseline --steps 12 --warmup 3"
$B > gpurun_out/r04_c1_canon.json 2> gpurun_out/r04_c1_canon.err && \
MI355_FAST_ACT=1 $B > gpurun_out/r04_c1_fast.json 2> gpurun_out/r04_c1_fast.err && \
MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_nowait.so MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_nowait $B > gpurun_out/r04_c1_nowait.json 2> gpurun_out/r04_c1_nowait.err && \
$B > gpurun_out/r04_c1_canon2.json 2> gpurun_out/r04_c1_canon2.err
for f in canon fast nowait canon2; do python - <<PY
import json
d=json.loads(open("gpurun_out/r04_c1_$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["roofline"]["frac"], d["roofline"]["launches_per_step"], d["roofline"]["plan_source"], d["device_ms_per_step"])
PY
done
