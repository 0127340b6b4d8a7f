#!/bin/bash
# Round 4, GPU call 3: data for the half=True kernels -- every candidate plan's time for config 5 at batch 16 and batch 2 (tune log),
# and whether the half=True conv's bits depend on the launch plan.
cd "$GRAFT_REPO_ROOT"
set -o pipefail
timeout -k 10 300 python tools/track_stages.py yolov8n 200 > gpurun_out/r04_c3_stages_n.log 2>&1; tail -9 gpurun_out/r04_c3_stages_n.log
timeout -k 10 300 python tools/track_stages.py yolov8s-pose 100 > gpurun_out/r04_c3_stages_spose.log 2>&1; tail -9 gpurun_out/r04_c3_stages_spose.log
timeout -k 10 300 python tools/f16_plan_equality.py > gpurun_out/r04_c3_f16_equal.log 2>&1; tail -12 gpurun_out/r04_c3_f16_equal.log
for B in 16 2; do
  MI355_PLAN_CACHE=0 MI355_TUNE_LOG=1 timeout -k 10 400 python bench.py --no-configs --no-cpu-baseline --half --model yolov8m --size 1280 --batch $B --chunk $B --steps 10 --warmup 3 \
      > gpurun_out/r04_c3_cfg5_b$B.json 2> gpurun_out/r04_c3_cfg5_b$B.tune.log
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_c3_cfg5_b$B.json").read().strip().splitlines()[-1])
print("cfg5 b$B", d["value"], d["roofline"]["frac"], d["roofline"]["launches_per_step"], d["device_ms_per_step"])
PY
done
grep -c "\[tune\]" gpurun_out/r04_c3_cfg5_b16.tune.log gpurun_out/r04_c3_cfg5_b2.tune.log
