#!/bin/bash
# Round 4, GPU call 5: stream-priority A/B of the track loop, multi-stream A/B of config 5 at batch 2, then the whole GPU suite.
cd "$GRAFT_REPO_ROOT"
set -o pipefail
bash tools/track_prio_ab.sh > gpurun_out/r04_c5_prio.log 2>&1; grep -A9 "^==" gpurun_out/r04_c5_prio.log | grep "==\|predict\|sum\|gmc_collect"
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_c5; mkdir -p $MI355_PLAN_CACHE
for SMB in 6 1 6 1; do
  MI355_STREAMS_MIN_BATCH=$SMB timeout -k 10 300 python bench.py --no-configs --no-cpu-baseline --half --model yolov8m --size 1280 --batch 2 --chunk 2 --steps 40 --warmup 5 \
      > gpurun_out/r04_c5_cfg5_b2_smb$SMB.json 2> gpurun_out/r04_c5_cfg5_b2_smb$SMB.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_c5_cfg5_b2_smb$SMB.json").read().strip().splitlines()[-1])
print("cfg5 b2 streams_min_batch=$SMB", d["value"], d["roofline"]["frac"], d["roofline"]["plan_source"])
PY
done
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_c5_tests.log 2>&1 || { tail -40 gpurun_out/r04_c5_tests.log; exit 1; }
tail -3 gpurun_out/r04_c5_tests.log
