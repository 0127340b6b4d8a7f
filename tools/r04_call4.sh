#!/bin/bash
# Round 4, GPU call 4: batched GMC -- tests, then the track pipeline's rates again.
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_tmp; mkdir -p $MI355_PLAN_CACHE
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_gmc.py tests/test_gpu_pipeline.py tests/test_gpu_half.py -x -q -m gpu > gpurun_out/r04_c4_tests.log 2>&1 || { tail -40 gpurun_out/r04_c4_tests.log; exit 1; }
tail -3 gpurun_out/r04_c4_tests.log
timeout -k 10 300 python tools/track_stages.py yolov8s-pose 100 > gpurun_out/r04_c4_stages_spose.log 2>&1; tail -9 gpurun_out/r04_c4_stages_spose.log
timeout -k 10 900 python - > gpurun_out/r04_c4_trackpipe.json 2> gpurun_out/r04_c4_trackpipe.err <<'PY'
import json, bench
print(json.dumps(bench.track_pipeline()))
PY
cat gpurun_out/r04_c4_trackpipe.json
