#!/bin/bash
# Round 4, GPU call 2: tracker core in C++ -- GPU GMC tests, the track pipeline's rates and its host-side stage times.
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_tmp; mkdir -p $MI355_PLAN_CACHE
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_gmc.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r04_c2_tests.log 2>&1 || { tail -40 gpurun_out/r04_c2_tests.log; exit 1; }
tail -3 gpurun_out/r04_c2_tests.log
timeout -k 10 300 python tools/track_stages.py yolov8n 300 > gpurun_out/r04_c2_stages.log 2>&1; cat gpurun_out/r04_c2_stages.log | tail -12
timeout -k 10 300 python tools/track_pipeline_bench.py 200 > gpurun_out/r04_c2_track.log 2>&1; tail -6 gpurun_out/r04_c2_track.log
timeout -k 10 600 python - > gpurun_out/r04_c2_trackpipe.json 2> gpurun_out/r04_c2_trackpipe.err <<'PY'
import json, bench
print(json.dumps(bench.track_pipeline()))
PY
cat gpurun_out/r04_c2_trackpipe.json
