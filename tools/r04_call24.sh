#!/bin/bash
# Round 4, GPU call 24: collect worker after the race fix (a step taken by the worker but not yet done was collected a second time by the caller):
# eight runs of the frame loop, the tracker / pipeline tests, bench's track_pipeline
cd "$GRAFT_REPO_ROOT"
set -o pipefail
for R in 1 2 3 4 5 6 7 8; do
  timeout -k 10 200 python tools/track_stages.py yolov8n 400 2>&1 | grep "sum \|Error\|error" | head -2
done
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_gmc.py tests/test_gpu_e2e.py -x -q -m gpu > gpurun_out/r04_c24_tests.log 2>&1 || { tail -40 gpurun_out/r04_c24_tests.log; exit 1; }
tail -2 gpurun_out/r04_c24_tests.log
timeout -k 10 900 python - > gpurun_out/r04_c24_trackpipe.json 2> gpurun_out/r04_c24_trackpipe.err <<'PY'
import json, bench
print(json.dumps(bench.track_pipeline()))
PY
cat gpurun_out/r04_c24_trackpipe.json
