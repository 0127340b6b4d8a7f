#!/bin/bash
# Round 4, GPU call 23: collect half of a motion-compensation step on a worker thread beside the detector pass -- tests, stage times, A/B, bench's track_pipeline
cd "$GRAFT_REPO_ROOT"
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_gmc.py -x -q -m gpu > gpurun_out/r04_c23_tests.log 2>&1 || { tail -40 gpurun_out/r04_c23_tests.log; exit 1; }
tail -2 gpurun_out/r04_c23_tests.log
for A in 1 0 1 0; do
  echo "== MI355_GMC_ASYNC=$A"
  MI355_GMC_ASYNC=$A timeout -k 10 300 python tools/track_stages.py yolov8n 300 2>&1 | grep -v amdgpu.ids | tail -9 | head -7
done
timeout -k 10 900 python - > gpurun_out/r04_c23_trackpipe.json 2> gpurun_out/r04_c23_trackpipe.err <<'PY'
import json, bench
print(json.dumps(bench.track_pipeline()))
PY
cat gpurun_out/r04_c23_trackpipe.json
