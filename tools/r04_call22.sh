#!/bin/bash
# Round 4, GPU call 22: model.track with the detector reading the frame the motion-compensation step uploaded -- tests, stage times, timeline, rates
cd "$GRAFT_REPO_ROOT"
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_gmc.py -x -q -m gpu > gpurun_out/r04_c22_tests.log 2>&1 || { tail -40 gpurun_out/r04_c22_tests.log; exit 1; }
tail -2 gpurun_out/r04_c22_tests.log
timeout -k 10 300 python tools/track_stages.py yolov8n 300 2>&1 | grep -v amdgpu.ids | tail -10
timeout -k 10 300 python tools/track_pipeline_bench.py 200 2>&1 | grep -v amdgpu.ids | tail -8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r04_track_tl
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r04_track_tl -- python3 tools/track_stages.py yolov8n 200 > gpurun_out/r04_track_tl.log 2>&1
python tools/track_timeline.py gpurun_out/r04_track_tl
