#!/bin/bash
# Round 4, GPU call 29: sweep with the detector pass reading the frames the batched motion-compensation step uploaded -- tests, rate A/B, host profile
cd "$GRAFT_REPO_ROOT"
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_gmc.py -x -q -m gpu > gpurun_out/r04_c29_tests.log 2>&1 || { tail -40 gpurun_out/r04_c29_tests.log; exit 1; }
tail -2 gpurun_out/r04_c29_tests.log
for S in 1 0 1 0; do
  MI355_SWEEP_SHARED_FRAMES=$S timeout -k 10 300 python tools/sweep_profile.py 1280 2>&1 | grep "frames in" | sed "s/^/shared=$S  /"
done
timeout -k 10 300 python tools/sweep_profile.py 1280 2>&1 | grep -v amdgpu.ids | sed -n 2,22p
