#!/bin/bash
# Round 4, GPU call 21: timeline of the frame-by-frame track loop (kernel + memory-copy trace): do the motion-compensation kernels overlap the detector pass?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r04_track_tl
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r04_track_tl -- python3 tools/track_stages.py yolov8n 200 > gpurun_out/r04_track_tl.log 2>&1
tail -12 gpurun_out/r04_track_tl.log | grep -v "^[EW]2026"
python tools/track_timeline.py gpurun_out/r04_track_tl
ls -la gpurun_out/r04_track_tl/*/ | head
