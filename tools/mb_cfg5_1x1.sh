#!/bin/bash
# launch-plan tables of the pointwise layers of YOLOv8m 1280x1280 half=True, batch 16 (run via gpurun)
cd "$GRAFT_REPO_ROOT"
export MI355_BENCH_HALF=1 MB_TOP=14
python tools/conv_microbench.py 16 320 320 96 96 1 1 1 0
python tools/conv_microbench.py 16 320 320 192 96 1 1 1 0
python tools/conv_microbench.py 16 160 160 576 192 1 1 1 0
python tools/conv_microbench.py 16 80 80 1152 384 1 1 1 0
python tools/conv_microbench.py 16 80 80 384 384 1 1 1 0
