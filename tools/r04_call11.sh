#!/bin/bash
# Round 4, GPU call 11: what-if runs of the version-7 kernel (diagnostic build): which part of an item the MFMAs wait for
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=0 MI355_PLAN_DIR=$PWD/gpurun_out/empty_dir; mkdir -p gpurun_out/empty_dir
export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_exp.so MI355_BENCH_HALF=1 MB_TOP=3 MB_FILTER=v7
for ex in 0 1 3 4 8 16 24 32 35 63; do
  echo "#### EXP=$ex"
  MI355_F16_EXP=$ex timeout -k 10 120 python tools/conv_microbench.py 16 80 80 192 192 3 1 1 0 2>&1 | grep "v7" | head -3
  MI355_F16_EXP=$ex timeout -k 10 120 python tools/conv_microbench.py 16 320 320 48 48 3 1 1 0 2>&1 | grep "v7" | head -1
done
