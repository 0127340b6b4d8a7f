#!/bin/bash
# Round 4, GPU call 54: what-if (diag build): conv3x3_lw_f16 without the epilogue's bias load (MI355_F16_EXP=64) on three config-5 shapes
cd "$GRAFT_REPO_ROOT"
export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_f16diag.so MI355_BENCH_HALF=1 MB_TOP=1 MB_FILTER=v7
for e in 0 64 0 64; do
  echo "== MI355_F16_EXP=$e"
  MI355_F16_EXP=$e timeout -k 10 120 python tools/conv_microbench.py 16 320 320 48 48 3 1 2>&1 | sed -n 3p
  MI355_F16_EXP=$e timeout -k 10 120 python tools/conv_microbench.py 16 160 160 96 96 3 1 2>&1 | sed -n 3p
  MI355_F16_EXP=$e timeout -k 10 120 python tools/conv_microbench.py 16 80 80 192 192 3 1 2>&1 | sed -n 3p
done
