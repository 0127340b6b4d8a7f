#!/bin/bash
# Round 4, GPU call 46: what-ifs (diag build) of the low-channel half-mode layers of config 5: model.2.m.*.cv1 (48->48 @320, v7 and v1 plans) and model.1 without its fused tail (48->96 s2)
cd "$GRAFT_REPO_ROOT"
export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_f16diag.so MI355_BENCH_HALF=1 MB_TOP=3
for e in 0 1 4 5 16 8 32; do
  echo "== MI355_F16_EXP=$e: 48->48 k3 s1 @320 (v7)"; MI355_F16_EXP=$e MB_FILTER=v7 timeout -k 10 120 python tools/conv_microbench.py 16 320 320 48 48 3 1 2>&1 | tail -n +2 | head -2
done
for e in 0 1 4 5; do
  echo "== MI355_F16_EXP=$e: 48->48 k3 s1 @320 (v1)"; MI355_F16_EXP=$e MB_FILTER=v1 timeout -k 10 200 python tools/conv_microbench.py 16 320 320 48 48 3 1 1 0 40 2>&1 | tail -n +2 | head -2
  echo "== MI355_F16_EXP=$e: 48->96 k3 s2 @640 (v1)"; MI355_F16_EXP=$e MB_FILTER=v1 timeout -k 10 200 python tools/conv_microbench.py 16 640 640 48 96 3 2 1 0 40 2>&1 | tail -n +2 | head -2
done
