#!/bin/bash
# Round 4, GPU call 27: what-if runs of the half-mode POINTWISE convs that carry config 5 (diagnostic build): staging / stores / weight loads / LDS reads removed
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=0 MI355_PLAN_DIR=$PWD/gpurun_out/empty_dir; mkdir -p gpurun_out/empty_dir
export MI355_BENCH_HALF=1 MB_TOP=2
for SH in "16 160 160 576 192" "16 320 320 192 96" "16 80 80 1152 384"; do
  echo "######## $SH (product library, all plans)"
  timeout -k 10 200 python tools/conv_microbench.py $SH 1 1 1 0 2>&1 | grep -v amdgpu.ids | head -6
  export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_exp.so
  for ex in 0 1 4 5 8 16 29; do
    echo "## EXP=$ex"
    MB_FILTER=v1 MI355_F16_EXP=$ex timeout -k 10 200 python tools/conv_microbench.py $SH 1 1 1 0 2>&1 | grep "v1" | head -2
  done
  unset MI355_YOLO_LIB
done
