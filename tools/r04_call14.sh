#!/bin/bash
# Round 4, GPU call 14: version-7 kernel A/B -- s_setprio(1) around the MFMA phase; weight-fragment ring 3 micro-steps ahead
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=0 MI355_PLAN_DIR=$PWD/gpurun_out/empty_dir; mkdir -p gpurun_out/empty_dir
export MI355_BENCH_HALF=1 MB_TOP=3 MB_FILTER=v7
for LIB in "" prio wr4 "" prio wr4; do
  if [ -n "$LIB" ]; then export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_$LIB.so; else unset MI355_YOLO_LIB; fi
  echo "#### lib=${LIB:-product}"
  for SH in "16 80 80 192 192" "16 160 160 96 96" "16 320 320 48 48" "16 160 160 192 256"; do
    timeout -k 10 120 python tools/conv_microbench.py $SH 3 1 1 0 2>&1 | grep "v7" | head -1
  done
done
