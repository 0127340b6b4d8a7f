#!/bin/bash
# Round 4, GPU call 13: version-7 kernel with the next item's global loads issued between the micro-steps of compute() (A/B against the
# loads as a phase of their own), parity first
cd "$GRAFT_REPO_ROOT"
set -o pipefail
export MI355_PLAN_CACHE=0 MI355_PLAN_DIR=$PWD/gpurun_out/empty_dir; mkdir -p gpurun_out/empty_dir
timeout -k 10 600 python -m pytest tests/test_gpu_half.py -x -q -m gpu -k "every_plan_against_float64 or bits_do_not_depend" > gpurun_out/r04_c13_tests.log 2>&1 || { tail -40 gpurun_out/r04_c13_tests.log; exit 1; }
tail -2 gpurun_out/r04_c13_tests.log
export MI355_BENCH_HALF=1 MB_TOP=3 MB_FILTER=v7
for LIB in "" noint "" noint; do
  if [ -n "$LIB" ]; then export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_$LIB.so; else unset MI355_YOLO_LIB; fi
  echo "#### lib=${LIB:-product}"
  for SH in "16 80 80 192 192" "16 160 160 96 96" "16 320 320 48 48" "16 160 160 192 256"; do
    timeout -k 10 120 python tools/conv_microbench.py $SH 3 1 1 0 2>&1 | grep "v7" | head -1
  done
done
