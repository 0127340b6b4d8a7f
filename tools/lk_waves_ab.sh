#!/bin/bash
# Track-loop A/B of the Lucas-Kanade block size (points per block = waves per block): product (8) against 4 and 16 (tools/ab_build.sh "-DMI355_LK_WAVES=n" lkn)
cd "$GRAFT_REPO_ROOT"
for name in product lk4 lk16 product lk4 lk16; do
  if [ $name = product ]; then unset MI355_YOLO_LIB; else export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_$name.so; fi
  echo "== $name"; python tools/track_pipeline_bench.py 2>&1 | grep "GPU GMC"
done
