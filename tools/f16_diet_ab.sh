#!/bin/bash
# half=True instruction-diet A/B: product library against a build with -DMI355_F16_DIET=0
# (tools/ab_build.sh "-DMI355_F16_DIET=0" nodiet): the shapes that carry YOLOv8m 1280x1280, then config 5 at batch 16 and 2.
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=0
for name in product nodiet; do
  if [ $name = product ]; then unset MI355_YOLO_LIB; else export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_$name.so; fi
  echo "== $name"
  for shp in "16 320 320 48 48 3 1 1 0" "16 320 320 48 48 3 1 1 1" "16 160 160 96 96 3 1 1 0" "16 80 80 192 192 3 1 1 0" "16 320 320 192 96 1 1 1 0" "16 160 160 576 192 1 1 1 0" "16 640 640 48 96 3 2 1 0"; do
    MI355_BENCH_HALF=1 MB_TOP=3 python tools/conv_microbench.py $shp 2>&1 | grep -v amdgpu.ids
  done
  for b in 16 2; do
    python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch $b --steps 30 --warmup 5 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 b$b', d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], d['roofline']['unit'])"
  done
done
