#!/bin/bash
# what-if builds of the stem (tools/ab_build.sh "-DMI355_STEM_EXP=<flags>" stem<flags>): stem_ms per step, fp32 batch 512 and config 5
cd "$GRAFT_REPO_ROOT"
for name in product stem1 stem2 stem4 stem8 stem15; do
  if [ $name = product ]; then unset MI355_YOLO_LIB; else export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_$name.so; fi
  echo "== $name"
  python bench.py --no-cpu-baseline --no-configs --steps 6 --warmup 2 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 b512 stem_ms', d['device_ms_per_step']['stem_ms'], 'fps', d['value'])"
  python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch 16 --steps 10 --warmup 3 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 stem_ms', d['device_ms_per_step']['stem_ms'], 'fps', d['value'])"
done
