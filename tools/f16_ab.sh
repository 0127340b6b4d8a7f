#!/bin/bash
# half=True A/B over several experimental builds (tools/ab_build.sh FLAGS NAME): config 5 at batch 16 and 2, each build tuning its own plans
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=0
for name in product "$@"; do
  if [ $name = product ]; then unset MI355_YOLO_LIB; else export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_$name.so; fi
  echo "== $name"
  for b in 16 2; do
    python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch $b --steps 30 --warmup 5 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], d['roofline']['unit'])"
  done
done
