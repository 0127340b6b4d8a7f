#!/bin/bash
# headline / n-pose b32 / s-pose b64 / s-pose b8 over the product library and any number of tools/ab_build.sh variants, each tuning
# its own plans (run via gpurun):  bash tools/ab_multi.sh name1 name2 ...
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=0
for name in product "$@"; do
  if [ $name = product ]; then unset MI355_YOLO_LIB; else export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_$name.so; fi
  echo "== $name"
  for args in "--steps 10 --warmup 3" "--model yolov8n-pose --batch 32 --steps 60 --warmup 10" "--model yolov8s-pose --batch 64 --steps 20 --warmup 5" "--model yolov8s-pose --batch 8 --steps 100 --warmup 20"; do
    python bench.py --no-cpu-baseline --no-configs $args 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['roofline']['achieved'], d['roofline']['unit'], end='   ')"
  done
  echo
done
