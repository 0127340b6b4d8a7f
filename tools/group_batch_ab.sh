#!/bin/bash
# Step schedule + grouped launches beyond 5 frames per pass (MI355_GROUP_MAX_BATCH): s-pose batch 8, n-pose batch 32, YOLOv8n batch 8 / 16
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=0
run() { python bench.py --no-cpu-baseline --no-configs "$@" 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('max_batch=$MI355_GROUP_MAX_BATCH', '$*', '|', d['value'], 'fps', d['roofline']['launches_per_step'], 'launches')"; }
for mb in 5 64 5 64; do
  export MI355_GROUP_MAX_BATCH=$mb
  run --model yolov8s-pose --batch 8 --chunk 8 --steps 200 --warmup 20
  run --model yolov8n-pose --batch 32 --chunk 32 --steps 100 --warmup 10
  run --model yolov8n --batch 8 --chunk 8 --steps 300 --warmup 20
done
