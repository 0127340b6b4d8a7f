#!/bin/bash
# Batch-1 host hand-over A/B: rows and counts written by the greedy NMS kernel straight into pinned host memory
# (MI355_DIRECT_ROWS=1, default) against compaction + copy-engine copies (=0).  Same plan file for all runs.
run() { python bench.py --no-cpu-baseline --no-configs "$@" 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$MI355_DIRECT_ROWS', '$*', '|', d['value'], 'fps', d['ms_per_step'], 'ms/step')"; }
for v in 1 0 1 0; do
  export MI355_DIRECT_ROWS=$v
  run --model yolov8n --batch 1 --chunk 1 --steps 3000 --warmup 100
done
for v in 1 0; do
  export MI355_DIRECT_ROWS=$v
  run --model yolov5mu --batch 1 --chunk 1 --steps 1000 --warmup 50
  run --model yolov8s-pose --batch 8 --chunk 8 --steps 300 --warmup 20
done
