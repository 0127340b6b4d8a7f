#!/bin/bash
# Batch-1 A/B of the one-launch NMS (sort + greedy pass; MI355_NMS_FUSED=1, default) against two launches (=0); one plan file for all runs
run() { python bench.py --no-cpu-baseline --no-configs "$@" 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused=$MI355_NMS_FUSED', '$*', '|', d['value'], 'fps', d['ms_per_step'], 'ms/step')"; }
export MI355_NMS_FUSED=1
run --model yolov8n --batch 1 --chunk 1 --steps 500 --warmup 50 > /dev/null     # tunes and writes the plan file
for v in 1 0 1 0 1 0; do
  export MI355_NMS_FUSED=$v
  run --model yolov8n --batch 1 --chunk 1 --steps 6000 --warmup 200
done
