#!/bin/bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG:-b1k}; mkdir -p $OUT
run() { echo "== $A :: $*"; env "$@" MI355_PLAN_CACHE=0 python bench.py --no-cpu-baseline --no-configs $ARGS 2>$OUT/err.log | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['launches_per_step'], 'conv launches')"; }
for A in "yolov8n 1" "yolov8n 4" "yolov8n-pose 1" "yolov5mu 1" "yolov8s-pose 2"; do
  set -- $A
  ARGS="--model $1 --batch $2 --chunk $2 --steps 500 --warmup 50"
  run MI355_X=0
  run MI355_X=0
  run MI355_CONV_V4=0
done
