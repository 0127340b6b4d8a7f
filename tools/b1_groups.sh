#!/bin/bash
# Grouped launches (conv_f32_group.hip) A/B in the latency-bound regime, run ON THE GPU BOX via gpurun.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG:-groups}; mkdir -p $OUT
run() { echo "== $*"; env "$@" MI355_PLAN_CACHE=0 python bench.py --no-cpu-baseline --no-configs $ARGS 2>$OUT/err.log | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], 'TF', d['roofline']['launches_per_step'], 'conv launches', d['device_ms_per_step'])"; }
ARGS="--model yolov8n --batch 1 --chunk 1 --steps 600 --warmup 50"
run MI355_GROUPS=0
run MI355_GROUPS=1
MI355_TUNE_LOG=1 MI355_PLAN_CACHE=0 python bench.py --no-cpu-baseline --no-configs --model yolov8n --batch 1 --chunk 1 --steps 5 --warmup 1 2>&1 | grep "group of" > $OUT/group_decisions_b1.txt
cat $OUT/group_decisions_b1.txt
ARGS="--model yolov8n --batch 4 --chunk 4 --steps 300 --warmup 30"
run MI355_GROUPS=0
run MI355_GROUPS=1
ARGS="--model yolov5mu --batch 1 --chunk 1 --steps 300 --warmup 30"
run MI355_GROUPS=0
run MI355_GROUPS=1
ARGS="--model yolov8n-pose --batch 1 --chunk 1 --steps 300 --warmup 30"
run MI355_GROUPS=0
run MI355_GROUPS=1
