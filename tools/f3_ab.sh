#!/bin/bash
# C2f-tail fusion (MI355_FUSE_TAIL) A/B, run ON THE GPU BOX via gpurun.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG:-f3}; mkdir -p $OUT
run() { echo "== $*"; env "$@" MI355_PLAN_CACHE=0 python bench.py --no-cpu-baseline --no-configs $ARGS 2>$OUT/err.log | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], 'TF', d['roofline']['launches_per_step'], 'conv launches')"; }
ARGS="--steps 12 --warmup 3"
run MI355_FUSE_TAIL=0
run MI355_FUSE_TAIL=1
MI355_TUNE_LOG=1 MI355_PLAN_CACHE=0 python bench.py --no-cpu-baseline --no-configs --steps 2 --warmup 1 2>&1 | grep "fused .* vs separate" > $OUT/fuse_decisions_b512.txt; cat $OUT/fuse_decisions_b512.txt
ARGS="--model yolov8n --batch 1 --chunk 1 --steps 600 --warmup 50"
run MI355_FUSE_TAIL=0
run MI355_FUSE_TAIL=1
ARGS="--model yolov8n-pose --batch 32 --chunk 32 --steps 60 --warmup 8"
run MI355_FUSE_TAIL=0
run MI355_FUSE_TAIL=1
ARGS="--model yolov8s-pose --batch 64 --chunk 64 --steps 16 --warmup 3"
run MI355_FUSE_TAIL=0
run MI355_FUSE_TAIL=1
