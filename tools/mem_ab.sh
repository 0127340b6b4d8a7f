#!/bin/bash
# Activation-arena sharing A/B on the headline workload and a mid-size batch (run ON THE GPU BOX via gpurun).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG:-memab}; mkdir -p $OUT
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-configs $ARGS 2>$OUT/err.log | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], 'TF', d['roofline']['launches_per_step'], 'launches', round(d['config']['activation_bytes_per_gpu']/2**30,2), 'GiB')"; }
ARGS="--steps 20 --warmup 3"
run MI355_MEM_REUSE=0
run MI355_MEM_REUSE=1
ARGS="--model yolov8n-pose --batch 32 --chunk 32 --steps 60 --warmup 5"
run MI355_MEM_REUSE=0
run MI355_MEM_REUSE=1
ARGS="--model yolov8s-pose --batch 64 --chunk 64 --steps 20 --warmup 3"
run MI355_MEM_REUSE=0
run MI355_MEM_REUSE=1
