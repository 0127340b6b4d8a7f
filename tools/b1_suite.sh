#!/bin/bash
# Small-batch latency experiments (run ON THE GPU BOX via gpurun): batch-1 / batch-32 throughput under the engine's knobs.
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline $ARGS 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], 'TF', d['device_ms_per_step'])"; }
ARGS="--model yolov8n --batch 1 --chunk 1 --steps 400 --warmup 30"
run MI355_SMALL_PT=0
run MI355_SMALL_PT=1
run MI355_SMALL_PT=1 MI355_GRAPH=1
run MI355_SMALL_PT=1 MI355_STREAMS=1
run MI355_SMALL_PT=1 MI355_STREAMS=1 MI355_GRAPH=1
run MI355_SMALL_PT=1 MI355_STREAMS=8 MI355_GRAPH=1
ARGS="--model yolov8n-pose --batch 32 --chunk 32 --steps 60 --warmup 5"
run MI355_SMALL_PT=0
run MI355_SMALL_PT=1
run MI355_SMALL_PT=1 MI355_GRAPH=1
