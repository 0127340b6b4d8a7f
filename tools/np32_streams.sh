#!/bin/bash
# n-pose batch 32 / s-pose batch 8: one in-order stream vs the DAG schedule on 4 / 8 streams (run via gpurun)
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; for r in 1 2; do env "$@" python bench.py --no-cpu-baseline --no-configs $ARGS 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step')"; done; }
ARGS="--model yolov8n-pose --batch 32 --steps 80 --warmup 10"
run MI355_STREAMS=1
run MI355_STREAMS=4
run MI355_STREAMS=8
run MI355_STREAMS=2
ARGS="--model yolov8s-pose --batch 8 --steps 100 --warmup 20"
run MI355_STREAMS=1
run MI355_STREAMS=4
