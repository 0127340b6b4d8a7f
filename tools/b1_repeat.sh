#!/bin/bash
# batch-1 throughput over fresh autotunes (plan cache off): how much the stopwatch's choices scatter (run via gpurun)
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=0
for r in 1 2 3 4 5; do
python bench.py --no-cpu-baseline --no-configs --model yolov8n --batch 1 --steps 600 --warmup 100 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step')"
done
python bench.py --no-cpu-baseline --no-configs --model yolov8n-pose --batch 32 --steps 60 --warmup 10 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('np32', d['value'], 'fps', d['ms_per_step'], 'ms/step')"
python bench.py --no-cpu-baseline --no-configs --model yolov8s-pose --batch 8 --steps 100 --warmup 20 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sp8', d['value'], 'fps', d['ms_per_step'], 'ms/step')"
