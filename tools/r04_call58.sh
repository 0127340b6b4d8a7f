#!/bin/bash
# Round 4, GPU call 58: config 5 at frames-per-pass values whose tile counts fill the persistent kernels' 512 block slots differently (quantisation of the last round)
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=$PWD/gpurun_out/r04_plans_b; mkdir -p $MI355_PLAN_CACHE
for b in 16 20 15 10 5; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch $b --chunk $b --steps 30 --warmup 5 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b$b', d['value'], 'fps', d['ms_per_step'], 'ms', d['roofline']['frac'], d['roofline'].get('plan_source'))"
done
