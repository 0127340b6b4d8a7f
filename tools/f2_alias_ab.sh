#!/bin/bash
# Fused Conv3x3 -> Conv1x1 launches with the first conv's image in the halo tile's LDS (MI355_F2_ALIAS=1, default) against behind it (=0)
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=0
run() { python bench.py --no-cpu-baseline --no-configs "$@" 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('alias=$MI355_F2_ALIAS', '$*', '|', d['value'], 'fps', d['roofline']['achieved'], d['roofline']['unit'])"; }
for v in 1 0 1 0; do
  export MI355_F2_ALIAS=$v
  run --steps 20 --warmup 5
  run --model yolov8m --size 1280 --half --batch 16 --steps 30 --warmup 5
  run --model yolov8s-pose --batch 64 --chunk 64 --steps 40 --warmup 5
done
