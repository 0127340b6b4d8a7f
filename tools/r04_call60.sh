#!/bin/bash
# Round 4, GPU call 60: config 5 as stated (2 frames per pass, half) -- eager launches vs hipGraph replay, one stream vs four
cd "$GRAFT_REPO_ROOT"
for g in 0 1; do for st in 4 1; do
  MI355_GRAPH=$g MI355_STREAMS=$st timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch 2 --chunk 2 --steps 100 --warmup 20 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph=$g streams=$st', d['value'], 'fps', d['ms_per_step'], 'ms', d['roofline']['frac'])"
done; done
