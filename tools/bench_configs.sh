#!/bin/bash
# BASELINE.json's configurations on one GPU: bench.py with other --model / --size / --batch.
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '|', d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], 'TF', d['roofline']['frac'])"; }
run --model yolov8n --batch 1 --chunk 1 --steps 300 --warmup 20
run --model yolov8n --batch 512 --chunk 512
run --model yolov8n --batch 512 --chunk 512 --half
run --model yolov8n-pose --batch 32 --chunk 32 --steps 60 --warmup 5
run --model yolov8n-pose --batch 512 --chunk 512
run --model yolov8s-pose --batch 8 --chunk 8 --steps 60 --warmup 5
run --model yolov8s-pose --batch 64 --chunk 64 --steps 40 --warmup 5
run --model yolov8s-pose --batch 256 --chunk 256
run --model yolov8m --size 1280 --batch 16 --chunk 16 --steps 10 --warmup 2
run --model yolov8m --size 1280 --batch 16 --chunk 16 --steps 10 --warmup 2 --half
run --model yolov8m --size 1280 --batch 2 --chunk 2 --steps 40 --warmup 4 --half
run --model yolov5mu --batch 256 --chunk 256
