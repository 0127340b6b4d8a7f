#!/bin/bash
# upsample-on-read vs upsample kernel, decided by the autotuner (MI355_UPSAMPLE_TUNE=0: always on read); run via gpurun
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=0
for args in "--model yolov8n-pose --batch 32 --steps 60 --warmup 10" "--model yolov8n --batch 1 --steps 600 --warmup 100" "--model yolov8s-pose --batch 8 --steps 100 --warmup 20"; do
for rep in 1 2; do for t in 1 0; do
echo "== UPSAMPLE_TUNE=$t $args"
MI355_UPSAMPLE_TUNE=$t python bench.py --no-cpu-baseline --no-configs $args 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], d['roofline']['unit'])"
done; done; done
