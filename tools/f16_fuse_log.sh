#!/bin/bash
# config 5 with the autotuner's fused-vs-separate decisions printed (run via gpurun)
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=0
for b in 16 2; do
MI355_TUNE_LOG=1 python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch $b --steps 30 --warmup 5 2>gpurun_out/tune.log | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], d['roofline']['unit'])"
grep "fused\b.*vs separate\|upsample on read" gpurun_out/tune.log
MI355_FUSE_1X1=0 python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch $b --steps 30 --warmup 5 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FUSE_1X1=0:', d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], d['roofline']['unit'])"
done
