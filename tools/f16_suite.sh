#!/bin/bash
# half=True checks after a kernel change (run via gpurun): plan tables of the three 3x3 shapes that carry YOLOv8m 1280x1280, then config 5
cd "$GRAFT_REPO_ROOT"
export MI355_PLAN_CACHE=0
MI355_BENCH_HALF=1 MB_TOP=8 python tools/conv_microbench.py 16 80 80 192 192 3 1 1 0 2>&1 | grep -v amdgpu.ids
MI355_BENCH_HALF=1 MB_TOP=8 python tools/conv_microbench.py 16 160 160 96 96 3 1 1 0 2>&1 | grep -v amdgpu.ids
MI355_BENCH_HALF=1 MB_TOP=8 python tools/conv_microbench.py 16 160 160 192 256 3 1 1 0 2>&1 | grep -v amdgpu.ids
MI355_BENCH_HALF=1 MB_TOP=8 python tools/conv_microbench.py 16 320 320 96 192 3 2 1 0 2>&1 | grep -v amdgpu.ids
for b in 16 2; do
python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch $b --steps 30 --warmup 5 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], d['roofline']['unit'])"
done
