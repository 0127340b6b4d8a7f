#!/bin/bash
# Round 4, GPU call 8: LDS pixel-stride A/B (MI355_LDS_PAD=4: stride 16 mod 64 bytes, two passes per B-operand ds_read_b128; 8: stride 32 mod 64,
# conflict-free) on the four 3x3 shapes that carry config 5, two fp32 shapes, config 5 at batch 16 and the fp32 headline.
cd "$GRAFT_REPO_ROOT"
set -o pipefail
export MI355_PLAN_CACHE=0 MI355_PLAN_DIR=$PWD/gpurun_out/empty_dir; mkdir -p gpurun_out/empty_dir
for PAD in 4 8; do
  export MI355_LDS_PAD=$PAD
  echo "=== MI355_LDS_PAD=$PAD"
  MI355_BENCH_HALF=1 MB_TOP=3 timeout -k 10 200 python tools/conv_microbench.py 16 80 80 192 192 3 1 1 0 2>&1 | grep -v amdgpu.ids
  MI355_BENCH_HALF=1 MB_TOP=3 timeout -k 10 200 python tools/conv_microbench.py 16 160 160 96 96 3 1 1 0 2>&1 | grep -v amdgpu.ids
  MI355_BENCH_HALF=1 MB_TOP=3 timeout -k 10 200 python tools/conv_microbench.py 16 320 320 48 48 3 1 1 0 2>&1 | grep -v amdgpu.ids
  MI355_BENCH_HALF=1 MB_TOP=3 timeout -k 10 200 python tools/conv_microbench.py 16 40 40 288 288 3 1 1 0 2>&1 | grep -v amdgpu.ids
  MB_TOP=3 timeout -k 10 200 python tools/conv_microbench.py 512 40 40 128 128 3 1 1 0 2>&1 | grep -v amdgpu.ids
  MB_TOP=3 timeout -k 10 200 python tools/conv_microbench.py 512 80 80 64 64 3 1 1 0 2>&1 | grep -v amdgpu.ids
done
for PAD in 4 8 4 8; do
  export MI355_LDS_PAD=$PAD
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch 16 --chunk 16 --steps 30 --warmup 5 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 b16 pad=$PAD', d['value'], 'fps', d['roofline']['frac'])"
done
for PAD in 4 8; do
  export MI355_LDS_PAD=$PAD
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline pad=$PAD', d['value'], 'fps', d['roofline']['frac'])"
done
