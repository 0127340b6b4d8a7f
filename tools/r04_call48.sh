#!/bin/bash
# Round 4, GPU call 48: flat staging of wide halo rows in conv_igemm_f16 (stride-2 tiles): half tests, the s2 shapes with and without, config 5 at batch 16 / 2
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_half.py -x -q -m gpu > gpurun_out/r04_c48_tests.log 2>&1; tail -2 gpurun_out/r04_c48_tests.log
export MI355_BENCH_HALF=1 MB_TOP=2
for f in 1 0; do
  echo "== MI355_F16_FLAT_STAGE=$f"
  MI355_F16_FLAT_STAGE=$f MB_FILTER=v1 timeout -k 10 200 python tools/conv_microbench.py 16 640 640 48 96 3 2 1 0 40 2>&1 | tail -n +2 | head -2
  MI355_F16_FLAT_STAGE=$f MB_FILTER=v1 timeout -k 10 200 python tools/conv_microbench.py 16 320 320 96 192 3 2 1 0 40 2>&1 | tail -n +2 | head -2
  MI355_F16_FLAT_STAGE=$f MB_FILTER=v1 timeout -k 10 200 python tools/conv_microbench.py 16 160 160 192 384 3 2 1 0 40 2>&1 | tail -n +2 | head -2
done
unset MI355_BENCH_HALF
for f in 1 0; do for b in 16 2; do
  MI355_F16_FLAT_STAGE=$f timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch $b --chunk $b --steps 30 --warmup 5 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flat=$f b$b', d['value'], 'fps', d['ms_per_step'], 'ms', d['config'].get('plan_source'))"
done; done
