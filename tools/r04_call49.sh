#!/bin/bash
# Round 4, GPU call 49: the k3 s2 fp32 stem (stem3s2_u8_f32): op tests, end-to-end bit-exactness, headline / fast_act / batch-1 rates with and without
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q -m gpu -k "stem or bit_exact" > gpurun_out/r04_c49_tests.log 2>&1; tail -2 gpurun_out/r04_c49_tests.log
for l in 1 0; do
  MI355_STEM_LEAN=$l timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --steps 6 --warmup 2 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lean=$l headline', d['value'], 'fps', d['ms_per_step'], 'ms', 'stem_ms', d['device_ms_per_step']['stem_ms'], d['roofline']['frac'])"
  MI355_FAST_ACT=1 MI355_STEM_LEAN=$l timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --steps 6 --warmup 2 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lean=$l fast_act', d['value'], 'fps', d['ms_per_step'], 'ms', 'stem_ms', d['device_ms_per_step']['stem_ms'], d['roofline']['frac'])"
  MI355_STEM_LEAN=$l timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs --batch 1 --chunk 1 --steps 400 --warmup 50 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lean=$l batch1', d['value'], 'fps', d['ms_per_step'], 'ms')"
done
