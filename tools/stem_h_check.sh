#!/bin/bash
# half=True stem with fp16 operands (default) vs the fp32-operand stem (MI355_STEM_F16=0): tests and config 5 (run via gpurun)
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_half.py -m gpu -x -q 2>&1 | tail -3
for v in 1 0; do for b in 16 2; do
MI355_STEM_F16=$v python bench.py --no-cpu-baseline --no-configs --model yolov8m --size 1280 --half --batch $b --steps 30 --warmup 5 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('STEM_F16=$v b$b', d['value'], 'fps', d['ms_per_step'], 'stem_ms', d['device_ms_per_step']['stem_ms'])"
done; done
