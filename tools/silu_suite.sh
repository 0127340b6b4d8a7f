cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -3
export MI355_PLAN_CACHE=0
for args in "--steps 10 --warmup 3" "--steps 10 --warmup 3" "--model yolov8n-pose --batch 32 --steps 60 --warmup 10" "--model yolov8s-pose --batch 64 --steps 20 --warmup 5" "--model yolov8n --batch 1 --steps 600 --warmup 100"; do
python bench.py --no-cpu-baseline --no-configs $args 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], d['roofline']['unit'], d['device_ms_per_step']['stem_ms'])"
done
