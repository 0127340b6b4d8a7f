cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -q -m gpu -x 2>&1 | tail -3
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-configs $ARGS 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], 'TF', d['device_ms_per_step'])"; }
ARGS="--steps 10 --warmup 3"
run MI355_SMALL_PT=1
ARGS="--model yolov8n --batch 1 --steps 400 --warmup 30"
run MI355_SMALL_PT=1
ARGS="--model yolov8n-pose --batch 32 --steps 60 --warmup 5"
run MI355_SMALL_PT=1
