#!/bin/bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG:-memab}; mkdir -p $OUT
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-configs $ARGS 2>$OUT/err.log | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'fps', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], 'TF', d['roofline']['launches_per_step'], 'launches', round(d['config']['activation_bytes_per_gpu']/2**30,2), 'GiB')"; }
ARGS="--steps 20 --warmup 3"
run MI355_MEM_REUSE=0
run MI355_MEM_REUSE=1
run MI355_MEM_REUSE=1 MI355_ARENA_ALIGN=4096
run MI355_MEM_REUSE=1 MI355_ARENA_ALIGN=2097152
run MI355_MEM_REUSE=0 MI355_ARENA_ALIGN=2097152
run MI355_MEM_REUSE=1 MI355_STREAMS=1
run MI355_MEM_REUSE=0 MI355_STREAMS=1
