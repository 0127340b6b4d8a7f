#!/bin/bash
# host-fed (pinned memory, PCIe-inclusive) rate of the headline workload by engine chunk size (run via gpurun)
cd "$GRAFT_REPO_ROOT"
for c in 32 64 128 256; do
python bench.py --no-cpu-baseline --no-configs --host-frames --chunk $c --steps 6 --warmup 2 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk $c', d['value'], 'fps', d['ms_per_step'], 'ms/step')"
done
