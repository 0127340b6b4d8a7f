#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the three rocprofv3 passes the committed profiles/ summaries come from.
#   kernel trace + stats, PMC FETCH_SIZE, PMC WRITE_SIZE -- counters in their own runs, as the guide prescribes.
set -e
TAG=${1:-r01_v3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_write.log 2>&1
ls gpurun_out/${TAG}_*/*/ | head -20
