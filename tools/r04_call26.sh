#!/bin/bash
# Round 4, GPU call 26: final build -- the whole GPU suite, then the driver's command (profiles/r04_bench_driver_style.json)
cd "$GRAFT_REPO_ROOT"
set -o pipefail
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/r04_c26_tests.log 2>&1 || { tail -40 gpurun_out/r04_c26_tests.log; exit 1; }
tail -3 gpurun_out/r04_c26_tests.log
bash tools/r04_call20.sh
