#!/bin/bash
# Round 4, GPU call 33: config-5 profiles on the final half-mode plans, then the driver's command
cd "$GRAFT_REPO_ROOT"
bash tools/r04_call19b.sh > gpurun_out/r04_c33_collect.log 2>&1; tail -4 gpurun_out/r04_c33_collect.log
bash tools/r04_call20.sh
