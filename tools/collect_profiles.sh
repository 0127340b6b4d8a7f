#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the rocprofv3 passes the committed profiles/ summaries come from.
#   kernel trace + stats, PMC FETCH_SIZE, PMC WRITE_SIZE (+ MFMA-busy when MFMA=1) -- every counter set in its own run
#   with --kernel-trace only, as the guide prescribes.
# usage: collect_profiles.sh TAG [bench.py flags ...]     e.g.  collect_profiles.sh r01_cfg5 --half --model yolov8m --size 1280 --batch 16 --chunk 16
set -e
TAG=${1:-r01_v3}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_mfma
export MI355_SCHED_LOG=1     # the engine prints its launch order (schedule, fused ops) for tools/layer_report.py
# per-kernel durations are taken on ONE in-order stream (same kernels, same plans); the product overlaps independent branches on
# 4 streams, which would smear the per-layer timings.  The PMC passes below run the default multi-stream configuration.
MI355_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs "$@" > gpurun_out/${TAG}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs "$@" > gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs "$@" > gpurun_out/${TAG}_write.log 2>&1
if [ "${MFMA:-0}" = "1" ]; then
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs "$@" > gpurun_out/${TAG}_mfma.log 2>&1
fi
ls gpurun_out/${TAG}_*/*/ | head -20
