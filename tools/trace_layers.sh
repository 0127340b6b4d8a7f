#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace of bench.py on ONE in-order stream + the per-layer report.
# usage: trace_layers.sh TAG MODEL BATCH [extra bench flags]
set -e
TAG=$1; MODEL=$2; BATCH=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/${TAG}_trace
export MI355_SCHED_LOG=1
MI355_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python3 bench.py --model $MODEL --batch $BATCH --chunk $BATCH --no-cpu-baseline --no-configs "$@" > gpurun_out/${TAG}_trace.log 2>&1
F=$(ls gpurun_out/${TAG}_trace/*/*_kernel_trace.csv | head -1)
python tools/layer_report.py $F $MODEL $BATCH ${SIZE:-640} ${ES:-4} gpurun_out/${TAG}_trace.log > gpurun_out/${TAG}_layer_report.txt 2>&1
tail -n 1 gpurun_out/${TAG}_trace.log | cut -c1-300
