#!/bin/bash
# Run ON THE GPU BOX: per-kernel SQ counters of the headline workload (each set in its own rocprofv3 pass, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-sq}; shift || true
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  rm -rf gpurun_out/${TAG}_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/${TAG}_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs "$@" > gpurun_out/${TAG}_$i.log 2>&1
  ls gpurun_out/${TAG}_$i/*/ | head -3
done
