#!/bin/bash
# Round 4, GPU call 9: the LDS-weights 3x3 kernel (version 7) -- parity of every plan, plan tables of the shapes that carry config 5, LDS
# bank-conflict counters with the two pixel strides and for the new kernel.
cd "$GRAFT_REPO_ROOT"
set -o pipefail
export MI355_PLAN_CACHE=0 MI355_PLAN_DIR=$PWD/gpurun_out/empty_dir; mkdir -p gpurun_out/empty_dir
timeout -k 10 600 python -m pytest tests/test_gpu_half.py -x -q -m gpu -k "every_plan_against_float64 or bits_do_not_depend" > gpurun_out/r04_c9_tests.log 2>&1 || { tail -40 gpurun_out/r04_c9_tests.log; exit 1; }
tail -3 gpurun_out/r04_c9_tests.log
for SH in "16 80 80 192 192" "16 160 160 96 96" "16 320 320 48 48" "16 40 40 288 288" "16 160 160 192 256" "16 160 160 64 64"; do
  MI355_BENCH_HALF=1 MB_TOP=3 timeout -k 10 200 python tools/conv_microbench.py $SH 3 1 1 0 2>&1 | grep -v amdgpu.ids
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for PAD in 4 8; do
  export MI355_LDS_PAD=$PAD MI355_BENCH_HALF=1
  rm -rf gpurun_out/r04_c9_pmc_$PAD
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d gpurun_out/r04_c9_pmc_$PAD -- python3 tools/conv_microbench.py 16 80 80 192 192 3 1 1 0 12 > gpurun_out/r04_c9_pmc_$PAD.log 2>&1
  echo "== pad $PAD"; python tools/pmc_by_kernel.py gpurun_out/r04_c9_pmc_$PAD | head -12
  rm -rf gpurun_out/r04_c9_pmc_$PAD
done
unset MI355_LDS_PAD
rm -rf gpurun_out/r04_c9_pmc_w
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/r04_c9_pmc_w -- python3 tools/conv_microbench.py 16 80 80 192 192 3 1 1 0 12 > gpurun_out/r04_c9_pmc_w.log 2>&1
echo "== waits"; python tools/pmc_by_kernel.py gpurun_out/r04_c9_pmc_w | head -12
rm -rf gpurun_out/r04_c9_pmc_w
