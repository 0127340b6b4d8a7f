#!/bin/bash
# Round 4, GPU call 59: A/B of a start offset between the two resident blocks of a CU in conv_igemm_f16 (MI355_F16_STAGGER, 64-cycle units) on the stride-2 and head shapes of config 5
cd "$GRAFT_REPO_ROOT"
export MI355_BENCH_HALF=1 MB_TOP=1 MB_FILTER=v1
for s in 0 64 128 256 0; do
  echo "== MI355_F16_STAGGER=$s"
  MI355_F16_STAGGER=$s timeout -k 10 200 python tools/conv_microbench.py 16 640 640 48 96 3 2 1 0 40 2>&1 | sed -n 3p
  MI355_F16_STAGGER=$s timeout -k 10 200 python tools/conv_microbench.py 16 320 320 96 192 3 2 1 0 40 2>&1 | sed -n 3p
  MI355_F16_STAGGER=$s timeout -k 10 200 python tools/conv_microbench.py 16 160 160 192 256 3 1 1 0 40 2>&1 | sed -n 3p
done
