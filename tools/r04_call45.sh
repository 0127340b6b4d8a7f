#!/bin/bash
# Round 4, GPU call 45: the k3 s2 half stem, second form (aligned dword staging, host-prepared A fragments, permuted couts): tests + what-if timings
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests/test_gpu_half.py -x -q -m gpu -k "stem or config5 or raw_head or reproducible or close_to_fp32" > gpurun_out/r04_c45_tests.log 2>&1; tail -3 gpurun_out/r04_c45_tests.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_stemdiag.so
rm -rf gpurun_out/r04_stem_whatif
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04_stem_whatif -- python3 tools/stem_whatif.py > gpurun_out/r04_stem_whatif.log 2>&1; tail -1 gpurun_out/r04_stem_whatif.log
python3 tools/stem_whatif.py gpurun_out/r04_stem_whatif
