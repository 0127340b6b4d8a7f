#!/bin/bash
# Round 4, GPU call 44: what-if timings of the k3 s2 half stem (diag build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_stemdiag.so
rm -rf gpurun_out/r04_stem_whatif
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04_stem_whatif -- python3 tools/stem_whatif.py > gpurun_out/r04_stem_whatif.log 2>&1; tail -1 gpurun_out/r04_stem_whatif.log
python3 tools/stem_whatif.py gpurun_out/r04_stem_whatif
