#!/bin/bash
# Kernel A/B experiments: build a SECOND library (same sources + extra -D flags) next to the product one.
#   tools/ab_build.sh "-DMI355_V1_MINWAVES=4" [name=exp]    -> computer-vision-shoplifting-detection_amd/libmi355yolo_<name>.so
# then on the GPU box:  MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_exp.so python tools/conv_microbench.py ...
set -e
cd "$(dirname "$0")/../computer-vision-shoplifting-detection_amd/csrc"
OUT=${2:-exp}
OBJ=/tmp/ab_obj_$OUT
mkdir -p $OBJ
SRCS="conv_f32_k3s1 conv_f32_k3s2 conv_f32_k1 conv_f32_pipe conv_f32_splitk conv_f32_fused_s1 conv_f32_fused_s2 conv_f32_group conv_igemm_f16 conv_f16_fused conv_f16_small conv_f16_lw conv_plan misc_kernels post_kernels engine_load engine_memory engine_plans engine_run engine_abi engine_ops gmc_kernels"
pids=""
for s in $SRCS; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 $1 -c $s.hip -o $OBJ/$s.o & pids="$pids $!"
done
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -c gmc_host.cpp -o $OBJ/gmc_host.o & pids="$pids $!"
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -c tracker_host.cpp -o $OBJ/tracker_host.o & pids="$pids $!"
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmi355yolo_$OUT.so $(for s in $SRCS gmc_host tracker_host; do echo $OBJ/$s.o; done)
ls -la ../libmi355yolo_$OUT.so
