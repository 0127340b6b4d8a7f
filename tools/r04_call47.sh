#!/bin/bash
# Round 4, GPU call 47: what-ifs of the fp32 headline's first layers (diag build: MI355_F32_EXP 1 = halo tiles read the zero page, 4 = stores dropped)
cd "$GRAFT_REPO_ROOT"
export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_f32diag.so
for e in 0 1 4 5; do
  MI355_F32_EXP=$e timeout -k 10 200 tools/trace_layers.sh r04_f32exp$e yolov8n 512 --steps 4 --warmup 2 > gpurun_out/r04_f32exp$e.log 2>&1
  echo "== MI355_F32_EXP=$e"; grep -E "^model\.(0|1|2\.m\.0\.cv1|2\.m\.0\.cv2|3|4\.m\.0\.cv1|4\.m\.1\.cv2|5|7|15\.m\.0\.cv2|22\.cv2\.0\.0.*|22\.cv3\.0\.1) " gpurun_out/r04_f32exp${e}_layer_report.txt | cut -c1-42,100-190; tail -1 gpurun_out/r04_f32exp${e}_layer_report.txt | cut -c1-200
  rm -f gpurun_out/r04_f32exp${e}_trace/*/*_kernel_trace.csv
done
