cd "$GRAFT_REPO_ROOT"
export MI355_YOLO_LIB=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_exp.so MI355_BENCH_HALF=1 MB_TOP=60
for ex in 0 1 4 8 16 24 25 29; do
  echo "#### EXP=$ex"
  MI355_F16_EXP=$ex python tools/conv_microbench.py 16 80 80 192 192 3 1 1 0 2>&1 | head -45
done
