cd "$GRAFT_REPO_ROOT"
EXP=$PWD/computer-vision-shoplifting-detection_amd/libmi355yolo_exp.so
for shape in "128 160 160 16 16 3 1" "128 160 160 32 32 1 1" "128 80 80 64 64 1 1" "128 80 80 32 32 3 1" "128 40 40 64 64 3 1"; do
  echo "--- A (product) $shape"; python tools/conv_microbench.py $shape 1 0 24 2>/dev/null | sed -n 2,3p
  echo "--- B (exp)     $shape"; MI355_YOLO_LIB=$EXP python tools/conv_microbench.py $shape 1 0 24 2>/dev/null | sed -n 2,3p
done
echo "=== bench A"; python bench.py --no-cpu-baseline --no-configs --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-120
echo "=== bench B"; MI355_PLAN_CACHE=0 MI355_YOLO_LIB=$EXP python bench.py --no-cpu-baseline --no-configs --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-120
