#!/bin/bash
# Sample the shader clock / power while a bench workload runs (GPU box): is a config clock- (power-) limited?
#   tools/clock_sample.sh <tag> <bench.py args...>
cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
python bench.py --no-cpu-baseline --no-configs "$@" > gpurun_out/clk_$TAG.json 2>/dev/null &
BP=$!
sleep 25     # model build + autotune
for i in $(seq 1 40); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|junction" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done > gpurun_out/clk_$TAG.txt
wait $BP
tail -c 400 gpurun_out/clk_$TAG.json | python -c "import sys,json; d=json.loads(sys.stdin.read()[sys.stdin.read().find('{'):] or '{}')" 2>/dev/null
python - <<E
import json
l=[x for x in open('gpurun_out/clk_$TAG.json') if x.startswith('{')]
d=json.loads(l[-1]); print('$TAG', d['value'], 'fps', d['roofline']['achieved'], d['roofline']['unit'])
E
sort gpurun_out/clk_$TAG.txt | uniq -c | sort -rn | head -8
