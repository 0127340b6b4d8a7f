#!/bin/bash
# Round 4, GPU call 20: the driver's command on the final build (shipped plans + committed profiles: roofline.traffic is quoted when the plan hash matches)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_style.json 2> gpurun_out/r04_bench_driver_style.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_bench_driver_style.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], r["frac"], r["flops_frac"], r["hbm_frac"], r["traffic"], r["plan_source"], r["plan_hash"], r["launches_per_step"])
print([(c["workload"], c["value"], c["roofline"]["frac"], c["roofline"].get("plan_source")) for c in d["configs"]])
print(d["track_pipeline"])
PY
