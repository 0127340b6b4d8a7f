#!/bin/bash
# Run ON THE GPU BOX: time the launch plans of every workload bench.py measures into gpurun_out/plans_new/ (to be committed under
# computer-vision-shoplifting-detection_amd/plans/).  Shipped files are ignored while doing so (MI355_PLAN_DIR points at an empty directory).
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/plans_new gpurun_out/empty_dir
rm -f gpurun_out/plans_new/*.plan
export MI355_PLAN_DIR=$PWD/gpurun_out/empty_dir MI355_PLAN_CACHE=$PWD/gpurun_out/plans_new
timeout -k 10 1100 python bench.py --steps 20 --warmup 5 > gpurun_out/make_plans_bench.json 2> gpurun_out/make_plans_bench.err || { tail -20 gpurun_out/make_plans_bench.err; exit 1; }
# the half=True headline form of config 5 at batch 16 as the rocprof collection runs it (chunk 16) is the same shape as the configs entry
ls gpurun_out/plans_new | wc -l
python - <<'PY'
import json
d = json.loads(open("gpurun_out/make_plans_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"]["plan_source"], [(c["workload"], c["value"], c["roofline"]["frac"]) for c in d["configs"]], d["track_pipeline"])
PY
