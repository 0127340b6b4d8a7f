"""Turn the raw rocprofv3 output of tools/collect_profiles.sh (under gpurun_out/) into the committed profiles/ files.

usage: summarize_profiles.py TAG BATCH [MODEL [SIZE [half]]]   (defaults: r01_v3 256 yolov8n 640)"""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_v3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
model = sys.argv[3] if len(sys.argv) > 3 else "yolov8n"
size = int(sys.argv[4]) if len(sys.argv) > 4 else 640
half = len(sys.argv) > 5 and sys.argv[5] == "half"
es = 2 if half else 4
g = lambda pat: max(glob.glob(os.path.join(ROOT, "gpurun_out", pat)), key=os.path.getmtime)   # newest run if stale ones linger
shutil.copy(g(f"{tag}_trace/*/*kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
with open(os.path.join(ROOT, "profiles", f"{tag}_bench_under_rocprof.json"), "w") as f:
    f.write([l for l in open(os.path.join(ROOT, "gpurun_out", f"{tag}_trace.log")) if l.startswith('{"metric')][-1])
rep = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "layer_report.py"), g(f"{tag}_trace/*/*kernel_trace.csv"), model, str(batch),
                      str(size), str(es), os.path.join(ROOT, "gpurun_out", f"{tag}_trace.log")], capture_output=True, text=True)
open(os.path.join(ROOT, "profiles", f"{tag}_layer_report.txt"), "w").write(rep.stdout)
print(rep.stdout[-600:], rep.stderr[-2000:])

sys.path.insert(0, ROOT)
from cvsd_amd.graph import build_program, parse_model_name
pg = build_program(*parse_model_name(model))
# conv launches of one step: fewer than the module's convs (sibling convs merged, Conv3x3 -> Conv1x1 pairs fused where the
# autotuner found that faster); bench.py reports the count it measured in the same process
_fetch_line = json.loads([l for l in open(os.path.join(ROOT, "gpurun_out", f"{tag}_fetch.log")) if l.startswith('{"metric')][-1])
n_conv = _fetch_line["roofline"]["launches_per_step"]
plan_hashes = {k: json.loads([l for l in open(os.path.join(ROOT, "gpurun_out", f"{tag}_{k}.log")) if l.startswith('{"metric')][-1])["roofline"].get("plan_hash")
               for k in ("trace", "fetch", "write") if os.path.exists(os.path.join(ROOT, "gpurun_out", f"{tag}_{k}.log"))}
is_conv = lambda name: "conv_igemm" in name or "conv1x1_" in name or "conv_splitk" in name or "conv_group" in name or "conv3x3_lw" in name


def conv_sum(pat, counter, passes=6):
    rows = sorted((int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(g(pat)))
                  if r["Counter_Name"] == counter and is_conv(r["Kernel_Name"]))
    return sum(v for _, v in rows[-passes * n_conv:]) / passes


f = conv_sum(f"{tag}_fetch/*/*counter_collection.csv", "FETCH_SIZE")
w = conv_sum(f"{tag}_write/*/*counter_collection.csv", "WRITE_SIZE")
sc = (size / 640.0) ** 2
alg_in = sum(es * c.cin * (640 // c.stride_div * c.s) ** 2 for c in pg.convs if c.cin != 3) * batch * sc
alg_out = sum(es * c.cout * (640 // c.stride_div) ** 2 for c in pg.convs if c.cin != 3) * batch * sc
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1`; conv dispatches "
                 "of the 6 real passes only (the engine's one-off autotune launches are excluded)",
       "workload": f"{model} {size}x{size} batch {batch}" + (" half=True" if half else ""),
       "units": "KiB; FETCH doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B) -- uncalibrated for 64-B-segment reads",
       "plan_hash": _fetch_line["roofline"].get("plan_hash"), "plan_source": _fetch_line["roofline"].get("plan_source"), "plan_hash_of_every_pass": plan_hashes,
       "frames_per_step": batch, "conv_launches_per_step": n_conv, "launches_per_step": n_conv, "fetch_kib_per_step_raw": f, "write_kib_per_step": w,
       "hbm_bytes_per_step_corrected": f * 2048 + w * 1024, "hbm_bytes_per_launch_avg": (f * 2048 + w * 1024) / n_conv,
       "algorithmic_input_bytes_per_step": alg_in, "algorithmic_output_bytes_per_step": alg_out}
try:
    pat = f"{tag}_mfma/*/*counter_collection.csv"
    busy = conv_sum(pat, "SQ_VALU_MFMA_BUSY_CYCLES"); sq = conv_sum(pat, "SQ_BUSY_CYCLES"); gui = conv_sum(pat, "GRBM_GUI_ACTIVE")
    out["mfma_counters_per_step"] = {"SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CYCLES": sq, "GRBM_GUI_ACTIVE": gui,
                                     "note": "sums over the conv dispatches of one step; GRBM_GUI_ACTIVE is the sum over the 8 XCDs"}
except (IndexError, FileNotFoundError, ValueError):
    pass
name = f"{tag}_conv_traffic.json"
json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
print(json.dumps(out, indent=1))
