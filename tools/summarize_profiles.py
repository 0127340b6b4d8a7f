"""Turn the raw rocprofv3 output of tools/collect_profiles.sh (under gpurun_out/) into the committed profiles/ files."""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_v3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = lambda pat: glob.glob(os.path.join(ROOT, "gpurun_out", pat))[0]
shutil.copy(g(f"{tag}_trace/*/*kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
with open(os.path.join(ROOT, "profiles", f"{tag}_bench_under_rocprof.json"), "w") as f:
    f.write([l for l in open(os.path.join(ROOT, "gpurun_out", f"{tag}_trace.log")) if l.startswith('{"metric')][-1])
rep = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "layer_report.py"), g(f"{tag}_trace/*/*kernel_trace.csv"), "yolov8n", str(batch)],
                     capture_output=True, text=True).stdout
open(os.path.join(ROOT, "profiles", f"{tag}_layer_report.txt"), "w").write(rep)
print(rep[-400:])

def conv_sum(pat, counter, passes=6, per_pass=62):
    rows = sorted((int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(g(pat)))
                  if r["Counter_Name"] == counter and ("conv_igemm" in r["Kernel_Name"] or "conv1x1_stream" in r["Kernel_Name"]))
    return sum(v for _, v in rows[-passes * per_pass:]) / passes
f = conv_sum(f"{tag}_fetch/*/*counter_collection.csv", "FETCH_SIZE")
w = conv_sum(f"{tag}_write/*/*counter_collection.csv", "WRITE_SIZE")
sys.path.insert(0, ROOT)
from cvsd_amd.graph import build_program
pg = build_program("v8", "n", "detect")
alg_in = sum(4 * c.cin * (640 // c.stride_div * c.s) ** 2 for c in pg.convs if c.cin != 3) * batch
alg_out = sum(4 * c.cout * (640 // c.stride_div) ** 2 for c in pg.convs if c.cin != 3) * batch
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1`; conv dispatches "
                 "of the 6 real passes only (the engine's one-off autotune launches are excluded)",
       "units": "KiB; FETCH doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B) -- uncalibrated for 64-B-segment reads",
       "frames_per_step": batch, "conv_launches_per_step": 62, "fetch_kib_per_step_raw": f, "write_kib_per_step": w,
       "hbm_bytes_per_step_corrected": f * 2048 + w * 1024, "hbm_bytes_per_launch_avg": (f * 2048 + w * 1024) / 62,
       "algorithmic_input_bytes_per_step": alg_in, "algorithmic_output_bytes_per_step": alg_out}
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_conv_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
