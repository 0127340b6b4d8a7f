"""What-if timings of the k3 s2 half stem (misc_kernels.hip:stem3s2_u8_h): the kernel with parts switched off, on config 5's frame shape.
Needs a library built with -DMI355_STEM_DIAG=1 (tools/ab_build.sh "-DMI355_STEM_DIAG=1" stemdiag; MI355_YOLO_LIB=...).
  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/stem_whatif.py      then      python3 tools/stem_whatif.py OUT
The first form launches, per what-if mask, REPS stems on 16 frames of 1280 x 1280; the second reads the trace and prints the medians."""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
MASKS = [0, 1, 2, 4, 16, 1 | 4, 2 | 16, 1 | 2 | 4 | 16]
NAMES = {0: "whole kernel", 1: "no SiLU", 2: "no input loads", 4: "no stores", 16: "no weight loads", 5: "no SiLU, no stores",
         18: "no input or weight loads", 23: "MFMAs and staging arithmetic only"}
REPS = 4
if len(sys.argv) > 1:
    import csv
    f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted((r for r in csv.DictReader(open(f)) if "stem" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    names = [r["Kernel_Name"].split("(")[0].replace("void mi355::", "") for r in rows]
    assert len(us) == (len(MASKS) + 1) * REPS, len(us)
    print(f"{names[0]:28s} general half stem            {np.median(us[:REPS]):8.1f} us")
    for i, m in enumerate(MASKS):
        print(f"{names[(i + 1) * REPS]:28s} {NAMES[m]:36s} {np.median(us[(i + 1) * REPS:(i + 2) * REPS]):8.1f} us")
    sys.exit(0)
from cvsd_amd import ops
rng = np.random.default_rng(0)
img = rng.integers(0, 256, size=(16, 1280, 1280, 3), dtype=np.uint8)
w = (rng.standard_normal((48, 3, 3, 3)) * 0.3).astype(np.float32)
b = (rng.standard_normal(48) * 0.1).astype(np.float32)
for _ in range(REPS):
    ops.stem(img, w, b, half=True, variant=1)
for m in MASKS:
    for _ in range(REPS):
        ops.stem(img, w, b, half=True, variant=2 | (m << 8))
print("done")
