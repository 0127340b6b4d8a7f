"""Debug aid (GPU box): repeat predict / raw_head on one shape and count passes that differ from the ungrouped engine."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvsd_amd import YOLO
from tools import synth
name, n, size, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
_, sd = synth.synthetic_checkpoint(name, seed=0)
frames = synth.synthetic_frames(n, size, size, seed=9)
g = os.environ.get("MI355_GROUPS", "1")
os.environ["MI355_GROUPS"] = "0"
ref = YOLO.from_state_dict(name, sd, batch_chunk=n).raw_head(frames, imgsz=size)
os.environ["MI355_GROUPS"] = g
m = YOLO.from_state_dict(name, sd, batch_chunk=n)
bad = 0
for k in range(reps):
    h = m.raw_head(frames, imgsz=size)
    if not np.array_equal(h, ref):
        bad += 1
        d = h != ref
        if bad <= 3:
            print("   pass", k, "bad elems", int(d.sum()), "channels", np.unique(np.nonzero(d)[1])[:12].tolist(), "anchors", np.unique(np.nonzero(d)[2])[:8].tolist())
print(name, n, size, "menu", os.environ.get("MI355_GROUP_MENU", "15"), "groups", g, "launches", m.plan_info()["launches_per_pass"], "bad passes", bad, "/", reps)
