import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvsd_amd import YOLO
from tools import synth
name, n, size = "yolov8n", 1, 640
_, sd = synth.synthetic_checkpoint(name, seed=0)
frames = synth.synthetic_frames(n, size, size, seed=9)
os.environ["MI355_GROUPS"] = "0"
ref = YOLO.from_state_dict(name, sd, batch_chunk=n).raw_head(frames, imgsz=size)
os.environ["MI355_GROUPS"] = "1"
m = YOLO.from_state_dict(name, sd, batch_chunk=n)
h = m.raw_head(frames, imgsz=size)
d = np.argwhere(h != ref)
print("bad", len(d))
logit = lambda s: np.log(s / (1 - s))
for (b, c, a) in d[:24]:
    print(int(c), int(a), "got", h[b, c, a], "ref", ref[b, c, a], "logit got", logit(float(h[b, c, a])), "ref", logit(float(ref[b, c, a])), "neighbours ref", ref[b, c, a - 1], ref[b, c - 1, a])
