import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvsd_amd import YOLO
from tools import synth
_, sd = synth.synthetic_checkpoint("yolov8n", seed=0)
frames = synth.synthetic_frames(4, 320, 320, seed=9)
os.environ["MI355_GROUPS"] = "0"
m0 = YOLO.from_state_dict("yolov8n", sd, batch_chunk=4)
ref = m0.raw_head(frames, imgsz=320)
os.environ["MI355_GROUPS"] = "1"; os.environ["MI355_GROUP_ONLY"] = "2"
m1 = YOLO.from_state_dict("yolov8n", sd, batch_chunk=4)
for k in range(3):
    h = m1.raw_head(frames, imgsz=320)
    d = h != ref                                   # [n, 84, A]; A = 1600 + 400 + 100
    for name, sl in (("lvl0", slice(0, 1600)), ("lvl1", slice(1600, 2000)), ("lvl2", slice(2000, 2100))):
        print(k, name, "box bad", int(d[:, :4, sl].sum()), "cls bad", int(d[:, 4:, sl].sum()), "of", d[:, 4:, sl].size,
              "| frames with bad cls:", d[:, 4:, sl].any(axis=(1, 2)).tolist(), "| bad anchors", int(d[:, 4:, sl].any(axis=1).sum()))
    bad = np.argwhere(d[:, 4:, :1600].any(axis=1))
    if len(bad):
        print("   first bad (frame, anchor):", bad[:8].tolist(), " last:", bad[-4:].tolist())
        a = bad[0]
        print("   classes bad at first:", np.nonzero(d[a[0], 4:, a[1]])[0].tolist()[:20])
