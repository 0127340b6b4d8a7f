"""Debug aid (GPU box): are repeated calls on one shape identical, and do the scheduling / memory knobs change results?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvsd_amd import YOLO
from tools import synth
_, sd = synth.synthetic_checkpoint("yolov8n", seed=0)
frames = torch.from_numpy(synth.synthetic_frames(6, 320, 320, seed=9)).cuda()
m = YOLO.from_state_dict("yolov8n", sd, batch_chunk=4)
res = []
for k in range(4):
    rows, counts, _ = m._infer_rows(frames, 0.25, 0.7, None, 300, 320)
    res.append((counts.copy(), rows.copy()))
    print(os.environ.get("TAGX", ""), "call", k, counts.tolist(), m.plan_info())
out = m.new_device_rows(6)
for k in range(3):
    m.infer_async(frames, out, conf=0.25, iou=0.7, imgsz=320)
    m.sync()
    print("async", k, out[1].cpu().numpy().tolist())
h = m.raw_head(frames[:4].cpu().numpy(), imgsz=320)
h2 = m.raw_head(frames[:4].cpu().numpy(), imgsz=320)
print("raw head repeat equal:", np.array_equal(h, h2), float(np.abs(h - h2).max()))
np.save(os.environ.get("OUTNPY", "/tmp/head.npy"), h)
