"""Debug: YOLOv8m 1280 half, batch 2 against frame-by-frame (rows must be identical)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvsd_amd import YOLO
from cvsd_amd.weights import build_from_state_dict
from tools import synth
name = sys.argv[1] if len(sys.argv) > 1 else "yolov8m"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ckpt = synth.synthetic_checkpoint(name, seed=0)
m = YOLO(build_from_state_dict(name, ckpt[1]), batch_chunk=nb, half=True)
frames = synth.synthetic_frames(nb, size, size, seed=3)
both = m.predict(frames, conf=0.25, imgsz=size)
raw_b = m.raw_head(frames, imgsz=size)
for i in range(nb):
    one = m.predict(frames[i], conf=0.25, imgsz=size)[0]
    raw_1 = m.raw_head(frames[i:i + 1], imgsz=size)
    d = np.abs(raw_b[i] - raw_1[0])
    bad = np.argwhere(d.max(0) > 0).ravel()
    print(f"frame {i}: rows batch {len(both[i].anchor_idx)} single {len(one.anchor_idx)}; head anchors differing {len(bad)} of {d.shape[1]}; max |d| {d.max():.4g}")
    if len(bad):
        print("   first differing anchors:", bad[:24], " mod 16:", sorted(set((bad % 16).tolist())))
        lv0 = (size // 8) ** 2
        print("   per level:", (bad < lv0).sum(), ((bad >= lv0) & (bad < lv0 * 5 // 4)).sum(), (bad >= lv0 * 5 // 4).sum(), " channels differing:", np.argwhere(d.max(1) > 0).ravel()[:20])
    if len(bad):
        a0 = bad[0]
        np.set_printoptions(precision=4, linewidth=200, suppress=True)
        print("   anchor", a0, "batch :", raw_b[i][4:24, a0])
        print("   anchor", a0, "single:", raw_1[0][4:24, a0])
        print("   anchor", a0 - 1, "batch :", raw_b[i][4:12, a0 - 1], "single:", raw_1[0][4:12, a0 - 1])
