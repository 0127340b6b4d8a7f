"""Where a frame of the reference's loop (model.track, /root/reference/model.py:38) spends its time on the host side: the stages of
YOLO.track timed one by one on the synthetic panning clip of bench.py's track_pipeline.
    python tools/track_stages.py [model=yolov8n] [frames=200]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cvsd_amd import YOLO
from cvsd_amd.tracker import BYTETracker
from cvsd_amd.weights import build_from_state_dict
from tools import synth

name = sys.argv[1] if len(sys.argv) > 1 else "yolov8n"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
_, sd = synth.synthetic_checkpoint(name, seed=0)
model = YOLO(build_from_state_dict(name, sd), batch_chunk=1)
rng = np.random.default_rng(5)
base = rng.integers(0, 256, size=(256, 320 + 3 * n + 16, 3), dtype=np.uint8)
base = ((base.astype(np.uint16) + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) // 4).astype(np.uint8)
frames = [np.ascontiguousarray(base[8:248, 3 * k:3 * k + 320]) for k in range(n)]
tr = BYTETracker(gmc_device=model.device)
T = dict(begin=0.0, predict=0.0, gmc_collect=0.0, core=0.0, rewrite=0.0)
rows_seen = dets = 0
for k, f in enumerate(frames):
    t0 = time.perf_counter()
    tr.gmc.begin(f)
    t1 = time.perf_counter()
    dev = tr.gmc.pending_device_frame()          # as YOLO.track: the detector reads the copy of the frame the step has just uploaded
    res = (model._predict_batch(YOLO._DeviceFrames(dev[0], 1, dev[1], dev[2]), None, 0.1, 0.7, None, 300, 640, None)[0] if dev is not None
           else model.predict(f[None], conf=0.1)[0])
    t3 = time.perf_counter()
    warp = tr.gmc.apply(f)
    t4 = time.perf_counter()
    tracks = tr.update(res.boxes.data.numpy(), warp=warp)
    t5 = time.perf_counter()
    if len(tracks):
        r2 = res[tracks[:, -1].astype(int)]
        r2.update(boxes=torch.as_tensor(tracks[:, :-1], dtype=torch.float32))
    t6 = time.perf_counter()
    if k >= 10:
        T["begin"] += t1 - t0; T["predict"] += t3 - t1; T["gmc_collect"] += t4 - t3; T["core"] += t5 - t4; T["rewrite"] += t6 - t5
        rows_seen += len(tracks); dets += len(res)
m = n - 10
print(f"{name}: per frame, microseconds over {m} frames ({rows_seen / m:.1f} tracks, {dets / m:.1f} detections per frame):")
for k, v in T.items():
    print(f"  {k:12s} {v / m * 1e6:8.1f}")
print(f"  {'sum':12s} {sum(T.values()) / m * 1e6:8.1f}")
# the detector alone, no motion compensation beside it: engine call only, then with the Results objects
t0 = time.perf_counter()
for f in frames[10:]:
    model._infer_rows(f[None], 0.1, 0.7, None, 300, 640)
t1 = time.perf_counter()
for f in frames[10:]:
    model.predict(f[None], conf=0.1)
t2 = time.perf_counter()
print(f"  detector alone: engine call {(t1 - t0) / m * 1e6:.1f}, predict() {(t2 - t1) / m * 1e6:.1f}")
info = model.plan_info()
print(f"  plans: {info['plan_source']}, {info['launches_per_pass']} launches per pass")
