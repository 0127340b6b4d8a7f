"""End-to-end rate of the reference's frame loop (model.track -> Tracker rows; /root/reference/model.py:38-64) on synthetic
320x240 clips with a panning camera: detector on the GPU, BoT-SORT with its global motion compensation on the host or on the GPU.
    python tools/track_pipeline_bench.py [frames=150]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvsd_amd import YOLO
from cvsd_amd.tracker import BYTETracker
from cvsd_amd.weights import build_from_state_dict
from tools import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
_, sd = synth.synthetic_checkpoint("yolov8n", seed=0)
model = YOLO(build_from_state_dict("yolov8n", sd), batch_chunk=1)
rng = np.random.default_rng(5)
base = rng.integers(0, 256, size=(240 + 16, 320 + 3 * n + 16, 3), dtype=np.uint8)
# blur a little so that corners exist at several scales
base = ((base.astype(np.uint16) + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) // 4).astype(np.uint8)
frames = [np.ascontiguousarray(base[8:248, 3 * k:3 * k + 320]) for k in range(n)]
for label, dev in (("host GMC", None), ("GPU GMC", 0), ("no GMC", "off")):
    model._tracker = BYTETracker(gmc_method=None) if dev == "off" else BYTETracker(gmc_device=dev)
    for f in frames[:5]:
        model.track(f, persist=True, classes=None, conf=0.1)
    t0 = time.perf_counter()
    for f in frames[5:]:
        model.track(f, persist=True, classes=None, conf=0.1)
    dt = time.perf_counter() - t0
    print(f"{label:9s}: {(n - 5) / dt:8.1f} frames/s  ({dt / (n - 5) * 1e3:.2f} ms per frame: predict + tracker)")

# the production form (cvsd_amd/sweep.py): batched detection, tracker frame by frame with the next frame's step enqueued ahead
from cvsd_amd.sweep import process_clip


class Clip:
    def __init__(self, fr):
        self.fr, self.pos = fr, 0

    def read(self):
        if self.pos >= len(self.fr):
            return False, None
        self.pos += 1
        return True, self.fr[self.pos - 1]

    def get(self, prop):
        return float(self.pos)

    def release(self):
        pass


big = YOLO(build_from_state_dict("yolov8n", sd), batch_chunk=64)
process_clip(big, Clip(frames[:64]), batch=64)              # plans of the 64-frame pass
long_clip = frames * 4                                      # the clip four times over: several batches in flight
t0 = time.perf_counter()
process_clip(big, Clip(long_clip), batch=64)
print(f"sweep batch 64, GPU GMC: {len(long_clip) / (time.perf_counter() - t0):8.1f} frames/s")
