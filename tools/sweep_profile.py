"""Where a batch of the clip-sharded sweep (cvsd_amd/sweep.py: process_clip) spends its host time: cProfile of one 1,280-frame synthetic clip
(320x240, panning camera) at batch 64 on YOLOv8n.   python tools/sweep_profile.py [frames=1280]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvsd_amd import YOLO
from cvsd_amd.sweep import process_clip
from cvsd_amd.weights import build_from_state_dict
from tools import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
rng = np.random.default_rng(5)
base = rng.integers(0, 256, size=(256, 320 + 3 * 160 + 16, 3), dtype=np.uint8)
base = ((base.astype(np.uint16) + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) // 4).astype(np.uint8)
frames = [np.ascontiguousarray(base[8:248, 3 * (k % 160):3 * (k % 160) + 320]) for k in range(n)]


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


_, sd = synth.synthetic_checkpoint("yolov8n", seed=0)
model = YOLO(build_from_state_dict("yolov8n", sd), batch_chunk=64)
process_clip(model, Clip(frames[:128]), batch=64)
t0 = time.perf_counter()
rows = process_clip(model, Clip(frames), batch=64)
dt = time.perf_counter() - t0
print(f"{n} frames in {dt * 1e3:.1f} ms -> {n / dt:.0f} frames/s ({dt / n * 1e6:.1f} us per frame), {len(rows)} rows")
pr = cProfile.Profile()
pr.enable()
process_clip(model, Clip(frames), batch=64)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
