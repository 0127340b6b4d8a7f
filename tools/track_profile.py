"""cProfile of the model.track loop (GPU motion compensation) on a synthetic 320x240 clip: where the host time goes."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvsd_amd import YOLO
from cvsd_amd.weights import build_from_state_dict
from tools import synth
n = 200
_, sd = synth.synthetic_checkpoint("yolov8n", seed=0)
model = YOLO(build_from_state_dict("yolov8n", sd), batch_chunk=1)
rng = np.random.default_rng(5)
base = rng.integers(0, 256, size=(256, 320 + 3 * n + 16, 3), dtype=np.uint8)
base = ((base.astype(np.uint16) + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) // 4).astype(np.uint8)
frames = [np.ascontiguousarray(base[8:248, 3 * k:3 * k + 320]) for k in range(n)]
for f in frames[:10]:
    model.track(f, persist=True, conf=0.1)
pr = cProfile.Profile(); pr.enable()
for f in frames[10:]:
    model.track(f, persist=True, conf=0.1)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
