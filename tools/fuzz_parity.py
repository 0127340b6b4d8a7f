"""One-off randomized parity sweep (run on the GPU box): engine rows vs the canonical-order C oracle, bit for bit, over
random models / frame sizes / imgsz / batch sizes / thresholds; half=True engine against its own contract's noise bound is
covered by tests/test_gpu_half.py and not repeated here.   python tools/fuzz_parity.py [n_cases] [seed]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvsd_amd import YOLO
from oracle import det
from tools import synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
models = ["yolov8n", "yolov8n-pose", "yolov8s", "yolov5nu", "yolov8s-pose", "yolov5su"]
cache = {}
bad = 0
for case in range(n_cases):
    name = models[int(rng.integers(len(models)))]
    if name not in cache:
        ck = synth.synthetic_checkpoint(name, seed=0)
        cache[name] = (YOLO.from_state_dict(name, ck[1], batch_chunk=int(rng.integers(1, 6))), det.DetOracleModel(name, ck[1]))
    m, dm = cache[name]
    imgsz = int(rng.choice([160, 256, 320, 416, 640]))
    h, w = int(rng.integers(48, 400)), int(rng.integers(48, 500))
    n = int(rng.integers(1, 8))
    conf = float(rng.choice([0.001, 0.1, 0.25, 0.5]))
    classes = None if rng.random() < 0.6 or dm.pose else [int(c) for c in rng.choice(80, 5, replace=False)]
    max_det = int(rng.choice([5, 50, 300]))
    frames = synth.synthetic_frames(n, h, w, seed=int(rng.integers(1 << 30)))
    t0 = time.time()
    want, _ = det.predict(dm, list(frames), conf=conf, classes=classes, max_det=max_det, imgsz=imgsz)
    got = m.predict(frames, conf=conf, classes=classes, max_det=max_det, imgsz=imgsz)
    ok = True
    for g, wv in zip(got, want):
        ok &= np.array_equal(g.anchor_idx, wv["anchor_idx"].numpy()) and np.array_equal(g.boxes.data.numpy(), wv["boxes"].numpy())
        if dm.pose and len(g.anchor_idx):
            ok &= np.array_equal(g.keypoints_raw, wv["kpts"].numpy())
    bad += not ok
    print(f"case {case:2d} {name:13s} n={n} {h}x{w} imgsz={imgsz} conf={conf} max_det={max_det} classes={'some' if classes else 'all'} "
          f"rows={sum(len(g) for g in got):4d}  {'OK' if ok else 'MISMATCH'}  ({time.time() - t0:.1f}s)", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
