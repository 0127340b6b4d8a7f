"""Lucas-Kanade set-up through LDS planes (product) against per-sample global loads (tools/ab_build.sh "-DMI355_LK_GLOBAL_SETUP=1" lkold):
same bits, time per call.  python tools/lk_setup_ab.py  (run once per library via MI355_YOLO_LIB; prints a checksum to compare)"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvsd_amd import gmc
rng = np.random.default_rng(3)
h, w = 120, 160
base = rng.integers(0, 256, size=(h + 16, w + 16), dtype=np.uint8)
base = ((base.astype(np.uint16) + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) // 4).astype(np.uint8)
prev, cur = np.ascontiguousarray(base[8:8 + h, 8:8 + w]), np.ascontiguousarray(base[7:7 + h, 5:5 + w])
pts = gmc.good_features_to_track(prev)
edge = np.array([[0.5, 0.5], [w - 1.0, 1.0], [2.0, h - 1.0], [w - 1.5, h - 1.5]], np.float32)
pts = np.concatenate([pts, edge])
nxt, st = gmc.calc_optical_flow_pyr_lk(prev, cur, pts, device=0)
t0 = time.perf_counter()
for _ in range(50):
    gmc.calc_optical_flow_pyr_lk(prev, cur, pts, device=0)
dt = (time.perf_counter() - t0) / 50
print(os.environ.get("MI355_YOLO_LIB", "product")[-24:], len(pts), "points", f"{dt * 1e3:.3f} ms per call", "sha", hashlib.sha1(nxt.tobytes() + st.tobytes()).hexdigest()[:16])
