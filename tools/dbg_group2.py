import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvsd_amd import ops
rng = np.random.default_rng(0)
for (n, h, w, cin, ca, ka, sa, cb, kb, sb) in [(4, 40, 40, 80, 80, 1, 1, 64, 3, 1), (4, 20, 20, 64, 80, 1, 1, 64, 3, 1)]:
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wa = (rng.standard_normal((ca, cin, ka, ka)) / np.sqrt(cin * ka * ka)).astype(np.float32); ba = rng.standard_normal(ca).astype(np.float32) * 0.1
    wb = (rng.standard_normal((cb, cin, kb, kb)) / np.sqrt(cin * kb * kb)).astype(np.float32); bb = rng.standard_normal(cb).astype(np.float32) * 0.1
    ra = ops.conv2d(x, wa, ba, stride=sa); rb = ops.conv2d(x, wb, bb, stride=sb)
    _, _, na, nb = ops.conv2d_group(x, wa, ba, wb, bb, sa, sb)
    bad = 0
    for rep in range(3):
      for pa in range(na):
        for pb in range(nb):
            ya, yb, _, _ = ops.conv2d_group(x, wa, ba, wb, bb, sa, sb, pa, pb)
            ea, eb = np.array_equal(ya, ra), np.array_equal(yb, rb)
            if not (ea and eb):
                bad += 1
                if bad <= 10:
                    print("  MISMATCH plans", pa, pb, "a ok" if ea else f"a bad {np.abs(ya-ra).max():.3g} {np.mean(ya!=ra):.3f}", "b ok" if eb else f"b bad {np.abs(yb-rb).max():.3g} {np.mean(yb!=rb):.3f}")
    print("case", (n, h, w, cin, ca, ka, sa, cb, kb, sb), "menu", na, nb, "bad pairs", bad)
