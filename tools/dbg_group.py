"""Debug aid (GPU box): grouped launch of two convs vs the two convs alone, over the menu's plans."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvsd_amd import ops
rng = np.random.default_rng(0)
cases = [(1, 40, 40, 64, 64, 3, 1, 80, 3, 1), (1, 40, 40, 64, 144, 3, 1, 64, 3, 2), (1, 20, 20, 128, 64, 1, 1, 128, 3, 1), (4, 40, 40, 64, 64, 3, 1, 64, 3, 1)]
for (n, h, w, cin, ca, ka, sa, cb, kb, sb) in cases:
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wa = (rng.standard_normal((ca, cin, ka, ka)) / np.sqrt(cin * ka * ka)).astype(np.float32); ba = rng.standard_normal(ca).astype(np.float32) * 0.1
    wb = (rng.standard_normal((cb, cin, kb, kb)) / np.sqrt(cin * kb * kb)).astype(np.float32); bb = rng.standard_normal(cb).astype(np.float32) * 0.1
    ra = ops.conv2d(x, wa, ba, stride=sa); rb = ops.conv2d(x, wb, bb, stride=sb)
    _, _, na, nb = ops.conv2d_group(x, wa, ba, wb, bb, sa, sb)
    bad = 0
    for pa in range(na):
        for pb in range(0, nb, max(1, nb // 6)):
            ya, yb, _, _ = ops.conv2d_group(x, wa, ba, wb, bb, sa, sb, pa, pb)
            ea, eb = np.array_equal(ya, ra), np.array_equal(yb, rb)
            if not (ea and eb):
                bad += 1
                if bad <= 6:
                    print("  MISMATCH case", (n, h, w, cin, ca, ka, sa, cb, kb, sb), "plans", pa, pb, "a ok" if ea else f"a bad {np.abs(ya-ra).max():.3g} {np.mean(ya!=ra):.3f}", "b ok" if eb else f"b bad {np.abs(yb-rb).max():.3g} {np.mean(yb!=rb):.3f}")
    print("case", (n, h, w, cin, ca, ka, sa, cb, kb, sb), "menu", na, nb, "bad pairs", bad)
print("---- fused pointwise member")
for (n, h, w, cin, ca, sa, c2, cb, kb, sb) in [(1, 40, 40, 64, 64, 1, 64, 80, 3, 1), (4, 40, 40, 64, 64, 1, 64, 80, 3, 1), (1, 80, 80, 64, 80, 1, 80, 64, 3, 1), (2, 40, 40, 32, 64, 2, 64, 32, 1, 1)]:
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wa = (rng.standard_normal((ca, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32); ba = rng.standard_normal(ca).astype(np.float32) * 0.1
    w2 = (rng.standard_normal((c2, ca, 1, 1)) / np.sqrt(ca)).astype(np.float32); b2 = rng.standard_normal(c2).astype(np.float32) * 0.1
    wb = (rng.standard_normal((cb, cin, kb, kb)) / np.sqrt(cin * kb * kb)).astype(np.float32); bb = rng.standard_normal(cb).astype(np.float32) * 0.1
    ra = ops.conv2d_fused(x, wa, ba, w2, b2, stride=sa); rb = ops.conv2d(x, wb, bb, stride=sb)
    _, _, na, nb = ops.conv2d_group(x, wa, ba, wb, bb, sa, sb, w2a=w2, b2a=b2)
    bad = 0
    for pa in range(na):
        for pb in range(0, nb, max(1, nb // 4)):
            ya, yb, _, _ = ops.conv2d_group(x, wa, ba, wb, bb, sa, sb, pa, pb, w2a=w2, b2a=b2)
            ea, eb = np.array_equal(ya, ra), np.array_equal(yb, rb)
            if not (ea and eb):
                bad += 1
                if bad <= 6:
                    print("  MISMATCH", (n, h, w, cin, ca, sa, c2, cb, kb, sb), "plans", pa, pb, "a ok" if ea else f"a bad {np.abs(ya-ra).max():.3g} {np.mean(ya!=ra):.3f}", "b ok" if eb else f"b bad {np.abs(yb-rb).max():.3g} {np.mean(yb!=rb):.3f}")
    print("case", (n, h, w, cin, ca, sa, c2, cb, kb, sb), "menu", na, nb, "bad pairs", bad)
