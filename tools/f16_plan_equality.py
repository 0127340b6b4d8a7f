"""Is the half=True conv's result independent of the launch plan?  ('make the f16 accumulation order plan-independent'.)
For a set of conv shapes: run every candidate plan and count the plans whose output bits differ from plan 0's.
    python tools/f16_plan_equality.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvsd_amd import ops

CASES = [  # n, h, w, cin, cout, k, stride, silu, residual
    (2, 40, 40, 192, 192, 3, 1, True, True), (2, 80, 80, 96, 96, 3, 1, True, False), (1, 80, 80, 48, 96, 3, 2, True, False),
    (2, 40, 40, 576, 192, 1, 1, True, False), (2, 40, 40, 384, 384, 1, 1, True, False), (1, 40, 40, 192, 256, 3, 1, True, False),
    (4, 20, 20, 288, 288, 3, 1, True, True), (2, 32, 32, 64, 64, 3, 1, True, False), (1, 24, 24, 1152, 576, 1, 1, True, False),
]
for case in CASES:
    n, h, w, cin, cout, k, stride, silu, residual = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, k, k), dtype=np.float32) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    res = rng.standard_normal((n, h // stride, w // stride, cout), dtype=np.float32) if residual else None
    y0, n_plans = ops.conv2d(x, wt, b, stride=stride, silu=silu, residual=res, half=True, return_n_plans=True)
    differ, worst = 0, 0.0
    for plan in range(1, n_plans):
        y = ops.conv2d(x, wt, b, stride=stride, silu=silu, residual=res, half=True, plan=plan)
        if not np.array_equal(y, y0):
            differ += 1
            worst = max(worst, float(np.abs(y - y0).max()))
    again = ops.conv2d(x, wt, b, stride=stride, silu=silu, residual=res, half=True, plan=0)
    print(f"{case}: {n_plans} plans, {differ} differ from plan 0 (max |delta| {worst:.3e}); plan 0 repeated: {'same bits' if np.array_equal(again, y0) else 'DIFFERENT'}",
          flush=True)
