"""Randomized sweep of single convs (fp32 and half=True, with / without activation, residual, fp32 output): every launch plan, twice,
against the canonical-order oracle (fp32: bit for bit) / a float64 reference (half).  Catches plan-dependent corruption such as the
store-data hazard of round 3.   python tools/fuzz_conv_plans.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvsd_amd import ops
from oracle import det
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    k = int(rng.choice([1, 1, 3]))
    stride = int(rng.choice([1, 2])) if k == 3 else 1
    n = int(rng.choice([1, 1, 2, 4]))
    h = w = int(rng.choice([20, 40, 80, 160])) if k == 1 else int(rng.choice([20, 40, 80]))
    cin = int(rng.choice([16, 32, 64, 128, 192, 256]))
    cout = int(rng.choice([16, 51, 64, 80, 96, 128]))
    half = bool(rng.integers(2))
    silu = bool(rng.integers(2))
    res = bool(rng.integers(2)) and stride == 1
    out_f32 = half and not silu and not res and bool(rng.integers(2))
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    r = rng.standard_normal((n, h // stride, w // stride, cout), dtype=np.float32) if res else None
    if half:
        x16, w16 = x.astype(np.float16).astype(np.float32), wt.astype(np.float16).astype(np.float32)
        r16 = None if r is None else r.astype(np.float16).astype(np.float32)
        ref = det.conv2d(x16, w16, b, stride=stride, act=silu, residual=r16).astype(np.float64)
        tol = lambda y: 3e-3 * np.abs(ref) + 3e-3
    else:
        ref = det.conv2d(x, wt, b, stride=stride, act=silu, residual=r)
    _, npl = ops.conv2d(x, wt, b, stride=stride, silu=silu, residual=r, half=half, out_f32=out_f32, return_n_plans=True)
    nb = 0
    for rep in range(2):
        for plan in range(npl):
            y = ops.conv2d(x, wt, b, stride=stride, silu=silu, residual=r, half=half, out_f32=out_f32, plan=plan)
            ok = (np.abs(y - ref) <= tol(y)).all() if half else np.array_equal(y, ref)
            if not ok:
                nb += 1
                if nb <= 2:
                    d = np.argwhere(np.abs(y.astype(np.float64) - ref) > (tol(y) if half else 0))
                    print("   plan", plan, "bad values", len(d), "pixel mod 16", sorted(set(((d[:, 1] * y.shape[2] + d[:, 2]) % 16).tolist()))[:8], "couts mod 4", sorted(set((d[:, 3] % 4).tolist())))
    bad += nb
    print(f"case {case}: n={n} {h}x{w} {cin}->{cout} k{k}s{stride} half={half} silu={silu} res={res} out_f32={out_f32} plans={npl} bad={nb}", flush=True)
print("bad plan runs:", bad)
