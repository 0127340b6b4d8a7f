"""Debug: one half=True conv shape, every launch plan, fp32 output: where does a plan differ from plan 0 / the float64 reference?
    python tools/dbg_conv_plans.py n h w cin cout k [out_f32=1] [silu=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvsd_amd import ops
n, h, w, cin, cout, k = (int(v) for v in sys.argv[1:7])
out_f32 = bool(int(sys.argv[7])) if len(sys.argv) > 7 else True
silu = bool(int(sys.argv[8])) if len(sys.argv) > 8 else False
rng = np.random.default_rng(5)
x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
wt = (rng.standard_normal((cout, cin, k, k), dtype=np.float32) / np.sqrt(cin * k * k)).astype(np.float32)
b = rng.standard_normal(cout).astype(np.float32) * 0.1
x16, w16 = x.astype(np.float16).astype(np.float64), wt.astype(np.float16).astype(np.float64)
if k == 1:
    want = np.einsum("nhwc,oc->nhwo", x16, w16[:, :, 0, 0]) + b
else:
    xp = np.pad(x16, ((0, 0), (1, 1), (1, 1), (0, 0)))
    want = sum(np.einsum("nhwc,oc->nhwo", xp[:, i:i + h, j:j + w], w16[:, :, i, j]) for i in range(3) for j in range(3)) + b
if silu:
    want = want / (1 + np.exp(-want))
y0, npl = ops.conv2d(x, wt, b, stride=1, silu=silu, half=True, out_f32=out_f32, return_n_plans=True)
print("plans:", npl)
for rep in range(3):
    for plan in range(npl):
        y = ops.conv2d(x, wt, b, stride=1, silu=silu, half=True, out_f32=out_f32, plan=plan)
        err = np.abs(y - want)
        tol = 1e-4 * max(1.0, np.abs(want).max()) if out_f32 else 2e-3 * np.abs(want) + 1e-3
        bad = np.argwhere(err > tol)
        if len(bad):
            px = bad[:, 1] * w + bad[:, 2]
            print(f"rep {rep} plan {plan}: {len(bad)} bad values; pixel mod 16: {sorted(set((px % 16).tolist()))}; couts mod 4: {sorted(set((bad[:, 3] % 4).tolist()))}; "
                  f"couts: {sorted(set(bad[:, 3].tolist()))[:12]}; sample got {y[tuple(bad[0])]:.4f} want {want[tuple(bad[0])]:.4f}")
print("done")
import ctypes as C
from cvsd_amd import _lib
lib = _lib.lib()
ms, npl2 = C.c_float(), C.c_int()
desc = C.create_string_buffer(256)
for i in range(npl):
    _lib.check(lib.mi355_bench_conv2d_f16(0, n, h, w, cin, cout, k, 1, int(silu), 0, i, 2, C.byref(ms), C.byref(npl2), desc, 256))
    print("plan", i, desc.value.decode())
