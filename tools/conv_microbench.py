"""Diagnostics: time conv launch plans on the GPU (device-resident, random data).
    python tools/conv_microbench.py n h w cin cout k stride [silu=1] [residual=0] [max_plans=0(all)]
    MI355_BENCH_HALF=1 times the half=True kernels instead."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvsd_amd import _lib
a = [int(x) for x in sys.argv[1:8]]
silu = int(sys.argv[8]) if len(sys.argv) > 8 else 1
res = int(sys.argv[9]) if len(sys.argv) > 9 else 0
maxp = int(sys.argv[10]) if len(sys.argv) > 10 else 0
n, h, w, cin, cout, k, s = a
lib = _lib.lib()
fn = lib.mi355_bench_conv2d_f16 if os.environ.get('MI355_BENCH_HALF') == '1' else lib.mi355_bench_conv2d
ms, npl = C.c_float(), C.c_int()
desc = C.create_string_buffer(256)
fl = 2.0 * n * (h // s) * (w // s) * cout * cin * k * k
rows = []
i = 0
flt = os.environ.get("MB_FILTER")          # e.g. "v7": time only the plans whose description starts with it (one probe launch tells)
while True:
    if flt:
        _lib.check(fn(0, n, h, w, cin, cout, k, s, silu, res, i, 1, C.byref(ms), C.byref(npl), desc, 256))
    if not flt or desc.value.decode().startswith(flt):
        _lib.check(fn(0, n, h, w, cin, cout, k, s, silu, res, i, 20, C.byref(ms), C.byref(npl), desc, 256))
        rows.append((ms.value, desc.value.decode()))
    i += 1
    if i >= npl.value or (maxp and i >= maxp):
        break
print(f"conv {cin}->{cout} k{k} s{s} @{h}x{w} n={n} silu={silu} res={res}: {npl.value} plans")
srt = sorted(rows)
for t, d in srt[:int(os.environ.get("MB_TOP", "10"))]:
    print(f"  {t*1e3:9.1f} us  {fl/t/1e9:7.1f} TFLOP/s   {d}")
v2 = [r for r in srt if r[1].startswith("v7")][:6]
if v2:
    print("  v7 plans (weights in LDS, persistent):")
for t, d in v2:
    print(f"  {t*1e3:9.1f} us  {fl/t/1e9:7.1f} TFLOP/s   {d}")
