"""numpy wrappers over the single-operator C-ABI entry points (``mi355_op_*``).

They exist so the parity tests can isolate one HIP kernel at a time; each call goes to the GPU
(there is no CPU implementation behind them).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def conv2d(x_nhwc: np.ndarray, w_oihw: np.ndarray, bias: np.ndarray, stride: int = 1, silu: bool = True,
           residual: Optional[np.ndarray] = None, device: int = 0, plan: int = 0, return_n_plans: bool = False,
           half: bool = False, out_f32: bool = False):
    """Fused conv + bias (+SiLU) (+residual): x [N,H,W,Cin] fp32 -> [N,H/s,W/s,Cout].
    ``plan`` picks one of the engine's candidate launch plans (all must give identical bits).
    ``half``: the half=True kernels -- x, w, residual rounded to fp16, fp32 accumulate, y rounded to fp16 once
    (``out_f32``: y kept in fp32, as for the head's final convs); values are returned as fp32 either way."""
    x, w, b = _f32(x_nhwc), _f32(w_oihw), _f32(bias)
    n, h, wd, cin = x.shape
    cout, cin2, k, k2 = w.shape
    if cin2 != cin or k != k2 or b.shape != (cout,):
        raise ValueError("shape mismatch between x, w and bias")
    y = np.empty((n, h // stride, wd // stride, cout), dtype=np.float32)
    npl = C.c_int(0)
    r = None
    if residual is not None:
        r = _f32(residual)
        if r.shape != y.shape:
            raise ValueError("residual must have the output's shape")
    if half:
        _lib.check(_lib.lib().mi355_op_conv2d_f16(device, x.ctypes.data, n, h, wd, cin, w.ctypes.data, b.ctypes.data, cout,
                                                  k, stride, int(silu), r.ctypes.data if r is not None else None,
                                                  y.ctypes.data, int(out_f32), int(plan), C.byref(npl)))
        return (y, npl.value) if return_n_plans else y
    _lib.check(_lib.lib().mi355_op_conv2d(device, x.ctypes.data, n, h, wd, cin, w.ctypes.data, b.ctypes.data, cout, k,
                                          stride, int(silu), r.ctypes.data if r is not None else None, y.ctypes.data,
                                          int(plan), C.byref(npl)))
    return (y, npl.value) if return_n_plans else y


def conv1x1_upcat(x_half: np.ndarray, x_skip: np.ndarray, w_oihw: np.ndarray, bias: np.ndarray, silu: bool = True, device: int = 0,
                  plan: int = 0, return_n_plans: bool = False, half: bool = False):
    """Pointwise conv over cat(upsample2x(x_half), x_skip) with the upsample fused into the read side (fp32):
    x_half [N,H/2,W/2,Cu], x_skip [N,H,W,Cs], w [Cout,Cu+Cs,1,1] -> [N,H,W,Cout].  ``half``: the half=True kernels."""
    xh, xs, w, b = _f32(x_half), _f32(x_skip), _f32(w_oihw), _f32(bias)
    n, h, wd, cs = xs.shape
    cu, cout = xh.shape[3], w.shape[0]
    if xh.shape != (n, h // 2, wd // 2, cu) or w.shape != (cout, cu + cs, 1, 1) or b.shape != (cout,):
        raise ValueError("shape mismatch")
    y = np.empty((n, h, wd, cout), dtype=np.float32)
    npl = C.c_int(0)
    fn = _lib.lib().mi355_op_conv1x1_upcat_f16 if half else _lib.lib().mi355_op_conv1x1_upcat
    _lib.check(fn(device, xh.ctypes.data, xs.ctypes.data, n, h, wd, cu, cs, w.ctypes.data, b.ctypes.data,
                                                 cout, int(silu), y.ctypes.data, int(plan), C.byref(npl)))
    return (y, npl.value) if return_n_plans else y


def conv2d_fused(x_nhwc: np.ndarray, w1: np.ndarray, b1: np.ndarray, w2: np.ndarray, b2: np.ndarray, stride: int = 1, silu2: bool = False,
                 device: int = 0, plan: int = 0, return_n_plans: bool = False, half: bool = False, out_f32: bool = False):
    """Conv3x3 + bias + SiLU -> Conv1x1 + bias (+SiLU) as one fused launch: x [N,H,W,Cin] -> [N,H/s,W/s,C2].
    ``half``: the half=True kernels (see conv2d); ``out_f32``: the pointwise stage writes fp32 (head finals)."""
    x, w1, b1, w2, b2 = _f32(x_nhwc), _f32(w1), _f32(b1), _f32(w2), _f32(b2)
    n, h, wd, cin = x.shape
    c1, c2 = w1.shape[0], w2.shape[0]
    if w1.shape != (c1, cin, 3, 3) or w2.shape != (c2, c1, 1, 1) or b1.shape != (c1,) or b2.shape != (c2,):
        raise ValueError("shape mismatch")
    y = np.empty((n, h // stride, wd // stride, c2), dtype=np.float32)
    npl = C.c_int(0)
    if half:
        _lib.check(_lib.lib().mi355_op_conv2d_fused_f16(device, x.ctypes.data, n, h, wd, cin, w1.ctypes.data, b1.ctypes.data, c1, stride,
                                                        w2.ctypes.data, b2.ctypes.data, c2, int(silu2), y.ctypes.data, int(out_f32),
                                                        int(plan), C.byref(npl)))
        return (y, npl.value) if return_n_plans else y
    _lib.check(_lib.lib().mi355_op_conv2d_fused(device, x.ctypes.data, n, h, wd, cin, w1.ctypes.data, b1.ctypes.data, c1, stride,
                                                w2.ctypes.data, b2.ctypes.data, c2, int(silu2), y.ctypes.data, int(plan), C.byref(npl)))
    return (y, npl.value) if return_n_plans else y


def stem(bgr_u8: np.ndarray, w_oihw: np.ndarray, bias: np.ndarray, stride: int = 2, device: int = 0, half: bool = False,
         variant: int = 0) -> np.ndarray:
    """uint8 BGR frames [N,H,W,3] -> /255, RGB -> conv kxk + bias + SiLU -> [N,H/s,W/s,Cout].
    half=True: the half predictor's arithmetic (input, weights and output rounded to fp16, fp32 accumulation), float16 result;
    variant 0 = the kernel the engine launches, 1 = the general kernel, 2 = the k 3 / stride 2 kernel."""
    img = np.ascontiguousarray(bgr_u8, dtype=np.uint8)
    w, b = _f32(w_oihw), _f32(bias)
    n, h, wd, _ = img.shape
    cout, _, k, _ = w.shape
    if half:
        yh = np.empty((n, h // stride, wd // stride, cout), dtype=np.float16)
        _lib.check(_lib.lib().mi355_op_stem_f16(device, img.ctypes.data, n, h, wd, w.ctypes.data, b.ctypes.data, cout, k,
                                                stride, int(variant), yh.ctypes.data))
        return yh
    y = np.empty((n, h // stride, wd // stride, cout), dtype=np.float32)
    _lib.check(_lib.lib().mi355_op_stem(device, img.ctypes.data, n, h, wd, w.ctypes.data, b.ctypes.data, cout, k,
                                        stride, y.ctypes.data))
    return y


def letterbox_shape(h: int, w: int, imgsz: int = 640):
    oh, ow = C.c_int(), C.c_int()
    _lib.check(_lib.lib().mi355_letterbox_shape(h, w, imgsz, C.byref(oh), C.byref(ow)))
    return oh.value, ow.value


def letterbox(bgr_u8: np.ndarray, imgsz: int = 640, device: int = 0) -> np.ndarray:
    img = np.ascontiguousarray(bgr_u8, dtype=np.uint8)
    if img.ndim == 3:
        img = img[None]
    n, h, w, _ = img.shape
    oh, ow = letterbox_shape(h, w, imgsz)
    out = np.empty((n, oh, ow, 3), dtype=np.uint8)
    _lib.check(_lib.lib().mi355_op_letterbox(device, img.ctypes.data, n, h, w, imgsz, out.ctypes.data))
    return out


def nms(pred: np.ndarray, nc: int, conf: float = 0.25, iou: float = 0.7, classes: Optional[Sequence[int]] = None,
        max_det: int = 300, device: int = 0):
    """non_max_suppression on pred [N, 4+nc+extra, A] -> list of (rows [n,6+extra], anchor_idx [n])."""
    p = _f32(pred)
    n, no, a = p.shape
    extra = no - 4 - nc
    rows = np.zeros((n, max_det, _lib.DET_WORDS), dtype=np.float32)
    counts = np.zeros(n, dtype=np.int32)
    cls_arr, ncls = None, 0
    if classes is not None:
        cls_arr = (C.c_int * len(classes))(*[int(c) for c in classes])
        ncls = len(classes)
    _lib.check(_lib.lib().mi355_op_nms(device, p.ctypes.data, n, nc, extra, a, conf, iou, cls_arr, ncls, max_det,
                                       rows.ctypes.data, max_det, counts.ctypes.data_as(C.POINTER(C.c_int))))
    out = []
    for i in range(n):
        r = rows[i, :counts[i]]
        ints = r[:, 5:7].view(np.int32)
        out.append((np.concatenate([r[:, :5], ints[:, :1].astype(np.float32), r[:, 7:7 + extra]], 1), ints[:, 1].copy()))
    return out


def plan_versions(n: int, h: int, w: int, cin: int, cout: int, k: int, stride: int = 1, src_cs: int = 0, dst_cs: int = 0, res_cs: int = 0,
                  f2_cout: int = 0, f2_dst_cs: int = 0, half: bool = False):
    """Kernel versions of the candidate launch plans the planner offers for this conv (host-only query, runs without a GPU):
    1 LDS-staged implicit GEMM, 3 streaming pointwise, 4 pipelined pointwise, 6 split-K; + 100 = with the fused pointwise stage."""
    cap = 4096
    out = (C.c_int * cap)()
    npl = C.c_int(0)
    r4 = lambda c: (c + 3) // 4 * 4
    _lib.check(_lib.lib().mi355_plan_query(n, h, w, cin, cout, k, stride, src_cs or r4(cin), dst_cs or r4(cout), res_cs, f2_cout,
                                           f2_dst_cs or r4(f2_cout), int(half), out, cap, C.byref(npl)))
    return [out[i] for i in range(min(cap, npl.value))]


def memory_plan(blob: bytes, n: int, height: int, width: int, imgsz: int = 640, half: bool = False, reuse: bool = True):
    """Where the engine would place the activation buffers of weight image ``blob`` for ``n`` frames per pass (host-only query):
    -> (offsets [n_buffers], sizes [n_buffers], arena_bytes, unshared_bytes)."""
    cap = 4096
    off, sz = (C.c_longlong * cap)(), (C.c_longlong * cap)()
    nb, arena, plain = C.c_int(0), C.c_longlong(0), C.c_longlong(0)
    _lib.check(_lib.lib().mi355_memory_plan(blob, len(blob), n, height, width, imgsz, int(half), int(reuse), off, sz, cap, C.byref(nb),
                                            C.byref(arena), C.byref(plain)))
    k = nb.value
    return np.array(off[:k], dtype=np.int64), np.array(sz[:k], dtype=np.int64), arena.value, plain.value


def conv2d_group(x_nhwc: np.ndarray, wa: np.ndarray, ba: np.ndarray, wb: np.ndarray, bb: np.ndarray, stride_a: int = 1, stride_b: int = 1,
                 plan_a: int = 0, plan_b: int = 0, device: int = 0, w2a: Optional[np.ndarray] = None, b2a: Optional[np.ndarray] = None):
    """Two independent convs of one input as ONE grouped launch (conv_f32_group.hip): -> (ya, yb, n_menu_a, n_menu_b).
    ``w2a`` / ``b2a``: a pointwise conv fused behind conv a (3x3), as in :func:`conv2d_fused`; ya is then its output."""
    x, wa, ba, wb, bb = _f32(x_nhwc), _f32(wa), _f32(ba), _f32(wb), _f32(bb)
    n, h, wd, cin = x.shape
    c2 = 0
    if w2a is not None:
        w2a, b2a = _f32(w2a), _f32(b2a)
        c2 = w2a.shape[0]
    ya = np.empty((n, h // stride_a, wd // stride_a, c2 or wa.shape[0]), dtype=np.float32)
    yb = np.empty((n, h // stride_b, wd // stride_b, wb.shape[0]), dtype=np.float32)
    na, nb = C.c_int(0), C.c_int(0)
    _lib.check(_lib.lib().mi355_op_conv2d_group(device, x.ctypes.data, n, h, wd, cin, wa.ctypes.data, ba.ctypes.data, wa.shape[0], wa.shape[2],
                                                stride_a, wb.ctypes.data, bb.ctypes.data, wb.shape[0], wb.shape[2], stride_b, ya.ctypes.data,
                                                yb.ctypes.data, int(plan_a), int(plan_b), C.byref(na), C.byref(nb),
                                                w2a.ctypes.data if c2 else None, b2a.ctypes.data if c2 else None, c2))
    return ya, yb, na.value, nb.value


def c2f_tail(x_nhwc: np.ndarray, w1: np.ndarray, b1: np.ndarray, lead: np.ndarray, w2: np.ndarray, b2: np.ndarray,
             residual: Optional[np.ndarray] = None, device: int = 0, plan: int = 0, return_n_plans: bool = False):
    """The tail of a C2f block as one fused launch: SiLU(conv1x1(cat(lead, SiLU(conv3x3(x) + b1) + residual)) + b2).
    x [N,H,W,Cin], lead [N,H,W,L] (L and the 3x3's cout multiples of 16), w2 [C2, L + C1, 1, 1]."""
    x, w1, b1, lead, w2, b2 = _f32(x_nhwc), _f32(w1), _f32(b1), _f32(lead), _f32(w2), _f32(b2)
    n, h, wd, cin = x.shape
    c1, c2, L = w1.shape[0], w2.shape[0], lead.shape[-1]
    if w1.shape != (c1, cin, 3, 3) or w2.shape != (c2, L + c1, 1, 1) or lead.shape[:3] != (n, h, wd):
        raise ValueError("shape mismatch")
    r = None
    if residual is not None:
        r = _f32(residual)
        if r.shape != (n, h, wd, c1):
            raise ValueError("residual must have the 3x3 conv's output shape")
    y = np.empty((n, h, wd, c2), dtype=np.float32)
    npl = C.c_int(0)
    _lib.check(_lib.lib().mi355_op_c2f_tail(device, x.ctypes.data, n, h, wd, cin, w1.ctypes.data, b1.ctypes.data, c1,
                                            r.ctypes.data if r is not None else None, lead.ctypes.data, L, w2.ctypes.data, b2.ctypes.data, c2,
                                            y.ctypes.data, int(plan), C.byref(npl)))
    return (y, npl.value) if return_n_plans else y
