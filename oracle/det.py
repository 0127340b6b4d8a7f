"""CPU ORACLE, TEST INFRASTRUCTURE ONLY -- the module-by-module YOLO forward of yolo_oracle.py evaluated with the
deterministic-order C kernels of det_oracle.c (canonical fp32 operation order, libm-free exp).

``DetOracleModel`` inherits the graph walk (parse_model / _predict_once restatement, C2f / C3 / SPPF / Bottleneck
structure, NMS, scale-back) from ``OracleModel`` and replaces only the floating-point leaves: conv+bias+SiLU, the
u8 stem and the head decode.  Max-pool, nearest upsample, concat and chunk are exact in any implementation.
An implementation that follows the canonical order can be compared with this oracle bit for bit; this oracle in
turn is compared with the torch one within fp32 re-association noise (tests/test_oracle_det.py).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import yolo_oracle as O
from .build import LIB, build

_lib = None
_fp = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.det_expf.restype = C.c_float
        L.det_expf.argtypes = [C.c_float]
        L.det_expf_array.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.det_silu_array.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.det_conv2d.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
        L.det_stem.argtypes = [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
        L.det_decode_level.argtypes = [C.c_void_p] * 3 + [C.c_int] * 9 + [C.c_void_p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def conv2d(x_nhwc, w_oihw, bias, stride=1, pad=None, act=True, residual=None) -> np.ndarray:
    x, w, b = _f32(x_nhwc), _f32(w_oihw), _f32(bias)
    n, h, wd, cin = x.shape
    cout, _, k, _ = w.shape
    pad = k // 2 if pad is None else pad
    y = np.empty((n, h // stride, wd // stride, cout), np.float32)
    r = _f32(residual) if residual is not None else None
    lib().det_conv2d(x.ctypes.data, n, h, wd, cin, w.ctypes.data, b.ctypes.data, cout, k, stride, pad, int(act),
                     r.ctypes.data if r is not None else None, y.ctypes.data)
    return y


def stem(bgr_u8, w_oihw, bias, stride=2, pad=None) -> np.ndarray:
    img = np.ascontiguousarray(bgr_u8, dtype=np.uint8)
    w, b = _f32(w_oihw), _f32(bias)
    n, h, wd, _ = img.shape
    cout, _, k, _ = w.shape
    pad = (2 if k == 6 else k // 2) if pad is None else pad
    y = np.empty((n, h // stride, wd // stride, cout), np.float32)
    lib().det_stem(img.ctypes.data, n, h, wd, w.ctypes.data, b.ctypes.data, cout, k, stride, pad, y.ctypes.data)
    return y


def expf(x) -> np.ndarray:
    x = _f32(x)
    y = np.empty_like(x)
    lib().det_expf_array(x.ctypes.data, y.ctypes.data, x.size)
    return y


def _nhwc(t: torch.Tensor) -> np.ndarray:
    return np.ascontiguousarray(t.permute(0, 2, 3, 1).numpy())


def _nchw(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a).permute(0, 3, 1, 2)


class DetOracleModel(O.OracleModel):
    """Same graph walk as OracleModel; fp32 leaves in canonical order."""

    _u8 = None

    def _fused_conv(self, prefix: str):
        """fuse_conv_and_bn with every operation a correctly rounded IEEE fp32 operation (numpy): torch's vectorised
        sqrt is not correctly rounded on every host, which would make the folded weights host-dependent.
            scale = gamma / sqrt(eps + var);  w' = w * scale;  b' = beta - (gamma * mean) / sqrt(var + eps)"""
        if prefix not in self._fused:
            f32 = np.float32
            w = self.sd[prefix + ".conv.weight"].numpy()
            g, b = self.sd[prefix + ".bn.weight"].numpy(), self.sd[prefix + ".bn.bias"].numpy()
            mu, var = self.sd[prefix + ".bn.running_mean"].numpy(), self.sd[prefix + ".bn.running_var"].numpy()
            eps = f32(1e-3)
            scale = (g / np.sqrt(eps + var)).astype(f32)
            wf = (w * scale[:, None, None, None]).astype(f32)
            bf = (b - (g * mu) / np.sqrt(var + eps)).astype(f32)
            self._fused[prefix] = (torch.from_numpy(wf), torch.from_numpy(bf))
        return self._fused[prefix]

    def Conv(self, x, prefix, k, s, p=None):
        w, b = self._fused_conv(prefix)
        if w.shape[1] == 3:                       # the stem reads the letterboxed uint8 frame
            return _nchw(stem(self._u8, w.numpy(), b.numpy(), stride=s, pad=p))
        return _nchw(conv2d(_nhwc(x), w.numpy(), b.numpy(), stride=s, pad=p, act=True))

    def Bottleneck(self, x, prefix, shortcut, k=(3, 3)):
        y = self.Conv(x, prefix + ".cv1", k[0], 1)
        w, b = self._fused_conv(prefix + ".cv2")
        # the shortcut add is fused behind the activation: silu(conv + bias) + x
        return _nchw(conv2d(_nhwc(y), w.numpy(), b.numpy(), act=True, residual=_nhwc(x) if shortcut else None))

    def _seq3(self, x, prefix):
        x = self.Conv(self.Conv(x, prefix + ".0", 3, 1), prefix + ".1", 3, 1)
        return _nchw(conv2d(_nhwc(x), self.sd[prefix + ".2.weight"].numpy(), self.sd[prefix + ".2.bias"].numpy(), act=False))

    def Head(self, feats: List[torch.Tensor], prefix: str):
        n = feats[0].shape[0]
        no = 4 + self.nc + self.nk
        a_total = sum(f.shape[2] * f.shape[3] for f in feats)
        pred = np.zeros((n, no, a_total), np.float32)
        a0 = 0
        for i, (f, s) in enumerate(zip(feats, self.stride)):
            box = _nhwc(self._seq3(f, f"{prefix}.cv2.{i}"))
            cls = _nhwc(self._seq3(f, f"{prefix}.cv3.{i}"))
            kpt = _nhwc(self._seq3(f, f"{prefix}.cv4.{i}")) if self.pose else None
            h, w = f.shape[2:]
            lib().det_decode_level(box.ctypes.data, cls.ctypes.data, kpt.ctypes.data if kpt is not None else None, n, h, w,
                                   self.nc, self.kpt_shape[0], self.kpt_shape[1], s, a0, a_total, pred.ctypes.data)
            a0 += h * w
        return torch.from_numpy(pred)

    @torch.no_grad()
    def forward_u8(self, letterboxed_bgr_u8: np.ndarray) -> torch.Tensor:
        """[N,H,W,3] uint8 BGR (already letterboxed) -> [N, no, A]"""
        self._u8 = np.ascontiguousarray(letterboxed_bgr_u8)
        n, h, w, _ = self._u8.shape
        dummy = torch.zeros((n, 3, h, w))
        try:
            return self.forward(dummy)
        finally:
            self._u8 = None


@torch.no_grad()
def predict(model: DetOracleModel, frames: Sequence[np.ndarray], conf=0.25, iou=0.7, classes=None, max_det=300, imgsz=640):
    """Same as yolo_oracle.predict with the deterministic forward."""
    lb = np.stack([O.letterbox(f, (imgsz, imgsz)) for f in frames])
    pred = model.forward_u8(lb)
    rows, idxs = O.non_max_suppression(pred, conf, iou, classes=classes, max_det=max_det, nc=model.nc, return_idxs=True)
    out = []
    for r, ai, f in zip(rows, idxs, frames):
        r = r.clone()
        r[:, :4] = O.scale_boxes(lb.shape[1:3], r[:, :4], f.shape)
        k = None
        if model.pose:
            k = r[:, 6:].view(len(r), *model.kpt_shape).clone()
            k = O.scale_coords(lb.shape[1:3], k, f.shape)
        out.append({"boxes": r[:, :6], "kpts": k, "anchor_idx": ai})
    return out, pred
