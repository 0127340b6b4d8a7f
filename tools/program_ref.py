"""TEST/BENCH INFRASTRUCTURE -- numpy executor of the op program (graph.Program).

Runs the fused-op program exactly as the engine does (NHWC buffers, channel-slice reads and
writes, residual fused into the conv epilogue) but on the CPU in float64/float32 numpy.  Two uses:
  * tools/synth.py calibrates the synthetic checkpoint's BatchNorm statistics with it;
  * tests/test_program.py checks that the program (channel-offset plumbing of graph.py) computes the
    same function as the module-by-module oracle -- before anything touches a GPU.
It is never imported by the product package.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from cvsd_amd.graph import ACT_SILU, OP_CONV, OP_SPPF_POOL, OP_STEM, OP_UPSAMPLE, Program


def conv_nhwc(x: np.ndarray, w: np.ndarray, stride: int, pad: int) -> np.ndarray:
    """x [N,H,W,Cin], w [Cout,Cin,k,k] -> [N,Ho,Wo,Cout] (cross-correlation, zero padding)."""
    n, h, wd, cin = x.shape
    cout, _, k, _ = w.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
    xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    out = np.zeros((n, ho, wo, cout), dtype=x.dtype)
    for kh in range(k):
        for kw in range(k):
            patch = xp[:, kh:kh + (ho - 1) * stride + 1:stride, kw:kw + (wo - 1) * stride + 1:stride, :]
            out += patch @ w[:, :, kh, kw].T.astype(x.dtype)
    return out


def silu(x: np.ndarray) -> np.ndarray:
    return x / (1.0 + np.exp(-x))


def maxpool5(x: np.ndarray) -> np.ndarray:
    n, h, w, c = x.shape
    xp = np.pad(x, ((0, 0), (2, 2), (2, 2), (0, 0)), constant_values=-np.inf)
    out = np.full_like(x, -np.inf)
    for dy in range(5):
        for dx in range(5):
            out = np.maximum(out, xp[:, dy:dy + h, dx:dx + w, :])
    return out


class ProgramExecutor:
    """Walk the program op by op.  `get_wb(conv_index, conv_input)` supplies the fused (w, b) of a conv
    (letting the caller calibrate it from its actual input first)."""

    def __init__(self, prog: Program, dtype=np.float64):
        self.prog, self.dtype = prog, dtype
        self.bufs: List[Optional[np.ndarray]] = []

    def run(self, x_rgb01: np.ndarray, get_wb: Callable[[int, np.ndarray], Tuple[np.ndarray, np.ndarray]]):
        """x_rgb01: [N,H,W,3] RGB in [0,1].  Returns the list of buffers (NHWC)."""
        prog = self.prog
        n, h, w, _ = x_rgb01.shape
        self.bufs = [np.zeros((n, h // sd, w // sd, c), self.dtype) for c, sd in prog.buffers]
        x0 = x_rgb01.astype(self.dtype)
        for op in prog.ops:
            if op.type in (OP_STEM, OP_CONV):
                c = prog.convs[op.conv]
                src = x0 if op.type == OP_STEM else self.bufs[op.src.buf][..., op.src.choff:op.src.choff + op.src.c]
                wt, b = get_wb(op.conv, src)
                y = conv_nhwc(src, wt.astype(self.dtype), c.s, c.pad) + b.astype(self.dtype)
                if op.act == ACT_SILU:
                    y = silu(y)
                if op.res is not None:
                    y = y + self.bufs[op.res.buf][..., op.res.choff:op.res.choff + op.dst.c]
                self.bufs[op.dst.buf][..., op.dst.choff:op.dst.choff + op.dst.c] = y
            elif op.type == OP_UPSAMPLE:
                src = self.bufs[op.src.buf][..., op.src.choff:op.src.choff + op.src.c]
                self.bufs[op.dst.buf][..., op.dst.choff:op.dst.choff + op.dst.c] = src.repeat(2, 1).repeat(2, 2)
            elif op.type == OP_SPPF_POOL:
                c = op.src.c
                cur = self.bufs[op.src.buf][..., op.src.choff:op.src.choff + c]
                for j in range(3):
                    cur = maxpool5(cur)
                    self.bufs[op.dst.buf][..., op.dst.choff + j * c:op.dst.choff + (j + 1) * c] = cur
            else:
                raise ValueError(op.type)
        return self.bufs

    def head_maps(self):
        """per level: (box [N,h,w,64], cls [N,h,w,nc], kpt [N,h,w,nk] or None)"""
        out = []
        p = self.prog
        for lv in p.levels:
            b = self.bufs[lv.buf]
            out.append((b[..., lv.box_off:lv.box_off + 64], b[..., lv.cls_off:lv.cls_off + p.nc],
                        b[..., lv.kpt_off:lv.kpt_off + p.nk] if p.nk else None))
        return out


def decode_head(prog: Program, maps, dtype=np.float64) -> np.ndarray:
    """Detect/Pose inference decode (DFL, dist2bbox, sigmoid, kpts_decode) of NHWC head maps -> [N, no, A]."""
    outs = []
    for lv, (box, cls, kpt) in zip(prog.levels, maps):
        n, h, w, _ = box.shape
        ax, ay = np.meshgrid(np.arange(w, dtype=dtype) + 0.5, np.arange(h, dtype=dtype) + 0.5)
        bl = box.reshape(n, h, w, 4, 16).astype(dtype)
        e = np.exp(bl - bl.max(-1, keepdims=True))
        dist = (e / e.sum(-1, keepdims=True) * np.arange(16, dtype=dtype)).sum(-1)        # [n,h,w,4] l,t,r,b
        x1, y1 = ax - dist[..., 0], ay - dist[..., 1]
        x2, y2 = ax + dist[..., 2], ay + dist[..., 3]
        parts = [np.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], -1) * lv.stride,
                 1.0 / (1.0 + np.exp(-cls.astype(dtype)))]
        if kpt is not None:
            k = kpt.reshape(n, h, w, prog.nkpt, prog.kdim).astype(dtype).copy()
            k[..., 0] = (k[..., 0] * 2.0 + (ax[..., None] - 0.5)) * lv.stride
            k[..., 1] = (k[..., 1] * 2.0 + (ay[..., None] - 0.5)) * lv.stride
            if prog.kdim == 3:
                k[..., 2] = 1.0 / (1.0 + np.exp(-k[..., 2]))
            parts.append(k.reshape(n, h, w, -1))
        outs.append(np.concatenate(parts, -1).reshape(n, h * w, -1))
    return np.concatenate(outs, 1).transpose(0, 2, 1)
