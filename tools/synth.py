"""TEST/BENCH INFRASTRUCTURE -- seeded synthetic checkpoints and frames.

There is no network and no ``.pt`` file on the build or GPU boxes, so
benchmarks and parity tests run on a *synthetic checkpoint*: an unfused, Ultralytics-named state
dict (``conv.weight`` + ``bn.{weight,bias,running_mean,running_var}`` per Conv, ``weight``/``bias``
for the head's final convs).  It goes through exactly the path a real checkpoint takes
(weights.fuse_state_dict -> .mi355w -> engine), and the oracle consumes the same dict.

Conv weights, BN gamma/beta come straight from a seeded numpy generator.  BN running statistics are
*calibrated* the way training would set them: the op program is executed once in float64 numpy
(tools/program_ref.py) on two small seeded frames and each BN gets the actual per-channel mean and
variance of its conv output, rounded to 12 mantissa bits so that last-bit BLAS differences between
hosts cannot change the float32 result.  That keeps every layer's activations O(1) with real spatial
variation (a data-free init cannot: with SiLU the variance map has no stable fixed point).  The
head's last layers are scaled/biased so that a few percent of anchors clear conf=0.25 and their
boxes overlap: NMS gets real work.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from cvsd_amd.graph import ACT_NONE, Program, build_program, parse_model_name

from .program_ref import ProgramExecutor

BOX_LOGIT_STD = 0.5
CLS_LOGIT_STD = 1.0
KPT_STD = 0.25
ANCHOR_PASS_FRACTION = 0.015   # target share of anchors whose best class clears conf=0.25 (on the calibration frames)
CALIB_SIZE = 640
CALIB_SEED = 20251226


def _q12(v: np.ndarray) -> np.ndarray:
    """round to 12 mantissa bits (host-independent float32 results from float64 statistics)."""
    m, e = np.frexp(np.asarray(v, dtype=np.float64))
    return np.ldexp(np.round(m * 4096.0) / 4096.0, e)


def synthetic_state_dict(prog: Program, seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    f32 = np.float32
    frames = synthetic_frames(2, CALIB_SIZE, CALIB_SIZE, seed=CALIB_SEED)
    x = frames[..., ::-1].astype(np.float64) / 255.0

    def get_wb(ci: int, src: np.ndarray):
        from .program_ref import conv_nhwc
        c = prog.convs[ci]
        fan_in = c.cin * c.k * c.k
        if c.has_bn:
            w = (rng.standard_normal((c.cout, c.cin, c.k, c.k)) / np.sqrt(fan_in)).astype(f32)
            gamma = rng.uniform(0.7, 1.3, c.cout).astype(f32)
            beta = (rng.standard_normal(c.cout) * 0.5).astype(f32)
            y = conv_nhwc(src, w.astype(np.float64), c.s, c.pad)
            rmean = _q12(y.mean((0, 1, 2))).astype(f32)
            rvar = _q12(y.var((0, 1, 2)) + 1e-6).astype(f32)
            sd[f"{c.name}.conv.weight"] = w
            sd[f"{c.name}.bn.weight"] = gamma
            sd[f"{c.name}.bn.bias"] = beta
            sd[f"{c.name}.bn.running_mean"] = rmean
            sd[f"{c.name}.bn.running_var"] = rvar
            scale = gamma.astype(np.float64) / np.sqrt(1e-3 + rvar.astype(np.float64))
            return w.astype(np.float64) * scale[:, None, None, None], beta - rmean * scale
        branch = c.name.split(".")[2]                                  # cv2 (box) / cv3 (cls) / cv4 (kpt)
        t = {"cv2": BOX_LOGIT_STD, "cv3": CLS_LOGIT_STD}.get(branch, KPT_STD)
        m2 = float(_q12((src * src).mean()))
        w = (rng.standard_normal((c.cout, c.cin, c.k, c.k)) * (t / np.sqrt(fan_in * m2))).astype(f32)
        y = conv_nhwc(src, w.astype(np.float64), c.s, c.pad)
        const = y.mean((0, 1, 2))
        if branch == "cv2":
            bias = 1.0 - const                                           # Detect.bias_init: box bias = 1.0
        elif branch == "cv3":
            best = (y - const).max(-1).ravel()                           # best-class logit per anchor, offsets removed
            thr = np.quantile(best, 1.0 - ANCHOR_PASS_FRACTION)
            bias = np.log(0.25 / 0.75) - thr - const
        else:
            bias = -const
        bias = (np.round(np.asarray(bias) * 256.0) / 256.0).astype(f32)
        sd[f"{c.name}.weight"] = w
        sd[f"{c.name}.bias"] = bias
        return w.astype(np.float64), bias.astype(np.float64)

    ProgramExecutor(prog, np.float64).run(x, get_wb)
    sd["model.%d.dfl.conv.weight" % (22 if prog.family == "v8" else 24)] = \
        np.arange(16, dtype=f32).reshape(1, 16, 1, 1)
    return sd


_CACHE: Dict[tuple, tuple] = {}


def synthetic_checkpoint(name: str = "yolov8n", seed: int = 0, nc: int | None = None):
    """-> (Program, unfused state dict).  Cached per process."""
    key = (name, seed, nc)
    if key not in _CACHE:
        family, scale, task = parse_model_name(name)
        prog = build_program(family, scale, task, nc=nc)
        _CACHE[key] = (prog, synthetic_state_dict(prog, seed))
    return _CACHE[key]


def synthetic_frames(n: int, h: int = 640, w: int = 640, seed: int = 0) -> np.ndarray:
    """uint8 BGR frames [n,h,w,3] (the layout ``cv2.VideoCapture.read`` hands to
    ``/root/reference/preprocess.py:38``).  Every frame is a seeded mix of a gradient background,
    filled rectangles of random colour and per-pixel noise, so frames have comparable statistics
    while detections cluster around the rectangles and overlap."""
    frames = np.empty((n, h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(n):
        rng = np.random.default_rng([seed, i])
        fx, fy = rng.integers(1, 5, size=2)
        base = ((xx * int(fx) + yy * int(fy)) % 256).astype(np.int32)
        img = np.stack([base, base[::-1], base[:, ::-1]], -1)
        for _ in range(int(rng.integers(4, 10))):
            x0, y0 = int(rng.integers(0, max(1, w - 8))), int(rng.integers(0, max(1, h - 8)))
            x1 = min(w, x0 + int(rng.integers(8, max(9, w // 2))))
            y1 = min(h, y0 + int(rng.integers(8, max(9, h // 2))))
            img[y0:y1, x0:x1] = rng.integers(0, 256, size=3)
        noise = rng.integers(-48, 49, size=(h, w, 3))
        frames[i] = np.clip(img + noise, 0, 255).astype(np.uint8)
    return frames


def synthetic_clip(n: int, h: int, w: int, seed: int = 0) -> np.ndarray:
    """A short 'video' [n,h,w,3]: ONE synthetic scene that drifts by one pixel every second frame (consecutive
    detections overlap, so a tracker keeps its ids) plus a little per-frame sensor noise.  Used by the PoseLift fixture
    (tests/golden/make_poselift_fixture.py) and the GPU test that must reproduce it."""
    base = synthetic_frames(1, h + 16, w + 16, seed=seed)[0]
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        d = k // 2
        f = base[d:d + h, d:d + w].astype(np.int16)
        f += rng.integers(-2, 3, size=f.shape, dtype=np.int16)
        out.append(np.clip(f, 0, 255).astype(np.uint8))
    return np.stack(out)
