"""TEST/BENCH INFRASTRUCTURE -- float64 execution of the op program as the accuracy yardstick.

north_star: "bbox and keypoint coordinates within 1e-3, identical box/class indices vs the Ultralytics CPU path".  The
Ultralytics CPU path is fp32 torch, whose own distance from exact arithmetic depends on the host's summation order, so
"distance from torch" cannot be told apart from torch's own noise.  The yardstick used instead is a float64 evaluation of
the SAME fused program (tools/program_ref.py): every fp32 implementation -- the torch oracle, the GPU engine -- is measured
against it, and the engine is required to be as close to it as torch is (tests/test_gpu_precision.py, bench.py).
Never imported by the product package.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


def f64_head(name: str, sd: Dict[str, np.ndarray], frames: np.ndarray, imgsz: int = 640) -> np.ndarray:
    """Pre-NMS head tensor [N, no, A] of `frames` (uint8 BGR) in float64 arithmetic on the product's fp32 fused weights."""
    from cvsd_amd import weights
    from cvsd_amd.graph import build_program, parse_model_name
    from oracle import yolo_oracle as O
    from tools import program_ref as PR
    prog = build_program(*parse_model_name(name))
    fused = weights.fuse_state_dict(prog, sd)
    names = [c.name for c in prog.convs]
    x = O.preprocess(list(frames), imgsz).permute(0, 2, 3, 1).numpy().astype(np.float64)
    ex = PR.ProgramExecutor(prog, np.float64)
    ex.run(x, lambda ci, src: fused[names[ci]])
    return PR.decode_head(prog, ex.head_maps())


def group_errors(pred: np.ndarray, ref64: np.ndarray, nc: int) -> Dict[str, Dict[str, float]]:
    """|pred - ref| per channel group of a [N, no, A] head tensor: box (px), score, kpt_xy (px), kpt_conf."""
    d = np.abs(pred.astype(np.float64) - ref64)
    groups = {"box": d[:, :4], "score": d[:, 4:4 + nc]}
    if d.shape[1] > 4 + nc:
        k = d[:, 4 + nc:].reshape(d.shape[0], -1, 3, d.shape[2])
        groups["kpt_xy"] = k[:, :, :2]
        groups["kpt_conf"] = k[:, :, 2]
    return {g: {"max": float(v.max()), "p999": float(np.quantile(v, 0.999)), "mean": float(v.mean())} for g, v in groups.items()}


def nms_rows(pred, conf: float, iou: float, nc: int, max_det: int = 300):
    """oracle NMS on a head tensor (any float dtype) -> per image (rows [n, 6+], anchor idx [n])"""
    from oracle import yolo_oracle as O
    rows, idxs = O.non_max_suppression(torch.as_tensor(pred), conf, iou, max_det=max_det, nc=nc, return_idxs=True)
    return [(r.numpy(), i.numpy()) for r, i in zip(rows, idxs)]


def _xyxy(pred_img: np.ndarray, a: int) -> np.ndarray:
    cx, cy, w, h = pred_img[:4, a]
    return np.array([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2])


def _iou(b1, b2) -> float:
    iw = max(0.0, min(b1[2], b2[2]) - max(b1[0], b2[0]))
    ih = max(0.0, min(b1[3], b2[3]) - max(b1[1], b2[1]))
    inter = iw * ih
    return inter / ((b1[2] - b1[0]) * (b1[3] - b1[1]) + (b2[2] - b2[0]) * (b2[3] - b2[1]) - inter + 1e-7)


def first_divergence_margin(ref64_img: np.ndarray, kept_ref: Sequence[int], kept_got: Sequence[int], nc: int, conf: float,
                            iou: float) -> Optional[Tuple[int, float, str]]:
    """Where two NMS runs on (nearly) the same head tensor first disagree, how thin was the float64 decision margin there?
    Returns None when the kept anchor lists are identical, else (position, margin, kind): the smallest of
      * |score - conf| of the two anchors involved (threshold decision),
      * |score_a - score_b| (ordering decision),
      * | IoU(anchor, earlier kept box of the same class) - iou | (suppression decision),
    all evaluated in float64.  A divergence is explained by fp32 noise iff that margin is below the noise level; later
    divergences may be consequences of the first and are not examined."""
    kept_ref, kept_got = list(kept_ref), list(kept_got)
    if kept_ref == kept_got:
        return None
    pos = next((i for i, (a, b) in enumerate(zip(kept_ref, kept_got)) if a != b), min(len(kept_ref), len(kept_got)))
    involved = [k[pos] for k in (kept_ref, kept_got) if pos < len(k)]
    sc = ref64_img[4:4 + nc]
    best = (np.inf, "none")
    for a in involved:
        s, c = float(sc[:, a].max()), int(sc[:, a].argmax())
        best = min(best, (abs(s - conf), "conf threshold"))
        box = _xyxy(ref64_img, a)
        for e in kept_ref[:pos]:
            if int(sc[:, e].argmax()) == c:
                best = min(best, (abs(_iou(box, _xyxy(ref64_img, e)) - iou), "iou threshold"))
    if len(involved) == 2:
        a, b = involved
        best = min(best, (abs(float(sc[:, a].max()) - float(sc[:, b].max())), "score order"))
        ca, cb = int(sc[:, a].argmax()), int(sc[:, b].argmax())
        if ca == cb:
            best = min(best, (abs(_iou(_xyxy(ref64_img, a), _xyxy(ref64_img, b)) - iou), "iou threshold"))
    return pos, float(best[0]), best[1]


def peaked_head_state_dict(sd: Dict[str, np.ndarray], amp: float = 1.0, seed: int = 7, n_levels: int = 3,
                           prefix: str = "model.22") -> Dict[str, np.ndarray]:
    """A LOW-ENTROPY variant of a synthetic checkpoint: the bias of the final box convs gets a unimodal profile
    -amp * (bin - mu)^2 per side, so the DFL distributions are peaked around a few bins the way a trained detector's are
    (the random head has near-uniform distributions: entropy 2.2 nats -> 1.0 with amp 1), with the upstream fp32 noise
    unchanged.  The DFL expectation is then far less sensitive to logit noise."""
    out = dict(sd)
    rng = np.random.default_rng(seed)
    for i in range(n_levels):
        k = f"{prefix}.cv2.{i}.2.bias"
        b = np.asarray(out[k], dtype=np.float32).copy().reshape(4, 16)
        mu = rng.uniform(2.0, 9.0, size=(4, 1))
        b += (-amp * (np.arange(16)[None, :] - mu) ** 2).astype(np.float32)
        out[k] = b.reshape(-1)
    return out
