"""``Results`` / ``Boxes`` / ``Keypoints`` containers with the Ultralytics surface the reference uses.

Mirrors ``ultralytics/engine/results.py`` as far as the hot path needs it:
``/root/reference/model.py:40`` reads ``results[0].boxes``; ``:45`` ``boxes.is_track``; ``:67`` iterates
``for box in boxes`` (1-row ``Boxes``); ``:60-64`` ``float(box.id)``, ``float(box.xywhn[0][k])``.
north_star adds ``.keypoints`` for pose models.  Backed by torch CPU tensors, like ``Results.cpu()``.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch


class _BaseTensor:
    def __init__(self, data, orig_shape: Tuple[int, int]):
        self.data = data
        self.orig_shape = orig_shape

    @property
    def shape(self):
        return self.data.shape

    def cpu(self):
        return self if isinstance(self.data, np.ndarray) else self.__class__(self.data.cpu(), self.orig_shape)

    def numpy(self):
        return self if isinstance(self.data, np.ndarray) else self.__class__(self.data.numpy(), self.orig_shape)

    def cuda(self):
        return self.__class__(torch.as_tensor(self.data).cuda(), self.orig_shape)

    def to(self, *args, **kwargs):
        return self.__class__(torch.as_tensor(self.data).to(*args, **kwargs), self.orig_shape)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        d = self.data
        if isinstance(d, torch.Tensor) and not d.is_cuda and isinstance(idx, (np.ndarray, list, torch.Tensor)):
            # row selection through a numpy view of the same memory: torch's CPU advanced-indexing kernel costs 10-60 MILLIseconds
            # for a [300, 17, 3] tensor once its OpenMP pool has many threads (measured: 54 ms at 8 threads against 10 us at 1),
            # which made `res[idx]` of the tracker callback the slowest step of a pose model's frame loop
            ix = idx.numpy() if isinstance(idx, torch.Tensor) else np.asarray(idx)
            return self.__class__(torch.from_numpy(np.ascontiguousarray(d.numpy()[ix])), self.orig_shape)
        return self.__class__(d[idx], self.orig_shape)


def _clone(x):
    return x.copy() if isinstance(x, np.ndarray) else x.clone()


def _empty_like(x):
    return np.empty_like(x) if isinstance(x, np.ndarray) else torch.empty_like(x)


def xyxy2xywh(x):
    """utils/ops.py:xyxy2xywh -> (cx, cy, w, h)"""
    y = _empty_like(x)
    y[..., 0] = (x[..., 0] + x[..., 2]) / 2
    y[..., 1] = (x[..., 1] + x[..., 3]) / 2
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def clip_boxes(boxes, shape):
    """utils/ops.py:clip_boxes -- x1,x2 to [0, w], y1,y2 to [0, h] on the first four columns, in place; shape = (h, w)."""
    h, w = shape[:2]
    if isinstance(boxes, np.ndarray):
        boxes[..., [0, 2]] = boxes[..., [0, 2]].clip(0, w)
        boxes[..., [1, 3]] = boxes[..., [1, 3]].clip(0, h)
    else:
        boxes[..., 0].clamp_(0, w)
        boxes[..., 1].clamp_(0, h)
        boxes[..., 2].clamp_(0, w)
        boxes[..., 3].clamp_(0, h)
    return boxes


class Boxes(_BaseTensor):
    """data: [N, 6] = x1,y1,x2,y2,conf,cls   or   [N, 7] = x1,y1,x2,y2,track_id,conf,cls (tracking)."""

    def __init__(self, boxes, orig_shape):
        if boxes.ndim == 1:
            boxes = boxes[None, :]
        n = boxes.shape[-1]
        assert n in (6, 7), f"expected 6 or 7 values but got {n}"
        super().__init__(boxes, orig_shape)
        self.is_track = n == 7

    @property
    def xyxy(self):
        return self.data[:, :4]

    @property
    def conf(self):
        return self.data[:, -2]

    @property
    def cls(self):
        return self.data[:, -1]

    @property
    def id(self):
        return self.data[:, -3] if self.is_track else None

    @property
    def xywh(self):
        return xyxy2xywh(self.xyxy)

    @property
    def xyxyn(self):
        xyxy = _clone(self.xyxy)
        xyxy[..., [0, 2]] /= self.orig_shape[1]
        xyxy[..., [1, 3]] /= self.orig_shape[0]
        return xyxy

    @property
    def xywhn(self):
        xywh = xyxy2xywh(self.xyxy)
        xywh[..., [0, 2]] /= self.orig_shape[1]
        xywh[..., [1, 3]] /= self.orig_shape[0]
        return xywh


class Keypoints(_BaseTensor):
    """data: [N, K, 3] = x, y, conf (or [N, K, 2]).  Points with conf < 0.5 get x = y = 0 on construction."""

    def __init__(self, keypoints, orig_shape):
        if keypoints.ndim == 2:
            keypoints = keypoints[None, :]
        if keypoints.shape[2] == 3:
            if isinstance(keypoints, torch.Tensor) and not keypoints.is_cuda:
                k = keypoints.numpy()                         # same memory (see _BaseTensor.__getitem__ on torch's CPU indexing kernels)
                k[..., :2][k[..., 2] < 0.5] = 0
            else:
                mask = keypoints[..., 2] < 0.5
                keypoints[..., :2][mask] = 0
        super().__init__(keypoints, orig_shape)
        self.has_visible = self.data.shape[-1] == 3

    @property
    def xy(self):
        return self.data[..., :2]

    @property
    def xyn(self):
        xy = _clone(self.xy)
        xy[..., 0] /= self.orig_shape[1]
        xy[..., 1] /= self.orig_shape[0]
        return xy

    @property
    def conf(self):
        return self.data[..., 2] if self.has_visible else None


class Results:
    """One image's predictions (``list[Results]`` is what ``model(frame)`` / ``model.track(frame)`` return)."""

    def __init__(self, orig_img: Optional[np.ndarray], path: str, names: Dict[int, str], boxes=None, keypoints=None,
                 orig_shape: Optional[Tuple[int, int]] = None, speed: Optional[dict] = None, anchor_idx=None):
        self.orig_img = orig_img
        self.orig_shape = tuple(orig_shape) if orig_shape is not None else tuple(orig_img.shape[:2])
        self.boxes = Boxes(boxes, self.orig_shape) if boxes is not None else None
        self.keypoints = Keypoints(keypoints, self.orig_shape) if keypoints is not None else None
        self.masks = self.probs = self.obb = None
        self.speed = speed or {"preprocess": None, "inference": None, "postprocess": None}
        self.names = names
        self.path = path
        self.anchor_idx = anchor_idx      # engine extra: source anchor of every row (index-identity checks)

    def __len__(self):
        return len(self.boxes) if self.boxes is not None else 0

    def __getitem__(self, idx):
        r = Results(self.orig_img, self.path, self.names, orig_shape=self.orig_shape, speed=self.speed)
        if self.boxes is not None:
            r.boxes = self.boxes[idx]
        if self.keypoints is not None:
            r.keypoints = self.keypoints[idx]
        return r

    def update(self, boxes=None):
        """engine/results.py:Results.update -- the tracker callback replaces the box tensor with track rows, clipped to
        the image like ``Boxes(ops.clip_boxes(boxes, self.orig_shape), self.orig_shape)`` (the rows are Kalman-state boxes
        and may overhang the frame; the reference's CSV reads ``xywhn`` of the clipped ones)."""
        if boxes is not None:
            self.boxes = Boxes(clip_boxes(boxes, self.orig_shape), self.orig_shape)

    def cpu(self):
        return self

    def numpy(self):
        r = Results(self.orig_img, self.path, self.names, orig_shape=self.orig_shape, speed=self.speed)
        r.boxes = self.boxes.numpy() if self.boxes is not None else None
        r.keypoints = self.keypoints.numpy() if self.keypoints is not None else None
        return r
