"""Pose detections -> PoseLift pickle layout consumed by ``shopformer/``.

``/root/reference/shopformer/data/poselift_dataset.py:256-295`` reads, per video, one pickle
``{frame_num: {person_id: [bbox, keypoints]}}`` with ``keypoints`` an array ``(17, 3)`` of (x, y, conf) in pixels and
``bbox`` a 4-vector; windows of ``seq_len`` consecutive frames per ``person_id`` become model inputs (``:297-323``).
The reference's own YOLO stage never produces this file (its CSV goes elsewhere); this writer closes that gap:
frames go through the pose engine in batches, ``person_id`` comes from the host tracker (``cvsd_amd.tracker``), and
keypoints are the RAW decoded values (no zeroing of low-confidence points: PoseLift stores every joint with its conf).
"""
from __future__ import annotations

import pickle
from typing import Dict, Iterable, Optional

import numpy as np


class PoseLiftWriter:
    """Accumulates ``{frame_num: {person_id: [bbox_xywh, kpts(17,3)]}}`` for one video."""

    def __init__(self, bbox_format: str = "xywh"):
        if bbox_format not in ("xywh", "xyxy"):
            raise ValueError("bbox_format must be 'xywh' (top-left x, y, w, h) or 'xyxy'")
        self.bbox_format = bbox_format
        self.data: Dict[int, Dict[int, list]] = {}

    def add_frame(self, frame_num: int, track_rows: np.ndarray, keypoints: np.ndarray) -> None:
        """track_rows [M,>=5] = x1,y1,x2,y2,id,...; keypoints [M,17,3] for the same M persons."""
        people = {}
        for row, kp in zip(np.asarray(track_rows), np.asarray(keypoints)):
            x1, y1, x2, y2 = (float(v) for v in row[:4])
            bbox = [x1, y1, x2 - x1, y2 - y1] if self.bbox_format == "xywh" else [x1, y1, x2, y2]
            people[int(row[4])] = [np.asarray(bbox, dtype=np.float32), np.asarray(kp, dtype=np.float32).reshape(-1, 3)]
        self.data[int(frame_num)] = people

    def save(self, path: str) -> None:
        with open(path, "wb") as f:
            pickle.dump(self.data, f)


def video_to_poselift(model, frames: Iterable[np.ndarray], out_path: Optional[str] = None, conf: float = 0.25,
                      batch: int = 64, first_frame: int = 0, gmc_device: Optional[int] = None, **predict_kw) -> Dict[int, Dict[int, list]]:
    """Run a pose model over the frames of ONE video and build its PoseLift dict (optionally pickled to ``out_path``).
    Detection runs in batches on the GPU; tracking is sequential on the host, in frame order.  ``gmc_device``: GPU that runs the
    tracker's motion compensation (``model.device`` for speed); the default keeps it on the host, whose track boxes are the ones
    the committed fixture holds bit for bit (the GPU routine agrees to 1e-3 px, not to the last bit)."""
    from .results import clip_boxes
    from .tracker import BYTETracker
    if getattr(model, "task", "pose") != "pose":
        raise ValueError("video_to_poselift needs a pose model (e.g. yolov8n-pose)")
    from concurrent.futures import ThreadPoolExecutor
    tracker = BYTETracker(gmc_device=gmc_device)
    w = PoseLiftWriter()
    n = first_frame

    def detect(buf):
        # a video's last batch is filled up to the next power of two (sweep.pad_bucket: few planned shapes, little wasted work)
        from .sweep import pad_bucket
        stack = np.stack(buf + [buf[-1]] * (pad_bucket(len(buf), batch) - len(buf)))
        return model.predict(stack, conf=min(conf, 0.1), **predict_kw)[:len(buf)]

    def track(fut, buf):
        nonlocal n
        warps = tracker.gmc.apply_batch(buf) if tracker.gmc.method is not None else [None] * len(buf)     # the batch's warps in one call
        for res, warp in zip(fut.result(), warps):
            rows = tracker.update(res.boxes.data.numpy(), warp=warp)   # every frame, empty ones too (frame_id / lost-track ageing)
            if len(rows):
                rows = clip_boxes(rows.copy(), res.orig_shape)  # Results.update clips the track boxes to the frame
                idx = rows[:, -1].astype(int)
                keep = rows[:, 5] >= conf                              # tracker sees conf >= 0.1, the file keeps conf >= conf
                raw = res.keypoints_raw if hasattr(res, "keypoints_raw") else res.keypoints.data.numpy()
                w.add_frame(n, rows[keep], np.asarray(raw)[idx][keep])
            else:
                w.add_frame(n, np.zeros((0, 8), np.float32), np.zeros((0, 17, 3), np.float32))
            n += 1

    def batches():
        buf = []
        for f in frames:
            buf.append(f)
            if len(buf) >= batch:
                yield buf
                buf = []
        if buf:
            yield buf

    # the detector pass of batch k + 1 runs on a worker thread (inside the engine's C call) while batch k is tracked here
    with ThreadPoolExecutor(max_workers=1, thread_name_prefix="detect") as pool:
        pending = None
        for buf in batches():
            fut = pool.submit(detect, buf)
            if pending is not None:
                track(*pending)
            pending = (fut, buf)
        if pending is not None:
            track(*pending)
    if out_path:
        w.save(out_path)
    return w.data
