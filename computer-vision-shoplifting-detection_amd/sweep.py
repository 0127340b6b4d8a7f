"""Whole-dataset preprocessing sweep, sharded over the GPUs of one node (BASELINE config 4).

The reference walks ``Anomaly_Train.txt`` sequentially in one process (``/root/reference/preprocess.py:15-53``) and
appends CSV rows per frame (``/root/reference/model.py:42-81``).  Here clips are dealt round-robin to the ranks (one
process per GPU, ``torch.distributed``), every rank pushes the frames of a clip through the engine in batches (the
reference is batch 1), runs the host-side tracker per clip in frame order, and rank 0 gathers the row blocks (RCCL
``gather_object`` with the ``nccl`` backend, gloo in the CPU tests) and writes the two CSV files in clip order -- the
same row order the sequential loop produces.

Deliberate, documented differences from the reference loop:
* the tracker is per clip.  The reference keeps ONE tracker alive across all clips (``persist=True`` and never reset,
  ``preprocess.py:7``), so ids keep growing and a track can leak into the next clip; that makes the loop inherently
  sequential.  Here ids are made globally increasing again on rank 0 (each clip's ids are offset by the ids the
  previous clips used), so ``person`` stays unique across the file.
* rows are appended once per clip, not once per frame (same bytes in the file).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import numpy as np

from .preprocess_driver import CAP_PROP_POS_FRAMES, VIDEOS_TO_PROCESS, open_capture
from .tracker_csv import ANOMALIES, BBox, write_rows


def list_clips(list_path: str, videos_to_process: Optional[List[str]] = None) -> List[Tuple[int, str, str, str]]:
    """-> [(clip number i as the reference counts it (every list line counts), label, name, relative path)]"""
    videos_to_process = VIDEOS_TO_PROCESS if videos_to_process is None else videos_to_process
    with open(list_path, "r") as f:
        videos = f.read().split("\n")
    out = []
    for i, video in enumerate(videos, start=1):
        parts = video.split("/")
        if len(parts) < 2 or parts[0] not in videos_to_process:
            continue
        out.append((i, parts[0], parts[1], video))
    return out


def process_clip(model, cap, batch: int = 64, conf: float = 0.1, classes=(0,), **predict_kw) -> np.ndarray:
    """All tracked boxes of one clip: array [rows, 6] = frame number (1-based), local track id, xywhn (centre x, centre y,
    w, h).  Detection is batched; the tracker sees the frames one by one, in order (model.py:38 semantics).

    Three things run side by side: the detector pass of batch k + 1 (a worker thread inside the engine's C call), the tracker's
    association of batch k on this thread, and -- on a GPU stream of its own -- the motion-compensation step of the NEXT frame,
    enqueued as soon as the current frame's has been collected."""
    from concurrent.futures import ThreadPoolExecutor
    from .results import Boxes, clip_boxes
    from .tracker import BYTETracker
    import torch
    tracker = BYTETracker(gmc_device=getattr(model, "device", None))     # motion compensation on the engine's GPU (csrc/gmc_kernels.hip)
    rows: List[List[float]] = []

    def detect(buf):
        # a clip's last batch is filled up with copies of its last frame (their results are dropped): the engine plans -- and on
        # first sight times -- its launches per batch size, and a sweep would otherwise meet every size from 1 to batch - 1
        stack = np.stack(buf + [buf[-1]] * (batch - len(buf)))
        return model.predict(stack, conf=conf, classes=list(classes), **predict_kw)[:len(buf)]

    def track(fut, nums, buf):
        tracker.gmc.begin(buf[0])                              # no-op when the previous batch's last frame already enqueued it
        results = fut.result()
        for j, (n, frame, res) in enumerate(zip(nums, buf, results)):
            # every frame, empty ones too (frame_id / lost-track ageing); frame j + 1's step is enqueued while frame j is associated
            tracks = tracker.update(res.boxes.data.numpy(), frame, next_img=buf[j + 1] if j + 1 < len(buf) else None)
            if len(tracks):                                    # `if not boxes.is_track: return` otherwise (model.py:45)
                b = Boxes(clip_boxes(torch.as_tensor(tracks[:, :-1], dtype=torch.float32), res.orig_shape), res.orig_shape)
                for box in b:
                    x = box.xywhn[0]
                    rows.append([float(int(n)), float(box.id), float(x[0]), float(x[1]), float(x[2]), float(x[3])])

    def batches():
        buf, nums = [], []
        while True:
            success, frame = cap.read()
            n = cap.get(CAP_PROP_POS_FRAMES)
            if not success:
                break
            buf.append(frame)
            nums.append(n)
            if len(buf) >= batch:
                yield nums, buf
                buf, nums = [], []
        if buf:
            yield nums, buf

    with ThreadPoolExecutor(max_workers=1, thread_name_prefix="detect") as pool:
        pending = None
        for nums, buf in batches():
            fut = pool.submit(detect, buf)                     # batch k + 1 on the detector ...
            if pending is not None:
                track(*pending)                                # ... while batch k is tracked
            pending = (fut, nums, buf)
        if pending is not None:
            track(*pending)
    cap.release()
    return np.asarray(rows, dtype=np.float64).reshape(-1, 6)


def sweep(model, list_path: str, dataset_root: str, out_dir: str = "dataset", batch: int = 64,
          capture: Callable = open_capture, videos_to_process: Optional[List[str]] = None, log: Callable = print) -> int:
    """Run the sweep on this rank's share of the clips; rank 0 writes the CSVs.  Returns the number of rows written
    (on rank 0; 0 elsewhere).  Works without an initialised process group (single GPU)."""
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    clips = list_clips(list_path, videos_to_process)
    mine = []
    for k, (i, label, name, rel) in enumerate(clips):
        if k % world != rank:
            continue
        cap = capture(dataset_root + rel)
        if not cap.isOpened():
            log(f"Failed to load video: {rel}")
            mine.append((k, None))
            continue
        mine.append((k, process_clip(model, cap, batch=batch)))
    if distributed:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)              # C3: row blocks to rank 0
        if rank != 0:
            return 0
        blocks = dict(kv for part in gathered for kv in part)
    else:
        blocks = dict(mine)
    written, id_offset = 0, 0
    for k, (i, label, name, rel) in enumerate(clips):          # clip order == the reference's loop order
        arr = blocks.get(k)
        if arr is None or not len(arr):
            continue
        is_anomaly = label in ANOMALIES
        data = [BBox(clip=i, name=name, frame=int(r[0]), person=float(r[1] + id_offset), left=float(r[2]), top=float(r[3]),
                     width=float(r[4]), height=float(r[5]), is_anomaly=is_anomaly, anomaly=label) for r in arr]
        write_rows(os.path.join(out_dir, "ucf-crime_dataset.csv" if is_anomaly else "ucf-crime_dataset-normal.csv"), data)
        id_offset += int(arr[:, 1].max())
        written += len(data)
    return written
