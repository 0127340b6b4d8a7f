"""Whole-dataset preprocessing sweep, sharded over the GPUs of one node (BASELINE config 4).

The reference walks ``Anomaly_Train.txt`` sequentially in one process (``/root/reference/preprocess.py:15-53``) and
appends CSV rows per frame (``/root/reference/model.py:42-81``).  Here clips are dealt to the ranks BY FRAME COUNT (one
process per GPU, ``torch.distributed``): UCF-Crime clip lengths differ by more than 10x, so the clips are sorted by length
and each goes to the rank with the least work so far (longest-processing-time rule; every rank computes the same assignment
from the same counts).  Every rank pushes the frames of a clip through the engine in batches (the reference is batch 1), runs
the tracker per clip in frame order, and sends the clip's row block to rank 0 as soon as the clip is done (one non-blocking
point-to-point message per clip -- RCCL with the ``nccl`` backend, gloo in the CPU tests); rank 0 receives the blocks in clip
order, writes each to its CSV file at once and drops it, so it never holds more than its own share.  The files have the row
order the sequential loop produces.

Deliberate, documented differences from the reference loop:
* the tracker is per clip.  The reference keeps ONE tracker alive across all clips (``persist=True`` and never reset,
  ``preprocess.py:7``), so ids keep growing and a track can leak into the next clip; that makes the loop inherently
  sequential.  Here ids are made globally increasing again on rank 0 (each clip's ids are offset by the ids the
  previous clips used), so ``person`` stays unique across the file -- but the ``person`` column is NOT the sequential
  loop's (``preprocess_driver.run`` keeps the reference's single tracker and reproduces it).
* rows are appended once per clip, not once per frame (same bytes in the file).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .preprocess_driver import CAP_PROP_FRAME_COUNT, CAP_PROP_POS_FRAMES, VIDEOS_TO_PROCESS, open_capture
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


def pad_bucket(n: int, batch: int) -> int:
    """Frames a partial batch of n is filled up to: the next power of two, at most ``batch``.  The engine plans -- and on first
    sight times -- its launches per batch size; a sweep would otherwise meet every size from 1 to batch - 1, while padding every
    tail to the full batch makes a 65-frame clip cost 128 frames of detector work.  log2(batch) + 1 shapes at most."""
    b = 1
    while b < n:
        b *= 2
    return min(b, batch)


def assign_clips(frame_counts: Sequence[int], world: int) -> List[int]:
    """Longest-processing-time assignment: clips by descending frame count (ties: list order), each to the rank with the least
    frames so far (ties: lowest rank).  -> owner rank per clip.  Unknown lengths (count <= 0) are weighted with the mean of the
    known ones.  Greedy LPT is within 4/3 - 1/(3 world) of the optimal makespan; on UCF-Crime-like length mixes it lands within a
    few per cent of the mean load (tests/test_sweep.py)."""
    known = [c for c in frame_counts if c > 0]
    fill = max(1, int(round(sum(known) / len(known)))) if known else 1
    w = [c if c > 0 else fill for c in frame_counts]
    order = sorted(range(len(w)), key=lambda k: (-w[k], k))
    load = [0] * world
    owner = [0] * len(w)
    for k in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[k] = r
        load[r] += w[k]
    return owner


def track_rows_xywhn(tracks: np.ndarray, n: float, shape: Tuple[int, int]) -> np.ndarray:
    """rows [frame number, track id, xywhn] of one frame's track rows [M, >= 5] = x1,y1,x2,y2,id,...: what the reference's
    ``for box in boxes: float(box.id), float(box.xywhn[0][k])`` reads (model.py:56-64) after Results.update has clipped the
    Kalman-state boxes to the frame -- the same float32 operations (utils/ops.py:clip_boxes, xyxy2xywh, / (w, h, w, h)) on the
    whole array at once instead of a torch tensor per box."""
    h, w = shape[:2]
    t = np.array(tracks[:, :4], dtype=np.float32)
    t[:, [0, 2]] = t[:, [0, 2]].clip(0, w)
    t[:, [1, 3]] = t[:, [1, 3]].clip(0, h)
    out = np.empty((len(t), 6), np.float64)
    out[:, 0] = float(int(n))
    out[:, 1] = tracks[:, 4]
    out[:, 2] = ((t[:, 0] + t[:, 2]) / 2) / np.float32(w)
    out[:, 3] = ((t[:, 1] + t[:, 3]) / 2) / np.float32(h)
    out[:, 4] = (t[:, 2] - t[:, 0]) / np.float32(w)
    out[:, 5] = (t[:, 3] - t[:, 1]) / np.float32(h)
    return out


def process_clip(model, cap, batch: int = 64, conf: float = 0.1, classes=(0,), **predict_kw) -> np.ndarray:
    """All tracked boxes of one clip: array [rows, 6] = frame number (1-based), local track id, xywhn (centre x, centre y,
    w, h).  Detection is batched; the tracker sees the frames one by one, in order (model.py:38 semantics).

    Per batch three things run side by side (round 4): the tracker's motion compensation of all its frames as ONE batched GPU step
    (``gmc.apply_batch``, a thread of its own inside one C call), the detector pass of the SAME batch -- reading the frames that step has
    just uploaded (``gmc.batch_device_frames`` -> ``mi355_yolo_infer_device``): one staging copy and one upload per batch instead of two of
    each, no ``np.stack``, no Results objects -- and, on this thread, the association of the PREVIOUS batch frame by frame (host C++).
    A clip's last, partial batch is padded for the detector only (power-of-two bucket) and takes the host path."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from .engine import YOLO
    from .tracker import BYTETracker
    tracker = BYTETracker(gmc_device=getattr(model, "device", None))     # motion compensation on the engine's GPU (csrc/gmc_kernels.hip)
    blocks: List[np.ndarray] = []
    gmc_on = tracker.gmc.method is not None
    share = gmc_on and tracker.gmc.device is not None and hasattr(model, "detect_rows") and os.environ.get("MI355_SWEEP_SHARED_FRAMES", "1") != "0"     # A/B and tests only
    kw = dict(conf=conf, classes=list(classes), **predict_kw)

    def rows_of(frames):
        if hasattr(model, "detect_rows"):
            return model.detect_rows(frames, **kw)
        res = model.predict(frames, **kw)                      # any object with the Ultralytics call surface
        return [r.boxes.data.numpy() for r in res], (res[0].orig_shape if res else tuple(frames.shape[1:3]))

    log_t = os.environ.get("MI355_SWEEP_LOG") == "1"             # per-batch stage times on stderr (diagnostics)

    def timed(name, fn, *a):
        if not log_t:
            return fn(*a)
        import sys, time
        t0 = time.perf_counter()
        r = fn(*a)
        print(f"[sweep] {name} {1e3 * (time.perf_counter() - t0):.2f} ms (start {1e3 * (t0 % 10):.1f})", file=sys.stderr)
        return r

    def detect(buf, after_seq):
        n = len(buf)
        if share and pad_bucket(n, batch) == n:
            dev = tracker.gmc.batch_device_frames(after_seq)
            if dev is not None and dev[1] == n and dev[2:] == tuple(buf[0].shape[:2]):
                return model.detect_rows(YOLO._DeviceFrames(dev[0], n, dev[2], dev[3]), **kw)
        # host path: a clip's last batch is filled up to the next power of two with copies of its last frame (their results are dropped)
        stack = np.stack(buf + [buf[-1]] * (pad_bucket(n, batch) - n))
        dets, shape = rows_of(stack)
        return dets[:n], shape

    def associate(nums, fut_d, fut_g):
        dets, shape = fut_d.result()
        warps = fut_g.result() if fut_g is not None else [None] * len(nums)
        for n, det, warp in zip(nums, dets, warps):
            tracks = tracker.update(det, warp=warp)            # every frame, empty ones too (frame_id / lost-track ageing)
            if len(tracks):                                    # `if not boxes.is_track: return` otherwise (model.py:45)
                blocks.append(track_rows_xywhn(tracks, n, shape))

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

    with ThreadPoolExecutor(max_workers=1, thread_name_prefix="gmc") as pool_g, ThreadPoolExecutor(max_workers=1, thread_name_prefix="detect") as pool_d:
        pending = None
        for nums, buf in batches():
            # the detector pass of batch k reads batch k's frames out of the motion-compensation step's device buffer: batch k - 1's detector
            # pass is over (its future was collected below before this point of the previous iteration's successor) ...
            if pending is not None:
                pending[1].result()
            seq = tracker.gmc.batch_seq() if share else 0
            fut_g = pool_g.submit(timed, "gmc", tracker.gmc.apply_batch, buf) if gmc_on else None
            fut_d = pool_d.submit(timed, "detect", detect, buf, seq)
            if pending is not None:
                timed("associate", associate, *pending)       # ... and batch k - 1 is associated while the GPU works on batch k
            pending = (nums, fut_d, fut_g)
        if pending is not None:
            associate(*pending)
    cap.release()
    return np.concatenate(blocks) if blocks else np.zeros((0, 6), np.float64)


def _comm_device():
    import torch
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def sweep(model, list_path: str, dataset_root: str, out_dir: str = "dataset", batch: int = 64,
          capture: Callable = open_capture, videos_to_process: Optional[List[str]] = None, log: Callable = print) -> int:
    """Run the sweep on this rank's share of the clips; rank 0 writes the CSVs.  Returns the number of rows written
    (on rank 0; 0 elsewhere).  Works without an initialised process group (single GPU)."""
    import torch
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    clips = list_clips(list_path, videos_to_process)
    # frame counts -> owners (every rank opens every clip's header: CAP_PROP_FRAME_COUNT, or the length of a frame dump)
    counts = []
    for (_, _, _, rel) in clips:
        cap = capture(dataset_root + rel)
        counts.append(int(cap.get(CAP_PROP_FRAME_COUNT)) if cap.isOpened() else -1)
        if cap.isOpened():
            cap.release()
    openable = [c >= 0 for c in counts]
    owner = assign_clips([c for c in counts], world) if world > 1 else [0] * len(clips)

    state = {"written": 0, "id_offset": 0}

    def write_block(k: int, arr: Optional[np.ndarray]) -> None:
        if arr is None or not len(arr):
            return
        i, label, name, _ = clips[k]
        is_anomaly = label in ANOMALIES
        data = [BBox(clip=i, name=name, frame=int(r[0]), person=float(r[1] + state["id_offset"]), left=float(r[2]), top=float(r[3]),
                     width=float(r[4]), height=float(r[5]), is_anomaly=is_anomaly, anomaly=label) for r in arr]
        write_rows(os.path.join(out_dir, "ucf-crime_dataset.csv" if is_anomaly else "ucf-crime_dataset-normal.csv"), data)
        state["id_offset"] += int(arr[:, 1].max())
        state["written"] += len(data)

    def run_clip(k: int) -> Optional[np.ndarray]:
        if not openable[k]:
            log(f"Failed to load video: {clips[k][3]}")
            return None
        return process_clip(model, capture(dataset_root + clips[k][3]), batch=batch)

    if not distributed or world == 1:
        for k in range(len(clips)):                            # clip order == the reference's loop order
            write_block(k, run_clip(k))
        return state["written"]

    dev = _comm_device()
    if rank != 0:
        # C3, per clip: [rows] header, then the block; non-blocking, so the next clip starts at once.  Clips in ascending list order:
        # rank 0 merges the ranks' streams by clip number, and messages between two ranks arrive in the order they were sent
        pending = []
        for k in range(len(clips)):
            if owner[k] != rank:
                continue
            arr = run_clip(k)
            rows = -1 if arr is None else len(arr)
            head = torch.tensor([rows], dtype=torch.int64, device=dev)
            pending.append((dist.isend(head, dst=0), head))
            if rows > 0:
                body = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
                pending.append((dist.isend(body, dst=0), body))
        for req, _keep in pending:
            req.wait()
        return 0
    mine = {k: run_clip(k) for k in range(len(clips)) if owner[k] == 0}
    for k in range(len(clips)):
        if owner[k] == 0:
            write_block(k, mine.pop(k))
            continue
        head = torch.zeros(1, dtype=torch.int64, device=dev)
        dist.recv(head, src=owner[k])
        rows = int(head.item())
        if rows > 0:
            body = torch.empty((rows, 6), dtype=torch.float64, device=dev)
            dist.recv(body, src=owner[k])
            write_block(k, body.cpu().numpy())
    return state["written"]
