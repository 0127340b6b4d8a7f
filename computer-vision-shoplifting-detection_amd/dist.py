"""Multi-GPU sharding of the frame stream: one process per GPU over ``torch.distributed``.

The reference is a single sequential loop (``/root/reference/preprocess.py:19-47``); detection is
independent per frame, so frames shard embarrassingly.  Three call sites only:

  C1 ``broadcast_weights``   rank 0's ``.mi355w`` image -> every rank (once, at start-up)
  C2 ``all_gather`` of per-frame row counts          \\ once per batch, inside ``gather_rows``
  C3 gather of the compact row block to rank 0       /

With the ``nccl`` backend (= RCCL on ROCm, over xGMI) the tensors live on the rank's GPU; with ``gloo``
(CPU tests) on the host.  Payloads are KB..MB, so the design minimises the *number* of collectives
(two per batch), not bytes.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _device() -> torch.device:
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of frames for ``rank``: [r*F/R, (r+1)*F/R) -- contiguous keeps CSV order trivial."""
    return (rank * total) // world, ((rank + 1) * total) // world


def broadcast_weights(blob: Optional[bytes], src: int = 0) -> bytes:
    """C1: every rank returns the bytes of rank ``src``'s weight image."""
    if not is_dist():
        assert blob is not None
        return blob
    dev = _device()
    n = torch.tensor([len(blob) if dist.get_rank() == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src)
    if dist.get_rank() == src:
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    else:
        t = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src)
    return t.cpu().numpy().tobytes()


def gather_rows(rows: np.ndarray, counts: np.ndarray, dst: int = 0, ncols: Optional[int] = None):
    """C2 + C3.  ``rows`` [B, cap, W] float32 (first counts[i] rows of frame i valid), ``counts`` [B] int32.
    ``ncols``: ship only the first ncols words of a row (a detect row needs 7 of the 58: box, conf, cls, anchor; a pose
    row 7 + 51) -- rank 0 receives world x this payload every step, so it is kept minimal.
    On ``dst``: (list over ranks of compact row blocks [sum(counts_r), W], list of counts arrays), in rank
    order == global frame order for contiguous shards.  Other ranks get (None, None)."""
    if ncols is not None:
        rows = rows[..., :ncols]
    counts = np.asarray(counts)
    if len(counts):           # rows[i, :counts[i]] for every frame, in frame order, without a Python loop over frames
        compact = rows[np.arange(rows.shape[1])[None, :] < counts[:, None]]
    else:
        compact = rows[:0, 0]
    if not is_dist():
        return [compact], [np.asarray(counts)]
    dev = _device()
    world, rank = dist.get_world_size(), dist.get_rank()
    c = torch.as_tensor(np.asarray(counts, dtype=np.int32), device=dev)
    all_c = [torch.empty_like(c) for _ in range(world)]
    dist.all_gather(all_c, c)                                        # C2
    totals = [int(t) for t in torch.stack(all_c).sum(dim=1).cpu().tolist()]
    width = rows.shape[-1]
    m = max(max(totals), 1)
    pad = torch.zeros((m, width), dtype=torch.float32, device=dev)
    if len(compact):
        pad[:len(compact)] = torch.as_tensor(compact, device=dev)
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)                                  # C3
    if rank != dst:
        return None, None
    return ([b[:t].cpu().numpy() for b, t in zip(bufs, totals)], [x.cpu().numpy() for x in all_c])


class DeviceRowGather:
    """C2 + C3 for rows that STAY in HBM (``YOLO.infer_async``): the collectives read the engine's device buffers directly
    and run one step behind the engine, so step k's gather overlaps step k+1's kernels.

        g = DeviceRowGather(model, n_frames)          # double-buffered outputs
        for k in range(steps):
            done = g.submit(frames)                   # enqueue step k; returns step k-1's gathered rows (rank 0) or None
        last = g.flush()

    With ``nccl`` (RCCL over xGMI) nothing touches the host except rank 0's final copy of the gathered block; with ``gloo``
    (CPU tests) the device buffers are staged through the host -- same call sequence, same result."""

    def __init__(self, model, n_frames: int, max_det: int = 300, ncols: Optional[int] = None, dst: int = 0, **infer_kw):
        self.model, self.n, self.max_det, self.dst, self.kw = model, n_frames, max_det, dst, infer_kw
        self.ncols = ncols
        self.bufs = [model.new_device_rows(n_frames, max_det) for _ in range(2)]
        self.pending = None
        self.k = 0

    def _collect(self, item):
        out, ready = item
        rows, counts, total = out
        torch.cuda.current_stream().wait_event(ready)                    # the engine has passed THIS step (not the next one)
        world = dist.get_world_size() if is_dist() else 1
        nccl = is_dist() and dist.get_backend() == "nccl"
        if not is_dist():
            t = int(total.item())
            r = rows[:t] if self.ncols is None else rows[:t, :self.ncols]
            return [r.cpu().numpy()], [counts.cpu().numpy()]
        c = counts if nccl else counts.cpu()
        all_c = [torch.empty_like(c) for _ in range(world)]
        dist.all_gather(all_c, c)                                        # C2
        totals = [int(t) for t in torch.stack(all_c).sum(dim=1).cpu().tolist()]   # ONE device->host sync per step, not one per rank
        m = max(max(totals), 1)
        block = rows[:m] if self.ncols is None else rows[:m, :self.ncols]
        block = block.contiguous() if nccl else block.cpu().contiguous()
        recv = [torch.empty_like(block) for _ in range(world)] if dist.get_rank() == self.dst else None
        dist.gather(block, recv, dst=self.dst)                           # C3
        if dist.get_rank() != self.dst:
            return None, None
        return [b[:t].cpu().numpy() for b, t in zip(recv, totals)], [x.cpu().numpy() for x in all_c]

    def submit(self, frames):
        out = self.bufs[self.k & 1]
        self.k += 1
        self.model.infer_async(frames, out, max_det=self.max_det, **self.kw)
        ready = torch.cuda.Event()
        ready.record(self.model.stream)
        prev, self.pending = self.pending, (out, ready)
        return self._collect(prev) if prev is not None else None

    def flush(self):
        prev, self.pending = self.pending, None
        return self._collect(prev) if prev is not None else None
