"""Host-side multi-object tracker behind ``YOLO.track``.

The reference calls ``model.track(frame, persist=True, classes=[0])`` (``/root/reference/model.py:38``), which in
ultralytics==8.3.225 runs predict at conf 0.1 and then its default ``botsort.yaml`` tracker: BoT-SORT = ByteTrack's
two-stage association (Zhang et al., "ByteTrack", ECCV 2022) on IoU cost with score fusion, a constant-velocity Kalman
filter over (cx, cy, w, h) (Aharon et al., "BoT-SORT", 2022), a 30-frame lost-track buffer, one-frame confirmation of
tracks born after the first frame, and output rows ``[x1,y1,x2,y2,id,score,cls,idx]`` whose box is the filter state.

This module implements that published algorithm from its definitions, organised around a table of plain records and a
structure-exploiting Kalman filter (the transition is ``x += v``, the measurement is the first four state components, so
no motion / projection matrices are ever formed).  Global motion compensation (``gmc_method: sparseOptFlow``, the
botsort.yaml default) is ``gmc.py``: when ``update`` is handed the frame, the background's partial-affine motion since the
previous frame is estimated (Shi-Tomasi corners, pyramidal Lucas-Kanade, RANSAC) and applied to the predicted Kalman
state of every pooled and every unconfirmed track before association, as ``BOTSORT`` does (``STrack.multi_gmc``);
``gmc_method=None`` keeps the identity.  Not implemented: ReID (``with_reid: False`` is the default).  Assignment is SciPy's
Hungarian solver, as in Ultralytics' ``linear_assignment(use_lap=False)`` path.  Tracking is sequential per video and stays
on the host.

Behaviour is pinned by hand-derived known answers (``tests/test_tracker_known_answers.py``).  PARITY UNPINNED against a
real Ultralytics run (no ultralytics / lap / cv2 here).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
from scipy.optimize import linear_sum_assignment

# botsort.yaml defaults
TRACK_HIGH_THRESH = 0.25      # first association: detections with score >= this
TRACK_LOW_THRESH = 0.1        # second association: this < score < high
NEW_TRACK_THRESH = 0.25       # a leftover detection starts a track only at or above this
TRACK_BUFFER = 30             # frames a lost track is kept (at 30 fps)
MATCH_THRESH = 0.8            # first association accepts fused cost <= this
FUSE_SCORE = True

POS_STD, VEL_STD = 1.0 / 20.0, 1.0 / 160.0     # process / measurement noise relative to the box size

FRESH, TRACKED, LOST, RETIRED = 0, 1, 2, 3


# ----------------------------------------------------------------------------------------------- Kalman filter
# State s = (cx, cy, w, h, vcx, vcy, vw, vh), covariance P = [[A, B], [B', C]] in 4x4 blocks.
def _noise_diag(w: float, h: float, pos: float, vel: float) -> np.ndarray:
    return np.square(np.array([pos * w, pos * h, pos * w, pos * h, vel * w, vel * h, vel * w, vel * h]))


class KalmanFilterXYWH:
    """Constant-velocity filter on (cx, cy, w, h).  ``initiate`` / ``predict`` / ``update`` take and return
    ``(mean[8], covariance[8, 8])``."""

    @staticmethod
    def initiate(z: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        z = np.asarray(z, dtype=np.float64)
        mean = np.concatenate([z, np.zeros(4)])
        return mean, np.diag(_noise_diag(z[2], z[3], 2 * POS_STD, 10 * VEL_STD))

    @staticmethod
    def predict(mean: np.ndarray, cov: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """x <- x + v;  P <- F P F' + Q with F = [[I, I], [0, I]]: A <- A + B + B' + C, B <- B + C, C <- C."""
        q = _noise_diag(mean[2], mean[3], POS_STD, VEL_STD)
        a, b, c = cov[:4, :4], cov[:4, 4:], cov[4:, 4:]
        new = np.empty((8, 8))
        new[:4, :4] = a + b + b.T + c
        new[:4, 4:] = b + c
        new[4:, :4] = new[:4, 4:].T
        new[4:, 4:] = c
        new[np.arange(8), np.arange(8)] += q
        out = mean.copy()
        out[:4] += mean[4:]
        return out, new

    @staticmethod
    def update(mean: np.ndarray, cov: np.ndarray, z: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """Measurement = the first four components, R = diag((w/20)^2, (h/20)^2, ...).  S = A + R, K = [A; B'] S^-1."""
        r = _noise_diag(mean[2], mean[3], POS_STD, VEL_STD)[:4]
        s = cov[:4, :4] + np.diag(r)
        gain = np.linalg.solve(s, cov[:4, :]).T                  # S symmetric: (P H' S^-1) = (S^-1 H P)'
        innovation = np.asarray(z, dtype=np.float64) - mean[:4]
        return mean + gain @ innovation, cov - gain @ s @ gain.T


# ----------------------------------------------------------------------------------------------- geometry / matching
def _iou_cost(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """1 - IoU between xyxy boxes a [n,4] and b [m,4] (fp32, eps 1e-7 as utils/metrics.py:bbox_ioa(iou=True))."""
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)), dtype=np.float32)
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    iw = (np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0])).clip(0)
    ih = (np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1])).clip(0)
    inter = iw * ih
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return 1 - inter / (area_a[:, None] + area_b[None, :] - inter + 1e-7)


def _assign(cost: np.ndarray, limit: float) -> Tuple[List[Tuple[int, int]], List[int], List[int]]:
    """Minimum-cost assignment; pairs whose cost exceeds ``limit`` are dropped.  -> (pairs, free rows, free columns)"""
    rows, cols = range(cost.shape[0]), range(cost.shape[1])
    if cost.size == 0:
        return [], list(rows), list(cols)
    ri, ci = linear_sum_assignment(cost)
    pairs = [(int(r), int(c)) for r, c in zip(ri, ci) if cost[r, c] <= limit]
    used_r, used_c = {p[0] for p in pairs}, {p[1] for p in pairs}
    return pairs, [r for r in rows if r not in used_r], [c for c in cols if c not in used_c]


@dataclass
class _Det:
    """One detection of the current frame in the tracker's terms."""
    xywh: np.ndarray          # centre x, centre y, w, h (float64)
    score: float
    cls: float
    idx: float                # row of the frame's detection array

    @property
    def xyxy(self) -> np.ndarray:
        x, y, w, h = self.xywh
        return np.array([x - w / 2, y - h / 2, x + w / 2, y + h / 2])


@dataclass
class Track:
    track_id: int
    mean: np.ndarray
    cov: np.ndarray
    score: float
    cls: float
    idx: float
    state: int
    confirmed: bool           # reported only once confirmed (born on frame 1, or matched on the frame after birth)
    born: int                 # frame of birth
    seen: int                 # last frame with a matched detection

    @property
    def xyxy(self) -> np.ndarray:
        x, y, w, h = self.mean[:4]
        return np.array([x - w / 2, y - h / 2, x + w / 2, y + h / 2])

    def absorb(self, d: _Det, frame: int) -> None:
        self.mean, self.cov = KalmanFilterXYWH.update(self.mean, self.cov, d.xywh)
        self.score, self.cls, self.idx = d.score, d.cls, d.idx
        self.state, self.confirmed, self.seen = TRACKED, True, frame


class BYTETracker:
    """``update(det [N,6] = x1,y1,x2,y2,conf,cls) -> [M,8] = x1,y1,x2,y2,id,score,cls,idx`` (idx = row of ``det``), one call
    per frame, EVERY frame (an empty frame still ages the lost tracks)."""

    def __init__(self, frame_rate: int = 30, gmc_method: Optional[str] = "sparseOptFlow", gmc_device: Optional[int] = None):
        from .gmc import GMC
        self.gmc = GMC(gmc_method, device=gmc_device)   # BOTSORT.__init__: GMC(method=args.gmc_method); None = identity
        self.frame_id = 0
        self.max_time_lost = int(frame_rate / 30.0 * TRACK_BUFFER)
        self._live: List[Track] = []          # tracked (confirmed or awaiting confirmation), in report order
        self._lost: List[Track] = []
        self._retired_ids: set = set()        # ids retired on EARLIER frames (see the bookkeeping note in update)
        # ids belong to the tracker instance (1, 2, ... in birth order): a second tracker created while this one is alive
        # (sweep / PoseLift bridge next to model.track(persist=True)) cannot disturb them
        self._ids_issued = 0

    @property
    def tracked_stracks(self) -> List[Track]:
        return self._live

    @property
    def lost_stracks(self) -> List[Track]:
        return self._lost

    def _detections(self, det: np.ndarray, keep: np.ndarray) -> List[_Det]:
        out = []
        for i in np.nonzero(keep)[0]:
            x1, y1, x2, y2 = (float(v) for v in np.asarray(det[i, :4], dtype=np.float32))
            xywh = np.asarray(np.float32([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1]), dtype=np.float64)
            out.append(_Det(xywh, float(det[i, 4]), float(det[i, 5]), float(i)))
        return out

    @staticmethod
    def _cost(tracks: Sequence[Track], dets: Sequence[_Det], fuse: bool) -> np.ndarray:
        cost = _iou_cost([t.xyxy for t in tracks], [d.xyxy for d in dets])
        if fuse and cost.size:
            cost = 1 - (1 - cost) * np.array([d.score for d in dets])[None, :]
        return cost

    def update(self, det: np.ndarray, img: Optional[np.ndarray] = None, next_img: Optional[np.ndarray] = None) -> np.ndarray:
        """``img``: the frame the detections come from (BGR uint8, as ``tracker.update(det, im0)`` receives it in
        trackers/track.py); without it no motion compensation takes place (byte_tracker.py: ``if ... img is not None``).
        ``next_img``: the frame the NEXT call will bring, when the caller already holds it (a batched sweep): its motion-compensation
        step is enqueued on the GPU as soon as this frame's has been collected, and runs beside this call's association."""
        self.frame_id += 1
        frame = self.frame_id
        det = np.asarray(det, dtype=np.float32).reshape(-1, 6)
        scores = det[:, 4]
        strong = self._detections(det, scores >= TRACK_HIGH_THRESH)
        weak = self._detections(det, (scores > TRACK_LOW_THRESH) & (scores < TRACK_HIGH_THRESH))

        confirmed = [t for t in self._live if t.confirmed]
        tentative = [t for t in self._live if not t.confirmed]
        # candidate pool: confirmed tracks, then lost ones not already in it
        pool = list(confirmed) + [t for t in self._lost if all(t.track_id != c.track_id for c in confirmed)]
        for t in pool:                                          # a track that is not currently tracked stops changing size
            m = t.mean.copy()
            if t.state != TRACKED:
                m[6] = m[7] = 0.0
            t.mean, t.cov = KalmanFilterXYWH.predict(m, t.cov)
        if img is not None and self.gmc.method is not None:
            # camera motion since the previous frame, applied to the predicted states (pool) and the unconfirmed tracks
            from .gmc import warp_kalman
            try:
                warp = self.gmc.apply(img)
            except (np.linalg.LinAlgError, ValueError, FloatingPointError, ZeroDivisionError):
                # byte_tracker.py bypasses errors of the gmc module the same way (degenerate point sets); a RuntimeError of the
                # device path (HIP error, library without the kernels) is NOT one of them and surfaces
                warp = np.eye(2, 3)
            if next_img is not None:
                self.gmc.begin(next_img)                        # device path only; a no-op on the host
            if not np.array_equal(warp, np.eye(2, 3)):
                R8 = np.kron(np.eye(4), warp[:2, :2])
                for t in pool + tentative:
                    t.mean, t.cov = warp_kalman(t.mean, t.cov, warp, R8)

        touched: List[Track] = []        # matched this frame and previously tracked ("activated")
        revived: List[Track] = []        # matched this frame and previously lost ("refound")
        newly_lost: List[Track] = []
        retired_now: List[Track] = []

        def take(t: Track, d: _Det) -> None:
            was_tracked = t.state == TRACKED
            t.absorb(d, frame)
            (touched if was_tracked else revived).append(t)

        # 1. strong detections against the pool (IoU cost fused with the detection score)
        pairs, free_t, free_d = _assign(self._cost(pool, strong, FUSE_SCORE), MATCH_THRESH)
        for ti, di in pairs:
            take(pool[ti], strong[di])
        # 2. weak detections against the still-unmatched TRACKED tracks (plain IoU cost, limit 0.5)
        rest = [pool[i] for i in free_t if pool[i].state == TRACKED]
        pairs2, free_rest, _ = _assign(self._cost(rest, weak, False), 0.5)
        for ti, di in pairs2:
            take(rest[ti], weak[di])
        for i in free_rest:
            if rest[i].state != LOST:
                rest[i].state = LOST
                newly_lost.append(rest[i])
        # 3. leftover strong detections against tracks awaiting confirmation (limit 0.7); unmatched ones are dropped
        leftover = [strong[i] for i in free_d]
        pairs3, free_tent, free_left = _assign(self._cost(tentative, leftover, FUSE_SCORE), 0.7)
        for ti, di in pairs3:
            tentative[ti].absorb(leftover[di], frame)
            touched.append(tentative[ti])
        for i in free_tent:
            tentative[i].state = RETIRED
            retired_now.append(tentative[i])
        # 4. births
        for i in free_left:
            d = leftover[i]
            if d.score < NEW_TRACK_THRESH:
                continue
            self._ids_issued += 1
            mean, cov = KalmanFilterXYWH.initiate(d.xywh)
            t = Track(self._ids_issued, mean, cov, d.score, d.cls, d.idx, TRACKED, confirmed=(frame == 1), born=frame, seen=frame)
            touched.append(t)
        # 5. lost tracks past the buffer
        for t in self._lost:
            if frame - t.seen > self.max_time_lost:
                t.state = RETIRED
                retired_now.append(t)

        # ---- bookkeeping, in ByteTrack's order: the lists are rebuilt BEFORE this frame's retirements are recorded, so a
        # track retired now leaves the candidate pool one frame later (pinned by the known-answer tests)
        live = [t for t in self._live if t.state == TRACKED]
        for group in (touched, revived):
            have = {t.track_id for t in live}
            live += [t for t in group if t.track_id not in have and not have.add(t.track_id)]
        live_ids = {t.track_id for t in live}
        lost = [t for t in self._lost if t.track_id not in live_ids] + newly_lost
        lost = [t for t in lost if t.track_id not in self._retired_ids]
        live, lost = self._drop_duplicates(live, lost)
        self._retired_ids.update(t.track_id for t in retired_now)
        self._live, self._lost = live, lost
        rows = [[*t.xyxy.tolist(), t.track_id, t.score, t.cls, t.idx] for t in self._live if t.confirmed]
        return np.asarray(rows, dtype=np.float32).reshape(-1, 8)

    @staticmethod
    def _drop_duplicates(live: List[Track], lost: List[Track]) -> Tuple[List[Track], List[Track]]:
        """a tracked and a lost track on (nearly) the same box (IoU > 0.85): the one with the longer history survives"""
        cost = _iou_cost([t.xyxy for t in live], [t.xyxy for t in lost])
        kill_live, kill_lost = set(), set()
        for p, q in zip(*np.where(cost < 0.15)):
            if live[p].seen - live[p].born > lost[q].seen - lost[q].born:
                kill_lost.add(int(q))
            else:
                kill_live.add(int(p))
        return [t for i, t in enumerate(live) if i not in kill_live], [t for i, t in enumerate(lost) if i not in kill_lost]
