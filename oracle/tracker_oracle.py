"""CPU restatement (numpy) of the multi-object tracker behind ``YOLO.track`` -- TEST INFRASTRUCTURE, not product code: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product (cvsd_amd/tracker.py) binds host C++
(csrc/tracker_host.cpp) and never imports it.  This is the round-2/3 tracker, kept as the checker of that C++ core.

What it restates: the reference calls ``model.track(frame, persist=True, classes=[0])`` (``/root/reference/model.py:38``), which in
ultralytics==8.3.225 (un-vendored: requirements.txt:121) runs predict at conf 0.1 and then its default ``botsort.yaml`` tracker
(``trackers/bot_sort.py``, ``trackers/byte_tracker.py``): BoT-SORT = ByteTrack's two-stage association (Zhang et al., "ByteTrack",
ECCV 2022) on IoU cost with score fusion (``trackers/utils/matching.py``), a constant-velocity Kalman filter over (cx, cy, w, h)
(``trackers/utils/kalman_filter.py:KalmanFilterXYWH``; Aharon et al., "BoT-SORT", 2022), a 30-frame lost-track buffer, one-frame
confirmation of tracks born after the first frame, and output rows ``[x1,y1,x2,y2,id,score,cls,idx]`` whose box is the filter state.
Assignment is ``lap.lapjv(cost, extend_cost=True, cost_limit=thresh)`` (``matching.py:linear_assignment``, ``use_lap=True``;
lap==0.5.12, requirements.txt:41, absent here): :func:`lapjv` below states that solver -- Jonker & Volgenant's dense algorithm
(Computing 38, 1987) on the (rows + cols)^2 extension whose padding costs ``cost_limit / 2`` per cell -- in plain Python loops.
Global motion compensation (``gmc_method: sparseOptFlow``) is ``oracle/gmc_oracle.py``; when ``update`` is handed the frame, the
background's partial-affine motion since the previous frame is applied to the predicted Kalman state of every pooled and every
unconfirmed track before association (``STrack.multi_gmc``).  Not implemented: ReID (``with_reid: False`` is the default).

PARITY UNPINNED against a real Ultralytics / lap run (neither is installable here and the reference holds no fixtures for this
path); pinned by hand-derived known answers (``tests/test_tracker_known_answers.py`` runs them against the product AND this
module) and, for the solver, by optimality against ``scipy.optimize.linear_sum_assignment`` on the extended matrix.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

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


_LARGE = 1000000.0


def _jv_dense(c: List[List[float]]) -> Tuple[List[int], List[int]]:
    """Jonker-Volgenant on a square matrix (list of rows) -> (x, y): x[i] = column of row i, y[j] = row of column j.
    Three phases as lap runs them: column reduction + reduction transfer, two rounds of augmenting row reduction, shortest-path
    augmentation of what is still free.  Tie rules: a column keeps the FIRST cheapest row, of several columns wanting one row the
    lowest-numbered gets it, a row's scan keeps the first minimum."""
    n = len(c)
    x, y, v = [-1] * n, [0] * n, [_LARGE] * n
    for i in range(n):
        for j in range(n):
            if c[i][j] < v[j]:
                v[j], y[j] = c[i][j], i
    unique = [True] * n
    for j in range(n - 1, -1, -1):
        i = y[j]
        if x[i] < 0:
            x[i] = j
        else:
            unique[i] = False
            y[j] = -1
    free = []
    for i in range(n):
        if x[i] < 0:
            free.append(i)
        elif unique[i]:
            j = x[i]
            v[j] -= min((c[i][j2] - v[j2] for j2 in range(n) if j2 != j), default=_LARGE)
    for _ in range(2):                                        # augmenting row reduction
        if not free:
            break
        queue, free, current, rounds = free + [0] * n, [], 0, 0
        n_free = len(queue) - n
        while current < n_free:
            rounds += 1
            fi = queue[current]
            current += 1
            j1, j2, v1, v2 = 0, -1, c[fi][0] - v[0], _LARGE
            for j in range(1, n):
                r = c[fi][j] - v[j]
                if r < v2:
                    if r >= v1:
                        v2, j2 = r, j
                    else:
                        v2, v1, j2, j1 = v1, r, j1, j
            i0 = y[j1]
            v1_new = v[j1] - (v2 - v1)
            lowers = v1_new < v[j1]
            if rounds < current * n:
                if lowers:
                    v[j1] = v1_new
                elif i0 >= 0 and j2 >= 0:
                    j1, i0 = j2, y[j2]
                if i0 >= 0:
                    if lowers:
                        current -= 1
                        queue[current] = i0
                    else:
                        free.append(i0)
            elif i0 >= 0:
                free.append(i0)
            x[fi], y[j1] = j1, fi
    for start in free:                                        # shortest augmenting paths
        cols, pred = list(range(n)), [start] * n
        d = [c[start][j] - v[j] for j in range(n)]
        lo = hi = n_ready = 0
        final = -1
        while final < 0:
            if lo == hi:
                n_ready = lo
                hi, mind = lo + 1, d[cols[lo]]
                for k in range(hi, n):
                    j = cols[k]
                    if d[j] <= mind:
                        if d[j] < mind:
                            hi, mind = lo, d[j]
                        cols[k], cols[hi] = cols[hi], j
                        hi += 1
                for k in range(lo, hi):
                    if y[cols[k]] < 0:
                        final = cols[k]
            if final < 0:
                slo, shi = lo, hi
                while slo != shi and final < 0:
                    j = cols[slo]
                    slo += 1
                    i, mind = y[j], d[j]
                    h = c[i][j] - v[j] - mind
                    for k in range(shi, n):
                        j = cols[k]
                        r = c[i][j] - v[j] - h
                        if r < d[j]:
                            d[j], pred[j] = r, i
                            if r == mind:
                                if y[j] < 0:
                                    final = j
                                    break
                                cols[k], cols[shi] = cols[shi], j
                                shi += 1
                if final < 0:
                    lo, hi = slo, shi
        mind = d[cols[lo]]
        for k in range(n_ready):
            j = cols[k]
            v[j] += d[j] - mind
        j, i = final, -1
        while i != start:
            i = pred[j]
            y[j] = i
            j, x[i] = x[i], j
    return x, y


def lapjv(cost: np.ndarray, cost_limit: float) -> Tuple[np.ndarray, np.ndarray]:
    """``lap.lapjv(cost, extend_cost=True, cost_limit=cost_limit)`` -> (x, y): x[i] = column of row i or -1, y[j] = row of column j
    or -1.  The problem solved is the (rows + cols)^2 one whose off-diagonal blocks cost ``cost_limit / 2`` per cell and whose
    lower-right block is free: leaving a row AND a column unmatched costs exactly ``cost_limit``."""
    cost = np.asarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    if nr == 0 or nc == 0:
        return np.full(nr, -1), np.full(nc, -1)
    n = nr + nc
    ext = np.full((n, n), cost_limit / 2.0)
    ext[nr:, nc:] = 0.0
    ext[:nr, :nc] = cost
    x, y = _jv_dense(ext.tolist())
    x, y = np.asarray(x[:nr]), np.asarray(y[:nc])
    return np.where(x >= nc, -1, x), np.where(y >= nr, -1, y)


def _assign(cost: np.ndarray, limit: float) -> Tuple[List[Tuple[int, int]], List[int], List[int]]:
    """matching.py:linear_assignment(cost, thresh=limit) with lap -> (pairs in row order, free rows, free columns)"""
    rows, cols = range(cost.shape[0]), range(cost.shape[1])
    if cost.size == 0:
        return [], list(rows), list(cols)
    x, y = lapjv(cost, limit)
    pairs = [(int(r), int(c)) for r, c in enumerate(x) if c >= 0]
    return pairs, [r for r in rows if x[r] < 0], [c for c in cols if y[c] < 0]


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

    def __init__(self, frame_rate: int = 30, gmc_method: Optional[str] = "sparseOptFlow"):
        from .gmc_oracle import GMC
        self.gmc = GMC(gmc_method)                      # BOTSORT.__init__: GMC(method=args.gmc_method); None = identity
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

    def update(self, det: np.ndarray, img: Optional[np.ndarray] = None, warp: Optional[np.ndarray] = None) -> np.ndarray:
        """``img``: the frame the detections come from (BGR uint8, as ``tracker.update(det, im0)`` receives it in
        trackers/track.py); without it no motion compensation takes place (byte_tracker.py: ``if ... img is not None``).
        ``warp``: a 2x3 camera-motion matrix to use instead of estimating one from ``img`` (the parity tests hand the product
        tracker and this one the same matrices)."""
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
        if warp is None and img is not None and self.gmc.method is not None:
            # camera motion since the previous frame, applied to the predicted states (pool) and the unconfirmed tracks
            try:
                warp = self.gmc.apply(img)
            except (np.linalg.LinAlgError, ValueError, FloatingPointError, ZeroDivisionError):
                warp = np.eye(2, 3)                             # byte_tracker.py bypasses errors of the gmc module the same way
        if warp is not None and not np.array_equal(warp, np.eye(2, 3)):
            from .gmc_oracle import warp_kalman
            warp = np.asarray(warp, dtype=np.float64).reshape(2, 3)
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
