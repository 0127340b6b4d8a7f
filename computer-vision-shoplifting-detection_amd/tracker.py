"""The multi-object tracker behind ``YOLO.track`` -- a thin binding of csrc/tracker_host.cpp.

The reference calls ``model.track(frame, persist=True, classes=[0])`` (``/root/reference/model.py:38``), which in
ultralytics==8.3.225 runs predict at conf 0.1 and then its default ``botsort.yaml`` tracker: BoT-SORT = ByteTrack's two-stage
association (Zhang et al., "ByteTrack", ECCV 2022) on IoU cost with score fusion, a constant-velocity Kalman filter over
(cx, cy, w, h) (Aharon et al., "BoT-SORT", 2022), a 30-frame lost-track buffer, one-frame confirmation of tracks born after the
first frame, ``lap.lapjv(extend_cost=True, cost_limit=thresh)`` as the assignment solver, and output rows
``[x1,y1,x2,y2,id,score,cls,idx]`` whose box is the filter state.  All of that -- filter, cost matrices, Jonker-Volgenant
assignment, track bookkeeping, the camera-motion warp of the predicted states (``STrack.multi_gmc``) -- is host C++ behind the C ABI
(``mi355_tracker_*`` in include/mi355_yolo.h): the tracker is sequential per video and a few dozen boxes per frame, so it stays
on the host as in the reference, but 0.3 ms of numpy per frame had made it, not the detector, the bottleneck of the reference's
frame loop.  Global motion compensation (``gmc_method: sparseOptFlow``, the botsort.yaml default) is ``gmc.py`` (HIP kernels or
host C++).  Not implemented: ReID (``with_reid: False`` is the default).

The numpy statement of the same tracker is the checker, ``oracle/tracker_oracle.py`` (tests only).  PARITY UNPINNED against a
real Ultralytics / lap run (neither is installable here); pinned by hand-derived known answers
(``tests/test_tracker_known_answers.py``) and against the oracle on scripted and random clips (``tests/test_tracker_core.py``).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from . import _lib

# botsort.yaml defaults (compiled into csrc/tracker_host.cpp; stated here for callers)
TRACK_HIGH_THRESH, TRACK_LOW_THRESH, NEW_TRACK_THRESH, TRACK_BUFFER, MATCH_THRESH, FUSE_SCORE = 0.25, 0.1, 0.25, 30, 0.8, True
TRACKED, LOST, RETIRED = 1, 2, 3
_IDENTITY = np.eye(2, 3)


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class KalmanFilterXYWH:
    """Constant-velocity filter on (cx, cy, w, h): ``initiate`` / ``predict`` / ``update`` take and return ``(mean[8], covariance[8, 8])``
    (trackers/utils/kalman_filter.py:KalmanFilterXYWH; std weights 1/20 and 1/160 of the box size)."""

    @staticmethod
    def initiate(z: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        z = np.ascontiguousarray(z, dtype=np.float64)
        mean, cov = np.empty(8), np.empty((8, 8))
        _lib.lib().mi355_kalman_initiate(_dp(z), _dp(mean), _dp(cov))
        return mean, cov

    @staticmethod
    def predict(mean: np.ndarray, cov: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        mean, cov = np.array(mean, dtype=np.float64), np.array(cov, dtype=np.float64)
        _lib.lib().mi355_kalman_predict(_dp(mean), _dp(cov))
        return mean, cov

    @staticmethod
    def update(mean: np.ndarray, cov: np.ndarray, z: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        mean, cov, z = np.array(mean, dtype=np.float64), np.array(cov, dtype=np.float64), np.ascontiguousarray(z, dtype=np.float64)
        _lib.lib().mi355_kalman_update(_dp(mean), _dp(cov), _dp(z))
        return mean, cov


def warp_kalman(mean: np.ndarray, cov: np.ndarray, H: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """``STrack.multi_gmc`` for one track: the rotation / scale block of the 2x3 ``H`` acts on every (x, y)-like pair of the state,
    the translation on the centre only; P <- R8 P R8' with R8 = kron(I4, H[:2, :2])."""
    mean, cov = np.array(mean, dtype=np.float64), np.array(cov, dtype=np.float64)
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(2, 3)
    _lib.lib().mi355_kalman_warp(_dp(mean), _dp(cov), _dp(H))
    return mean, cov


def lapjv(cost: np.ndarray, cost_limit: float) -> Tuple[np.ndarray, np.ndarray]:
    """``lap.lapjv(cost, extend_cost=True, cost_limit=cost_limit)[1:]``: x[i] = column of row i or -1, y[j] = row of column j or -1."""
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    if cost.ndim != 2:
        raise ValueError("cost must be a matrix")
    x, y = np.full(cost.shape[0], -1, np.int32), np.full(cost.shape[1], -1, np.int32)
    if _lib.lib().mi355_lapjv(_dp(cost), cost.shape[0], cost.shape[1], float(cost_limit), _dp(x), _dp(y)) != 0:
        raise ValueError("mi355_lapjv: bad argument")
    return x, y


class TrackView:
    """Read-only snapshot of one track (``tracked_stracks`` / ``lost_stracks``)."""
    __slots__ = ("track_id", "state", "confirmed", "born", "seen", "score", "cls", "idx", "mean")

    def __init__(self, row: np.ndarray):
        self.track_id, self.state, self.confirmed = int(row[0]), int(row[1]), bool(row[2])
        self.born, self.seen = int(row[3]), int(row[4])
        self.score, self.cls, self.idx = float(row[5]), float(row[6]), float(row[7])
        self.mean = row[8:16].copy()

    @property
    def xyxy(self) -> np.ndarray:
        x, y, w, h = self.mean[:4]
        return np.array([x - w / 2, y - h / 2, x + w / 2, y + h / 2])


class BYTETracker:
    """``update(det [N,6] = x1,y1,x2,y2,conf,cls) -> [M,8] = x1,y1,x2,y2,id,score,cls,idx`` (idx = row of ``det``), one call
    per frame, EVERY frame (an empty frame still ages the lost tracks)."""

    def __init__(self, frame_rate: int = 30, gmc_method: Optional[str] = "sparseOptFlow", gmc_device: Optional[int] = None):
        from .gmc import GMC
        self.gmc = GMC(gmc_method, device=gmc_device)   # BOTSORT.__init__: GMC(method=args.gmc_method); None = identity
        self._h = C.c_void_p()
        if _lib.lib().mi355_tracker_create(int(frame_rate), C.byref(self._h)) != 0:
            raise ValueError("mi355_tracker_create: bad frame rate")
        self._out = np.empty((64, 8), np.float32)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value:
            try:
                _lib.lib().mi355_tracker_destroy(h)
            except Exception:
                pass

    # ---- state of the C++ object ---------------------------------------------------------------------------------------------
    def _state(self) -> Tuple[int, int, int, int]:
        v = [C.c_int() for _ in range(4)]
        _lib.lib().mi355_tracker_state(self._h, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    @property
    def frame_id(self) -> int:
        return self._state()[0]

    @property
    def _ids_issued(self) -> int:
        return self._state()[1]

    def _tracks(self, which: int) -> List[TrackView]:
        n = self._state()[2 + which]
        buf = np.empty((max(n, 1), 16), np.float64)
        n = _lib.lib().mi355_tracker_tracks(self._h, which, _dp(buf), len(buf))
        return [TrackView(buf[i]) for i in range(min(n, len(buf)))]

    @property
    def tracked_stracks(self) -> List[TrackView]:
        return self._tracks(0)

    @property
    def lost_stracks(self) -> List[TrackView]:
        return self._tracks(1)

    # ---- one frame -----------------------------------------------------------------------------------------------------------
    def update(self, det: np.ndarray, img: Optional[np.ndarray] = None, next_img: Optional[np.ndarray] = None,
               warp: Optional[np.ndarray] = None) -> np.ndarray:
        """``img``: the frame the detections come from (BGR uint8, as ``tracker.update(det, im0)`` receives it in
        trackers/track.py); without it no motion compensation takes place (byte_tracker.py: ``if ... img is not None``).
        ``next_img``: the frame the NEXT call will bring, when the caller already holds it (a batched sweep): its motion-compensation
        step is enqueued on the GPU as soon as this frame's has been collected, and runs beside this call's association.
        ``warp``: a 2x3 camera-motion matrix to use instead of estimating one from ``img`` (tests)."""
        det = np.ascontiguousarray(det, dtype=np.float32).reshape(-1, 6)
        if warp is None and img is not None and self.gmc.method is not None:
            try:
                warp = self.gmc.apply(img)
            except (np.linalg.LinAlgError, ValueError, FloatingPointError, ZeroDivisionError):
                # byte_tracker.py bypasses errors of the gmc module the same way (degenerate point sets); a RuntimeError of the
                # device path (HIP error, library without the kernels) is NOT one of them and surfaces
                warp = _IDENTITY
            if next_img is not None:
                self.gmc.begin(next_img)                        # device path only; a no-op on the host
        wp = None
        if warp is not None:
            warp = np.ascontiguousarray(warp, dtype=np.float64).reshape(2, 3)
            wp = _dp(warp)
        n = _lib.lib().mi355_tracker_update(self._h, _dp(det), len(det), wp, _dp(self._out), len(self._out))
        if n < 0:
            raise ValueError("mi355_tracker_update: bad argument")
        if n > len(self._out):                                  # more confirmed tracks than the buffer holds: fetch the rows again
            self._out = np.empty((2 * n, 8), np.float32)
            n = _lib.lib().mi355_tracker_last_rows(self._h, _dp(self._out), len(self._out))
        return self._out[:n].copy()
