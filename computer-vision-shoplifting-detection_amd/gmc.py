"""Global motion compensation of BoT-SORT (``gmc_method: sparseOptFlow``, the Ultralytics default behind
``model.track`` -- ``/root/reference/model.py:38``) -- a thin binding of csrc/gmc_kernels.hip and csrc/gmc_host.cpp.

Ultralytics' ``trackers/utils/gmc.py:GMC.apply_sparseoptflow`` estimates, per frame, the 2x3 partial-affine (similarity)
transform of the BACKGROUND between the previous and the current frame and BoT-SORT applies it to the Kalman state of
every track before association (``STrack.multi_gmc``), so that tracks survive camera motion.  The recipe is four OpenCV
calls -- ``cvtColor(BGR2GRAY)``, ``resize`` to half size, ``goodFeaturesToTrack`` (Shi-Tomasi corners),
``calcOpticalFlowPyrLK`` (pyramidal Lucas-Kanade) and ``estimateAffinePartial2D`` (RANSAC) -- and OpenCV is not part of
this stack.  The engine implements those published algorithms from their definitions, with OpenCV's default parameters as
Ultralytics passes them (stated in oracle/gmc_oracle.py, the numpy restatement the tests check this module against):

  * on the GPU (``device=k``): frame preparation and Lucas-Kanade as HIP kernels on a stream of their own (one wavefront per
    corner), corner ordering and RANSAC in host C++ behind them -- what ``model.track`` uses;
  * on the host (``device=None``): every stage in host C++ (csrc/gmc_host.cpp) -- a tracker outside an engine, CPU tests.
    Chosen by argument, never as a fallback: a HIP error in the device path raises.

The state machine of ``GMC.apply`` (previous plane, previous corners) lives in the C++ object (``mi355_gmc_track_*``): a step is
two calls from here -- :meth:`GMC.begin` (enqueue) and :meth:`GMC.apply` (collect -> matrix).  PARITY UNPINNED against OpenCV
(absent here): float64 arithmetic instead of OpenCV's fixed point inside the LK loop and a generator of its own instead of
cv::RNG in RANSAC, so the estimate agrees with OpenCV's to sub-pixel noise, not bit for bit.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib
from .tracker import warp_kalman  # noqa: F401  (STrack.multi_gmc for one track; re-exported for callers of this module)

MAX_CORNERS, QUALITY_LEVEL, BLOCK_SIZE = 1000, 0.01, 3
LK_WIN, LK_LEVELS, LK_MAX_ITERS, LK_EPS, LK_MIN_EIG = 21, 3, 30, 0.01, 1e-4
RANSAC_THRESHOLD, RANSAC_CONFIDENCE, RANSAC_MAX_ITERS = 3.0, 0.99, 2000


def _linear_coeffs(dn: int, sn: int):
    """INTER_LINEAR sample positions of a dn-long axis resampled from sn: source index, two taps in 1/2048 units."""
    scale = sn / dn
    fx = ((np.arange(dn) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(fx).astype(np.int64)
    fx = (fx - s).astype(np.float32)
    lo, hi = s < 0, s >= sn - 1
    s = np.where(lo, 0, np.where(hi, sn - 1, s))
    fx = np.where(lo | hi, np.float32(0), fx)
    c1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    c0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    return s, c0, c1


_TABLES: dict = {}


def _coeff_table(dn: int, sn: int) -> np.ndarray:
    """int32 [dn, 3] rows (source index, tap 0, tap 1) of :func:`_linear_coeffs`, kept per (dn, sn): a video has one frame size"""
    key = (dn, sn)
    if key not in _TABLES:
        if len(_TABLES) > 64:
            _TABLES.clear()
        _TABLES[key] = np.ascontiguousarray(np.stack(_linear_coeffs(dn, sn), axis=1), dtype=np.int32)
    return _TABLES[key]


def prepare_frame(raw_frame: np.ndarray, downscale: int = 2, device: Optional[int] = None, max_corners: int = MAX_CORNERS,
                  quality: float = QUALITY_LEVEL) -> Tuple[np.ndarray, np.ndarray]:
    """The first three OpenCV calls of ``GMC.apply_sparseoptflow`` for one BGR frame -> (gray plane at 1 / downscale, corners
    float32 [n, 2] strongest first).  ``device=k``: csrc/gmc_kernels.hip on GPU k; ``device=None``: the same expressions as host
    C++ (csrc/gmc_host.cpp) -- luma + resize in cv2's fixed point, the structure tensor in float64, threshold and non-maximum
    suppression; then the ordering of the kept corners.  Both give the plane and the corner list of the numpy statement."""
    if raw_frame.ndim != 3:
        raise ValueError("prepare_frame takes a BGR frame [H, W, 3]")
    h, w = raw_frame.shape[:2]
    dh, dw = (h // downscale, w // downscale) if downscale > 1 else (h, w)
    frame = np.ascontiguousarray(raw_frame, dtype=np.uint8)
    xt = yt = None
    if downscale > 1:
        xt, yt = _coeff_table(dw, w), _coeff_table(dh, h)
    gray = np.empty((dh, dw), np.uint8)
    eig = np.empty((dh, dw), np.float32)
    ok = np.empty((dh, dw), np.uint8)
    xp, yp = (xt.ctypes.data if xt is not None else None), (yt.ctypes.data if yt is not None else None)
    if device is None:
        rc = _lib.lib().mi355_gmc_prepare_host(frame.ctypes.data, h, w, dh, dw, xp, yp, float(quality), gray.ctypes.data, eig.ctypes.data, ok.ctypes.data)
    else:
        rc = _lib.lib().mi355_gmc_prepare_device(int(device), frame.ctypes.data, h, w, dh, dw, xp, yp, float(quality), gray.ctypes.data, eig.ctypes.data,
                                                 ok.ctypes.data)
    if rc == -2:
        raise RuntimeError(f"mi355_gmc_prepare_device: HIP error on device {device}")
    if rc != 0:
        raise ValueError("mi355_gmc_prepare: bad argument")
    return gray, order_corners(eig, ok, max_corners)


def calc_optical_flow_pyr_lk(prev: np.ndarray, cur: np.ndarray, pts: np.ndarray, win: int = LK_WIN, levels: int = LK_LEVELS,
                             max_iters: int = LK_MAX_ITERS, eps: float = LK_EPS, min_eig: float = LK_MIN_EIG,
                             device: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """cv2.calcOpticalFlowPyrLK as the product runs it.  ``device=None``: the host C++ loops of csrc/gmc_host.cpp (the numpy
    statement of the algorithm is oracle/gmc_oracle.py:calc_optical_flow_pyr_lk, its cross-check in the tests).
    ``device=k``: the HIP kernels of csrc/gmc_kernels.hip on GPU k (one wavefront per point; what ``model.track`` uses: on the
    host the 1000-corner budget costs 10-27 ms per frame, thirty times the detector pass)."""
    import ctypes as C
    from . import _lib
    prev, cur = np.ascontiguousarray(prev, dtype=np.uint8), np.ascontiguousarray(cur, dtype=np.uint8)
    if prev.shape != cur.shape or prev.ndim != 2:
        raise ValueError("prev and cur must be gray planes of one shape")
    p = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
    nxt = np.zeros_like(p)
    status = np.zeros(len(p), dtype=np.uint8)
    if device is None:
        rc = _lib.lib().mi355_gmc_pyr_lk(prev.ctypes.data, cur.ctypes.data, prev.shape[0], prev.shape[1], p.ctypes.data, len(p), int(win), int(levels),
                                         int(max_iters), float(eps), float(min_eig), nxt.ctypes.data, status.ctypes.data)
    else:
        rc = _lib.lib().mi355_gmc_pyr_lk_device(int(device), prev.ctypes.data, cur.ctypes.data, prev.shape[0], prev.shape[1], p.ctypes.data, len(p),
                                                int(win), int(levels), int(max_iters), float(eps), float(min_eig), nxt.ctypes.data,
                                                status.ctypes.data)
    if rc == -2:
        raise RuntimeError(f"mi355_gmc_pyr_lk_device: HIP error on device {device} (no GPU? the host routine is device=None)")
    if rc != 0:
        raise ValueError("mi355_gmc_pyr_lk: bad argument")
    return nxt, status.astype(bool)


def order_corners(eig: np.ndarray, ok: np.ndarray, max_corners: int = MAX_CORNERS) -> np.ndarray:
    """goodFeaturesToTrack's last step on a corner map: kept corners strongest first, raster order among equals -> float32 [n, 2]
    (host C++, csrc/gmc_host.cpp; the numpy form is the tail of :func:`good_features_to_track`)."""
    from . import _lib
    eig = np.ascontiguousarray(eig, dtype=np.float32)
    ok = np.ascontiguousarray(ok, dtype=np.uint8)
    out = np.empty((max(1, max_corners), 2), np.float32)
    n = _lib.lib().mi355_gmc_order_corners(eig.ctypes.data, ok.ctypes.data, eig.shape[0], eig.shape[1], int(max_corners), out.ctypes.data)
    if n < 0:
        raise ValueError("mi355_gmc_order_corners: bad argument")
    return out[:n].copy()


def estimate_affine_partial_2d_host(src: np.ndarray, dst: np.ndarray, threshold: float = RANSAC_THRESHOLD, confidence: float = RANSAC_CONFIDENCE,
                                    max_iters: int = RANSAC_MAX_ITERS, seed: int = 0) -> Tuple[Optional[np.ndarray], np.ndarray]:
    """:func:`estimate_affine_partial_2d` as host C++ loops (csrc/gmc_host.cpp: what the tracker runs, 10x less time per frame);
    same algorithm, draws from that routine's own seeded generator."""
    from . import _lib
    src = np.ascontiguousarray(src, dtype=np.float64).reshape(-1, 2)
    dst = np.ascontiguousarray(dst, dtype=np.float64).reshape(-1, 2)
    n = len(src)
    H = np.zeros((2, 3), np.float64)
    mask = np.zeros(n, np.uint8)
    rc = _lib.lib().mi355_gmc_affine_partial(src.ctypes.data, dst.ctypes.data, n, float(threshold), float(confidence), int(max_iters), int(seed),
                                             H.ctypes.data, mask.ctypes.data)
    if rc < 0:
        raise ValueError("mi355_gmc_affine_partial: bad argument")
    return (H if rc == 1 else None), mask.astype(bool)


# ------------------------------------------------------------------------------------------------- the GMC object
class GMC:
    """``GMC(method="sparseOptFlow", downscale=2).apply(frame_bgr) -> 2x3`` (float64); identity on the first frame, when too
    few points survive, or when ``method`` is None / "none".

    ``device=k`` runs the frame preparation and the optical flow on GPU k (csrc/gmc_kernels.hip) and splits a step in two:
    :meth:`begin` enqueues it (``model.track`` does so BEFORE the detector runs on the frame, so that both share the GPU),
    :meth:`apply` collects it -- or enqueues and collects, when nobody called ``begin`` for that frame.  ``device=None``: the
    same object with every stage in host C++."""

    def __init__(self, method: Optional[str] = "sparseOptFlow", downscale: int = 2, device: Optional[int] = None):
        self.device = device
        if method in ("none", "None"):
            method = None
        if method not in (None, "sparseOptFlow"):
            raise ValueError(f"GMC method {method!r} is not implemented (sparseOptFlow, the botsort.yaml default, or None)")
        self.method, self.downscale = method, max(1, int(downscale))
        self._h = None                         # mi355_gmc object, created by the first step
        self._pending = None                   # the frame object of the enqueued step
        self._H = np.empty((2, 3), np.float64)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value:
            try:
                _lib.lib().mi355_gmc_destroy(h)
            except Exception:
                pass

    def _obj(self):
        if self._h is None:
            h = C.c_void_p()
            rc = _lib.lib().mi355_gmc_create(-1 if self.device is None else int(self.device), C.byref(h))
            if rc != 0:
                raise RuntimeError(f"mi355_gmc_create: error {rc} on device {self.device}")
            self._h = h
        return self._h

    def reset(self) -> None:
        if self._h is not None:
            _lib.lib().mi355_gmc_track_reset(self._h)
        self._pending = None

    # ---- the previous frame as the object holds it (tests) -------------------------------------------------------------------------
    def _state(self):
        if self._h is None:
            return None, None
        oh, ow, n = C.c_int(), C.c_int(), C.c_int()
        _lib.lib().mi355_gmc_track_state(self._h, C.byref(oh), C.byref(ow), C.byref(n), None, None, 0)
        if oh.value == 0:
            return None, None
        gray, pts = np.empty((oh.value, ow.value), np.uint8), np.empty((max(n.value, 1), 2), np.float32)
        _lib.lib().mi355_gmc_track_state(self._h, None, None, None, gray.ctypes.data, pts.ctypes.data, len(pts))
        return gray, pts[:n.value]

    @property
    def prev_frame(self) -> Optional[np.ndarray]:
        return self._state()[0]

    @property
    def prev_points(self) -> Optional[np.ndarray]:
        return self._state()[1]

    # ---- the step ------------------------------------------------------------------------------------------------------------
    def begin(self, raw_frame: np.ndarray) -> None:
        """Enqueue this frame's step (on the GPU for a device object); :meth:`apply` of the same frame object collects it."""
        if self.method is None or raw_frame is None or raw_frame.ndim != 3 or self._pending is not None:
            return
        frame = raw_frame if (raw_frame.dtype == np.uint8 and raw_frame.flags.c_contiguous) else np.ascontiguousarray(raw_frame, dtype=np.uint8)
        rc = _lib.lib().mi355_gmc_track_begin(self._obj(), frame.ctypes.data, frame.shape[0], frame.shape[1], self.downscale)
        if rc == -2:
            raise RuntimeError(f"mi355_gmc_track_begin: HIP error on device {self.device}")
        if rc != 0:
            raise ValueError(f"mi355_gmc_track_begin: error {rc}")
        self._pending = raw_frame

    def batch_seq(self) -> int:
        """number of the last batch upload of this object (take it BEFORE starting :meth:`apply_batch` on another thread)"""
        return int(_lib.lib().mi355_gmc_batch_seq(self._obj())) if (self.method is not None and self.device is not None) else 0

    def batch_device_frames(self, after_seq: int, timeout_ms: int = 2000):
        """(device pointer, n, height, width) of the frames an :meth:`apply_batch` call -- running on another thread, started after
        :meth:`batch_seq` returned ``after_seq`` -- has uploaded for its batch, or None (host object; nothing uploaded within the timeout;
        frames not dense on the device).  ``sweep.process_clip`` runs the detector pass of the batch on them while ``apply_batch`` carries
        on: one staging copy and one upload instead of two."""
        if self.method is None or self.device is None:
            return None
        ptr, n, h, w, stride = C.c_void_p(), C.c_int(), C.c_int(), C.c_int(), C.c_longlong()
        rc = _lib.lib().mi355_gmc_batch_frames(self._obj(), int(after_seq), int(timeout_ms), C.byref(ptr), C.byref(n), C.byref(h), C.byref(w), C.byref(stride))
        if rc == -2:
            raise RuntimeError(f"mi355_gmc_batch_frames: HIP error on device {self.device}")
        if rc != 0 or not ptr.value or stride.value != h.value * w.value * 3:
            return None
        return ptr.value, n.value, h.value, w.value

    def pending_device_frame(self):
        """(device pointer, height, width) of the frame :meth:`begin` has just uploaded for its step -- dense BGR uint8 on this object's
        GPU, valid until the next :meth:`begin` -- or None (host object, nothing pending).  ``YOLO.track`` hands it to the detector pass
        of the same frame: one upload instead of two, and the detector's kernels no longer queue behind this step's copies."""
        if self.method is None or self._pending is None or self.device is None:
            return None
        ptr, h, w = C.c_void_p(), C.c_int(), C.c_int()
        rc = _lib.lib().mi355_gmc_pending_frame(self._obj(), C.byref(ptr), C.byref(h), C.byref(w))
        if rc == -2:
            raise RuntimeError(f"mi355_gmc_pending_frame: HIP error on device {self.device}")
        return (ptr.value, h.value, w.value) if rc == 0 and ptr.value else None

    def apply_batch(self, frames) -> np.ndarray:
        """The warps of n consecutive frames of one video in ONE call -> float64 [n, 2, 3], continuing from the object's previous frame:
        all n frame preparations as one set of launches and all n Lucas-Kanade steps as one launch on the GPU, corner ordering and
        RANSAC on host threads (``mi355_gmc_track_batch``).  Bit for bit what n ``apply`` calls return -- for a caller that holds the
        frames of a detector batch before the tracker needs their warps (``sweep.process_clip``), at a fraction of the latency."""
        frames = list(frames)
        n = len(frames)
        if self.method is None or n == 0:
            return np.tile(np.eye(2, 3), (n, 1, 1))
        if any(f.ndim != 3 or f.shape != frames[0].shape for f in frames):
            raise ValueError("apply_batch takes BGR frames [H, W, 3] of one size")
        if self._pending is not None:
            self.reset()                                         # a step enqueued for a single frame: stale
        keep = [f if (f.dtype == np.uint8 and f.flags.c_contiguous) else np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
        ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in keep])
        H = np.empty((n, 2, 3), np.float64)
        rc = _lib.lib().mi355_gmc_track_batch(self._obj(), ptrs, n, keep[0].shape[0], keep[0].shape[1], self.downscale, H.ctypes.data)
        if rc == -2:
            raise RuntimeError(f"mi355_gmc_track_batch: HIP error on device {self.device}")
        if rc != 0:
            raise ValueError(f"mi355_gmc_track_batch: error {rc}")
        return H

    def apply(self, raw_frame: np.ndarray, detections=None) -> np.ndarray:
        if self.method is None or raw_frame is None:
            return np.eye(2, 3)
        if raw_frame.ndim != 3:
            raise ValueError("GMC.apply takes a BGR frame [H, W, 3]")
        if self._pending is not None and self._pending is not raw_frame:
            self.reset()                                         # a step enqueued for another frame: its results are stale
        if self._pending is None:
            self.begin(raw_frame)
        self._pending = None
        rc = _lib.lib().mi355_gmc_track_finish(self._h, self._H.ctypes.data)
        if rc == -2:
            raise RuntimeError(f"mi355_gmc_track_finish: HIP error on device {self.device}")
        if rc != 0:
            raise ValueError(f"mi355_gmc_track_finish: error {rc}")
        return self._H.copy()
