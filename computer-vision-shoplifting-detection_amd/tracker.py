"""Host-side multi-object tracker behind ``YOLO.track``.

The reference calls ``model.track(frame, persist=True, classes=[0])`` (``/root/reference/model.py:38``), which in
ultralytics==8.3.225 runs predict at conf 0.1 and then the default ``botsort.yaml`` tracker
(``trackers/bot_sort.py`` on top of ``trackers/byte_tracker.py``): two-stage IoU association (high- then
low-score detections) with score fusion, an XYWH Kalman filter (``trackers/utils/kalman_filter.py:KalmanFilterXYWH``),
track ids, and rows ``[x1,y1,x2,y2,id,score,cls,idx]`` whose box is the Kalman state.

Restated here without the two parts that need OpenCV / a ReID net: global motion compensation
(``gmc_method: sparseOptFlow`` -- identity here, correct for the static CCTV cameras of UCF-Crime) and ReID
(``with_reid: False`` by default).  Assignment uses scipy's Hungarian solver the way Ultralytics' own
``linear_assignment(use_lap=False)`` fallback does.  Tracking is sequential per video and stays on the host; it is
small next to the detector.  PARITY UNPINNED against real BoT-SORT (no ultralytics / lap / cv2 here).
"""
from __future__ import annotations

from typing import List

import numpy as np
import scipy.linalg
from scipy.optimize import linear_sum_assignment

TRACK_HIGH_THRESH = 0.25
TRACK_LOW_THRESH = 0.1
NEW_TRACK_THRESH = 0.25
TRACK_BUFFER = 30
MATCH_THRESH = 0.8
FUSE_SCORE = True

NEW, TRACKED, LOST, REMOVED = 0, 1, 2, 3


class KalmanFilterXYWH:
    """8-d state (x, y, w, h, vx, vy, vw, vh), constant velocity."""

    def __init__(self):
        ndim, dt = 4, 1.0
        self._motion_mat = np.eye(2 * ndim, 2 * ndim)
        for i in range(ndim):
            self._motion_mat[i, ndim + i] = dt
        self._update_mat = np.eye(ndim, 2 * ndim)
        self._std_weight_position = 1.0 / 20
        self._std_weight_velocity = 1.0 / 160

    def initiate(self, measurement):
        mean = np.r_[measurement, np.zeros_like(measurement)]
        w, h = measurement[2], measurement[3]
        std = [2 * self._std_weight_position * w, 2 * self._std_weight_position * h,
               2 * self._std_weight_position * w, 2 * self._std_weight_position * h,
               10 * self._std_weight_velocity * w, 10 * self._std_weight_velocity * h,
               10 * self._std_weight_velocity * w, 10 * self._std_weight_velocity * h]
        return mean, np.diag(np.square(std))

    def _noise(self, w, h):
        sp, sv = self._std_weight_position, self._std_weight_velocity
        return np.diag(np.square(np.r_[[sp * w, sp * h, sp * w, sp * h], [sv * w, sv * h, sv * w, sv * h]]))

    def predict(self, mean, covariance):
        motion_cov = self._noise(mean[2], mean[3])
        mean = np.dot(mean, self._motion_mat.T)
        covariance = np.linalg.multi_dot((self._motion_mat, covariance, self._motion_mat.T)) + motion_cov
        return mean, covariance

    def project(self, mean, covariance):
        sp = self._std_weight_position
        innovation_cov = np.diag(np.square([sp * mean[2], sp * mean[3], sp * mean[2], sp * mean[3]]))
        mean = np.dot(self._update_mat, mean)
        covariance = np.linalg.multi_dot((self._update_mat, covariance, self._update_mat.T))
        return mean, covariance + innovation_cov

    def update(self, mean, covariance, measurement):
        projected_mean, projected_cov = self.project(mean, covariance)
        chol_factor, lower = scipy.linalg.cho_factor(projected_cov, lower=True, check_finite=False)
        kalman_gain = scipy.linalg.cho_solve((chol_factor, lower), np.dot(covariance, self._update_mat.T).T,
                                             check_finite=False).T
        innovation = measurement - projected_mean
        new_mean = mean + np.dot(innovation, kalman_gain.T)
        new_covariance = covariance - np.linalg.multi_dot((kalman_gain, projected_cov, kalman_gain.T))
        return new_mean, new_covariance


class STrack:
    def __init__(self, xywh_idx, score, cls):
        self._xywh = np.asarray(xywh_idx[:4], dtype=np.float32)       # centre x, y, w, h
        self.idx = xywh_idx[-1]
        self.score, self.cls = score, cls
        self.kalman_filter = None
        self.mean = self.covariance = None
        self.is_activated = False
        self.state = NEW
        self.track_id = 0
        self.tracklet_len = 0
        self.frame_id = self.start_frame = 0

    @property
    def end_frame(self):
        return self.frame_id

    def predict(self):
        mean = self.mean.copy()
        if self.state != TRACKED:
            mean[6] = 0
            mean[7] = 0
        self.mean, self.covariance = self.kalman_filter.predict(mean, self.covariance)

    def activate(self, kalman_filter, frame_id, track_id):
        self.kalman_filter = kalman_filter
        self.track_id = track_id
        self.mean, self.covariance = kalman_filter.initiate(self._xywh.astype(np.float64))
        self.tracklet_len = 0
        self.state = TRACKED
        if frame_id == 1:
            self.is_activated = True
        self.frame_id = self.start_frame = frame_id

    def re_activate(self, new_track, frame_id):
        self.mean, self.covariance = self.kalman_filter.update(self.mean, self.covariance, new_track._xywh.astype(np.float64))
        self.tracklet_len = 0
        self.state = TRACKED
        self.is_activated = True
        self.frame_id = frame_id
        self.score, self.cls, self.idx = new_track.score, new_track.cls, new_track.idx

    def update(self, new_track, frame_id):
        self.frame_id = frame_id
        self.tracklet_len += 1
        self.mean, self.covariance = self.kalman_filter.update(self.mean, self.covariance, new_track._xywh.astype(np.float64))
        self.state = TRACKED
        self.is_activated = True
        self.score, self.cls, self.idx = new_track.score, new_track.cls, new_track.idx

    @property
    def tlwh(self):
        if self.mean is None:
            ret = self._xywh.astype(np.float64).copy()
        else:
            ret = self.mean[:4].copy()
        ret[:2] -= ret[2:] / 2
        return ret

    @property
    def xyxy(self):
        ret = self.tlwh.copy()
        ret[2:] += ret[:2]
        return ret

    @property
    def result(self):
        return [*self.xyxy.tolist(), self.track_id, self.score, self.cls, self.idx]


def _iou_matrix(a, b) -> np.ndarray:
    """utils/metrics.py:bbox_ioa(iou=True) on xyxy boxes (eps 1e-7)."""
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)), dtype=np.float32)
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    iw = (np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0])).clip(0)
    ih = (np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1])).clip(0)
    inter = iw * ih
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (area_a[:, None] + area_b[None, :] - inter + 1e-7)


def iou_distance(atracks: List[STrack], btracks: List[STrack]) -> np.ndarray:
    return 1 - _iou_matrix([t.xyxy for t in atracks], [t.xyxy for t in btracks])


def fuse_score(cost_matrix: np.ndarray, detections: List[STrack]) -> np.ndarray:
    if cost_matrix.size == 0:
        return cost_matrix
    iou_sim = 1 - cost_matrix
    det_scores = np.array([d.score for d in detections])[None].repeat(cost_matrix.shape[0], axis=0)
    return 1 - iou_sim * det_scores


def linear_assignment(cost_matrix: np.ndarray, thresh: float):
    """trackers/utils/matching.py:linear_assignment(use_lap=False)."""
    if cost_matrix.size == 0:
        return np.empty((0, 2), dtype=int), list(range(cost_matrix.shape[0])), list(range(cost_matrix.shape[1]))
    x, y = linear_sum_assignment(cost_matrix)
    matches = np.asarray([[x[i], y[i]] for i in range(len(x)) if cost_matrix[x[i], y[i]] <= thresh])
    if len(matches) == 0:
        return np.empty((0, 2), dtype=int), list(range(cost_matrix.shape[0])), list(range(cost_matrix.shape[1]))
    um_a = sorted(frozenset(range(cost_matrix.shape[0])) - frozenset(matches[:, 0]))
    um_b = sorted(frozenset(range(cost_matrix.shape[1])) - frozenset(matches[:, 1]))
    return matches, um_a, um_b


def _joint(a, b):
    seen, res = set(), []
    for t in list(a) + list(b):
        if t.track_id not in seen:
            seen.add(t.track_id)
            res.append(t)
    return res


def _sub(a, b):
    ids = {t.track_id for t in b}
    return [t for t in a if t.track_id not in ids]


def _remove_duplicates(a, b):
    pdist = iou_distance(a, b)
    pairs = np.where(pdist < 0.15)
    dupa, dupb = [], []
    for p, q in zip(*pairs):
        timep = a[p].frame_id - a[p].start_frame
        timeq = b[q].frame_id - b[q].start_frame
        if timep > timeq:
            dupb.append(q)
        else:
            dupa.append(p)
    return [t for i, t in enumerate(a) if i not in dupa], [t for i, t in enumerate(b) if i not in dupb]


class BYTETracker:
    """trackers/byte_tracker.py:BYTETracker.update with BoT-SORT's XYWH Kalman filter."""

    def __init__(self, frame_rate: int = 30):
        self.tracked_stracks: List[STrack] = []
        self.lost_stracks: List[STrack] = []
        self.removed_stracks: List[STrack] = []
        self.frame_id = 0
        self.max_time_lost = int(frame_rate / 30.0 * TRACK_BUFFER)
        self.kalman_filter = KalmanFilterXYWH()
        # ids are owned by the tracker instance (1, 2, ... in activation order): a second tracker created while this one is
        # alive (sweep / poselift bridge next to model.track(persist=True)) cannot disturb them.  For one tracker per process
        # this is what Ultralytics' class-wide BaseTrack counter + reset_id() in __init__ produces.
        self._ids_issued = 0

    def _next_id(self) -> int:
        self._ids_issued += 1
        return self._ids_issued

    @staticmethod
    def _init_track(dets, scores, cls):
        return [STrack(xywh, s, c) for xywh, s, c in zip(dets, scores, cls)] if len(dets) else []

    def update(self, det: np.ndarray) -> np.ndarray:
        """det: [N,6] x1,y1,x2,y2,conf,cls -> [M,8] x1,y1,x2,y2,id,score,cls,idx (idx = row of ``det``)."""
        self.frame_id += 1
        activated, refind, lost, removed = [], [], [], []
        scores, cls = det[:, 4], det[:, 5]
        xywh = np.stack([(det[:, 0] + det[:, 2]) / 2, (det[:, 1] + det[:, 3]) / 2, det[:, 2] - det[:, 0],
                         det[:, 3] - det[:, 1], np.arange(len(det), dtype=det.dtype)], 1)
        remain = scores >= TRACK_HIGH_THRESH
        second = (scores > TRACK_LOW_THRESH) & (scores < TRACK_HIGH_THRESH)
        detections = self._init_track(xywh[remain], scores[remain], cls[remain])
        unconfirmed = [t for t in self.tracked_stracks if not t.is_activated]
        tracked = [t for t in self.tracked_stracks if t.is_activated]
        pool = _joint(tracked, self.lost_stracks)
        for t in pool:
            t.predict()
        dists = iou_distance(pool, detections)
        if FUSE_SCORE:
            dists = fuse_score(dists, detections)
        matches, u_track, u_det = linear_assignment(dists, MATCH_THRESH)
        for it, idet in matches:
            t, d = pool[it], detections[idet]
            if t.state == TRACKED:
                t.update(d, self.frame_id)
                activated.append(t)
            else:
                t.re_activate(d, self.frame_id)
                refind.append(t)
        det2 = self._init_track(xywh[second], scores[second], cls[second])
        r_tracked = [pool[i] for i in u_track if pool[i].state == TRACKED]
        matches, u_track2, _ = linear_assignment(iou_distance(r_tracked, det2), 0.5)
        for it, idet in matches:
            t, d = r_tracked[it], det2[idet]
            if t.state == TRACKED:
                t.update(d, self.frame_id)
                activated.append(t)
            else:
                t.re_activate(d, self.frame_id)
                refind.append(t)
        for it in u_track2:
            t = r_tracked[it]
            if t.state != LOST:
                t.state = LOST
                lost.append(t)
        detections = [detections[i] for i in u_det]
        dists = iou_distance(unconfirmed, detections)
        if FUSE_SCORE:
            dists = fuse_score(dists, detections)
        matches, u_unconf, u_det = linear_assignment(dists, 0.7)
        for it, idet in matches:
            unconfirmed[it].update(detections[idet], self.frame_id)
            activated.append(unconfirmed[it])
        for it in u_unconf:
            unconfirmed[it].state = REMOVED
            removed.append(unconfirmed[it])
        for inew in u_det:
            t = detections[inew]
            if t.score < NEW_TRACK_THRESH:
                continue
            t.activate(self.kalman_filter, self.frame_id, self._next_id())
            activated.append(t)
        for t in self.lost_stracks:
            if self.frame_id - t.end_frame > self.max_time_lost:
                t.state = REMOVED
                removed.append(t)
        self.tracked_stracks = [t for t in self.tracked_stracks if t.state == TRACKED]
        self.tracked_stracks = _joint(self.tracked_stracks, activated)
        self.tracked_stracks = _joint(self.tracked_stracks, refind)
        self.lost_stracks = _sub(self.lost_stracks, self.tracked_stracks)
        self.lost_stracks.extend(lost)
        self.lost_stracks = _sub(self.lost_stracks, self.removed_stracks)
        self.tracked_stracks, self.lost_stracks = _remove_duplicates(self.tracked_stracks, self.lost_stracks)
        self.removed_stracks.extend(removed)
        if len(self.removed_stracks) > 1000:
            self.removed_stracks = self.removed_stracks[-999:]
        return np.asarray([t.result for t in self.tracked_stracks if t.is_activated], dtype=np.float32).reshape(-1, 8)
