"""CPU restatement (numpy) of BoT-SORT's global motion compensation -- TEST INFRASTRUCTURE, not product code: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product (cvsd_amd/gmc.py) runs HIP kernels
(csrc/gmc_kernels.hip) or host C++ (csrc/gmc_host.cpp) and never imports it.

What it restates: ``ultralytics/trackers/utils/gmc.py:GMC.apply_sparseoptflow`` (ultralytics==8.3.225, un-vendored:
/root/reference/requirements.txt:121), reached from ``/root/reference/model.py:38`` through ``model.track``: per frame the 2x3
partial-affine transform of the background between the previous and the current frame, from four OpenCV calls
(opencv-python==4.12.0.88, requirements.txt:69, absent here) -- ``cvtColor(BGR2GRAY)``, ``resize`` to half size,
``goodFeaturesToTrack`` (Shi-Tomasi corners), ``calcOpticalFlowPyrLK`` (Bouguet's pyramidal Lucas-Kanade) and
``estimateAffinePartial2D`` (RANSAC) -- with OpenCV's default parameters as Ultralytics passes them:

  * gray         : 14-bit fixed-point luma  (1868 B + 9617 G + 4899 R + 8192) >> 14
  * half size    : INTER_LINEAR on uint8 with 11-bit coefficients (for an exact 1/2 scale: the mean of each 2x2 block)
  * corners      : min-eigenvalue of the 3x3-block structure tensor of 3x3 Sobel gradients, quality 0.01 of the best
                   corner, 3x3 non-maximum suppression, strongest first, at most 1000, minDistance 1
  * optical flow : 21x21 windows, 4 pyramid levels (5-tap Gaussian pyrDown), Scharr gradients, at most 30 iterations or
                   |step| < 0.01 px, minEigThreshold 1e-4, points that leave the image are dropped
  * transform    : RANSAC over 2-point similarity hypotheses (reprojection threshold 3 px, confidence 0.99, at most 2000
                   draws) + least-squares refit on the inliers

PARITY UNPINNED against OpenCV (absent here; the reference holds no fixtures for this path): float64 arithmetic instead of
OpenCV's fixed point inside the LK loop and numpy's generator instead of cv::RNG in RANSAC, so the estimate agrees with
OpenCV's to sub-pixel noise, not bit for bit.  Pinned by known-answer tests on synthetic frame pairs (tests/test_gmc.py).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

MAX_CORNERS, QUALITY_LEVEL, BLOCK_SIZE = 1000, 0.01, 3
LK_WIN, LK_LEVELS, LK_MAX_ITERS, LK_EPS, LK_MIN_EIG = 21, 3, 30, 0.01, 1e-4
RANSAC_THRESHOLD, RANSAC_CONFIDENCE, RANSAC_MAX_ITERS = 3.0, 0.99, 2000


# ------------------------------------------------------------------------------------------------- image preparation
def bgr_to_gray(frame: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(COLOR_BGR2GRAY) on uint8: 14-bit fixed-point coefficients, round to nearest."""
    f = frame.astype(np.int32)
    return ((f[..., 0] * 1868 + f[..., 1] * 9617 + f[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)


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


def resize_linear(gray: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv2.resize(INTER_LINEAR) of a uint8 plane (two 11-bit fixed-point passes, as the engine's letterbox kernel does)."""
    sh, sw = gray.shape
    xi, xa0, xa1 = _linear_coeffs(dw, sw)
    yi, yb0, yb1 = _linear_coeffs(dh, sh)
    src = gray.astype(np.int64)
    hor = src[:, xi] * xa0[None, :] + src[:, np.minimum(xi + 1, sw - 1)] * xa1[None, :]
    s0, s1 = hor[yi], hor[np.minimum(yi + 1, sh - 1)]
    out = (((yb0[:, None] * (s0 >> 4)) >> 16) + ((yb1[:, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def _pad101(a: np.ndarray, p: int) -> np.ndarray:
    return np.pad(a, p, mode="reflect")                      # BORDER_REFLECT_101


# ------------------------------------------------------------------------------------------------- Shi-Tomasi corners
def good_features_to_track(gray: np.ndarray, max_corners: int = MAX_CORNERS, quality: float = QUALITY_LEVEL,
                           block: int = BLOCK_SIZE) -> np.ndarray:
    """-> float32 [n, 2] (x, y), strongest corner first."""
    g = _pad101(gray.astype(np.float64), 1)
    # 3x3 Sobel, scaled as cornerMinEigenVal does for 8-bit input: 1 / (2^(ksize-1) * block * 255)
    sc = 1.0 / (4.0 * block * 255.0)
    dx = ((g[:-2, 2:] - g[:-2, :-2]) + 2 * (g[1:-1, 2:] - g[1:-1, :-2]) + (g[2:, 2:] - g[2:, :-2])) * sc
    dy = ((g[2:, :-2] - g[:-2, :-2]) + 2 * (g[2:, 1:-1] - g[:-2, 1:-1]) + (g[2:, 2:] - g[:-2, 2:])) * sc

    def box(a):                                              # unnormalised block x block box filter
        p = _pad101(a, block // 2)
        h, w = a.shape
        return sum(p[i:i + h, j:j + w] for i in range(block) for j in range(block))

    a, b, c = box(dx * dx) * 0.5, box(dx * dy), box(dy * dy) * 0.5
    eig = (a + c) - np.sqrt((a - c) * (a - c) + b * b)
    eig = eig.astype(np.float32)
    mx = float(eig.max()) if eig.size else 0.0
    if mx <= 0:
        return np.zeros((0, 2), np.float32)
    eig = np.where(eig > mx * quality, eig, np.float32(0))   # THRESH_TOZERO
    p = np.pad(eig, 1, mode="constant", constant_values=-np.inf)
    h, w = eig.shape
    dil = np.max([p[i:i + h, j:j + w] for i in range(3) for j in range(3)], axis=0)
    ok = (eig != 0) & (eig == dil)
    ok[0, :] = ok[-1, :] = False
    ok[:, 0] = ok[:, -1] = False
    ys, xs = np.nonzero(ok)
    order = np.argsort(-eig[ys, xs], kind="stable")[:max_corners]
    return np.stack([xs[order], ys[order]], axis=1).astype(np.float32)


def prepare_frame(raw_frame: np.ndarray, downscale: int = 2, max_corners: int = MAX_CORNERS, quality: float = QUALITY_LEVEL
                  ) -> Tuple[np.ndarray, np.ndarray]:
    """The first three OpenCV calls of ``GMC.apply_sparseoptflow`` for one BGR frame -> (gray plane at 1 / downscale, corners
    float32 [n, 2] strongest first)."""
    h, w = raw_frame.shape[:2]
    gray = bgr_to_gray(raw_frame) if raw_frame.ndim == 3 else raw_frame
    if downscale > 1:
        gray = resize_linear(gray, w // downscale, h // downscale)
    return gray, good_features_to_track(gray, max_corners, quality)


# ------------------------------------------------------------------------------------------------- pyramidal Lucas-Kanade
def _pyr_down(img: np.ndarray) -> np.ndarray:
    """cv2.pyrDown on uint8: separable [1 4 6 4 1] / 16, reflect-101 borders, every second pixel, round to nearest."""
    k = np.array([1, 4, 6, 4, 1], dtype=np.int64)
    p = _pad101(img.astype(np.int64), 2)
    h, w = img.shape
    rows = sum(k[i] * p[:, i:i + w] for i in range(5))[:, ::2]
    out = sum(k[i] * rows[i:i + h] for i in range(5))[::2]
    return ((out + 128) >> 8).astype(np.uint8)


def _scharr(img: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    g = _pad101(img.astype(np.float64), 1)
    dx = 3 * (g[:-2, 2:] - g[:-2, :-2]) + 10 * (g[1:-1, 2:] - g[1:-1, :-2]) + 3 * (g[2:, 2:] - g[2:, :-2])
    dy = 3 * (g[2:, :-2] - g[:-2, :-2]) + 10 * (g[2:, 1:-1] - g[:-2, 1:-1]) + 3 * (g[2:, 2:] - g[:-2, 2:])
    return dx, dy


def _patches(img_pad: np.ndarray, pts: np.ndarray, pad: int, win: int) -> np.ndarray:
    """Bilinear win x win patches whose top-left corner is pts - win // 2 (pts float64 [n, 2] in unpadded coordinates)."""
    half = win // 2
    x = pts[:, 0] - half + pad
    y = pts[:, 1] - half + pad
    ix, iy = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    ax, ay = (x - ix)[:, None, None], (y - iy)[:, None, None]
    jj = np.arange(win)
    yy = iy[:, None, None] + jj[None, :, None]
    xx = ix[:, None, None] + jj[None, None, :]
    i00, i01, i10, i11 = img_pad[yy, xx], img_pad[yy, xx + 1], img_pad[yy + 1, xx], img_pad[yy + 1, xx + 1]
    return (1 - ay) * ((1 - ax) * i00 + ax * i01) + ay * ((1 - ax) * i10 + ax * i11)


def calc_optical_flow_pyr_lk(prev: np.ndarray, cur: np.ndarray, pts: np.ndarray, win: int = LK_WIN, levels: int = LK_LEVELS,
                                   max_iters: int = LK_MAX_ITERS, eps: float = LK_EPS, min_eig: float = LK_MIN_EIG
                                   ) -> Tuple[np.ndarray, np.ndarray]:
    """Bouguet's pyramidal Lucas-Kanade tracker with OpenCV's defaults.  prev / cur: uint8 planes, pts float32 [n, 2] in prev.
    -> (next points float32 [n, 2], status bool [n])."""
    n = len(pts)
    if n == 0:
        return np.zeros((0, 2), np.float32), np.zeros((0,), bool)
    pyr_p, pyr_c = [prev], [cur]
    for _ in range(levels):
        nh, nw = (pyr_p[-1].shape[0] + 1) // 2, (pyr_p[-1].shape[1] + 1) // 2
        if nh <= win or nw <= win:                           # buildOpticalFlowPyramid stops at levels not larger than the window
            break
        pyr_p.append(_pyr_down(pyr_p[-1]))
        pyr_c.append(_pyr_down(pyr_c[-1]))
    top = len(pyr_p) - 1
    pad = win + 2
    half = win // 2
    status = np.ones(n, bool)
    nxt = np.zeros((n, 2), np.float64)
    p0 = pts.astype(np.float64)
    for lvl in range(top, -1, -1):
        ip, ic = pyr_p[lvl], pyr_c[lvl]
        h, w = ip.shape
        dx, dy = _scharr(ip)
        ipp, dxp, dyp, icp = (_pad101(a.astype(np.float64), pad) for a in (ip, dx, dy, ic))
        pl = p0 / (1 << lvl)
        nl = pl.copy() if lvl == top else nxt * 2.0
        # a point whose window's corner falls outside the (window-padded) image is dropped -- at level 0 for good
        tl = np.floor(pl - half)
        inside = (tl[:, 0] >= -win) & (tl[:, 0] < w) & (tl[:, 1] >= -win) & (tl[:, 1] < h)
        if lvl == 0:
            status &= inside
        act = np.nonzero(inside)[0]
        if len(act):
            plc = np.clip(pl[act], [-half, -half], [w - 1 + half, h - 1 + half])
            I = _patches(ipp, plc, pad, win)
            Ix = _patches(dxp, plc, pad, win)
            Iy = _patches(dyp, plc, pad, win)
            # OpenCV's scaling: gradients in Scharr units (32 x per-pixel slope), products times 2^-20
            s = 1.0 / (1 << 20)
            a11, a12, a22 = (Ix * Ix).sum((1, 2)) * s, (Ix * Iy).sum((1, 2)) * s, (Iy * Iy).sum((1, 2)) * s
            det = a11 * a22 - a12 * a12
            mineig = (a22 + a11 - np.sqrt((a11 - a22) ** 2 + 4 * a12 * a12)) / (2 * win * win)
            good = (mineig >= min_eig) & (det >= np.finfo(np.float32).eps)
            if lvl == 0:
                status[act[~good]] = False
            idx = act[good]
            I, Ix, Iy = I[good], Ix[good], Iy[good]
            a11, a12, a22, det = a11[good], a12[good], a22[good], det[good]
            cur_pts = nl[idx].copy()
            prev_delta = np.zeros((len(idx), 2))
            live = np.ones(len(idx), bool)
            for it in range(max_iters):
                if not live.any():
                    break
                li = np.nonzero(live)[0]
                q = cur_pts[li]
                tlq = np.floor(q - half)
                inq = (tlq[:, 0] >= -win) & (tlq[:, 0] < w) & (tlq[:, 1] >= -win) & (tlq[:, 1] < h)
                if lvl == 0:
                    status[idx[li[~inq]]] = False
                live[li[~inq]] = False
                li = li[inq]
                if not len(li):
                    break
                qc = np.clip(cur_pts[li], [-half, -half], [w - 1 + half, h - 1 + half])
                diff = (_patches(icp, qc, pad, win) - I[li]) * 32.0          # intensities carry 5 fractional bits in OpenCV
                b1, b2 = (diff * Ix[li]).sum((1, 2)) * s, (diff * Iy[li]).sum((1, 2)) * s
                dxy = np.stack([(a12[li] * b2 - a22[li] * b1) / det[li], (a12[li] * b1 - a11[li] * b2) / det[li]], axis=1)
                cur_pts[li] += dxy
                done = (dxy * dxy).sum(1) <= eps * eps
                if it > 0:
                    osc = (np.abs(dxy + prev_delta[li]) < 0.01).all(1) & ~done
                    cur_pts[li[osc]] -= dxy[osc] * 0.5
                    done |= osc
                prev_delta[li] = dxy
                live[li[done]] = False
            nl[idx] = cur_pts
        nxt = nl
    if status.any():
        h, w = prev.shape
        out = (nxt[:, 0] < 0) | (nxt[:, 1] < 0) | (nxt[:, 0] >= w) | (nxt[:, 1] >= h)
        status &= ~out
    return nxt.astype(np.float32), status


# ------------------------------------------------------------------------------------------------- partial affine by RANSAC
def _similarity_from_pairs(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    """Least-squares [[a, -b, tx], [b, a, ty]] mapping p -> q (exact for two pairs)."""
    pm, qm = p.mean(0), q.mean(0)
    pc, qc = p - pm, q - qm
    den = (pc * pc).sum()
    if den <= 0:
        return np.array([[1.0, 0.0, qm[0] - pm[0]], [0.0, 1.0, qm[1] - pm[1]]])
    a = (pc * qc).sum() / den
    b = (pc[:, 0] * qc[:, 1] - pc[:, 1] * qc[:, 0]).sum() / den
    return np.array([[a, -b, qm[0] - (a * pm[0] - b * pm[1])], [b, a, qm[1] - (b * pm[0] + a * pm[1])]])


def _same_point(a, b) -> bool:
    """np.allclose(a, b) for two 2-vectors (rtol 1e-5, atol 1e-8) without its array machinery (30 us a call, twice per draw)"""
    ax, ay, bx, by = float(a[0]), float(a[1]), float(b[0]), float(b[1])
    return abs(ax - bx) <= 1e-8 + 1e-5 * abs(bx) and abs(ay - by) <= 1e-8 + 1e-5 * abs(by)


def estimate_affine_partial_2d(src: np.ndarray, dst: np.ndarray, threshold: float = RANSAC_THRESHOLD, confidence: float = RANSAC_CONFIDENCE,
                               max_iters: int = RANSAC_MAX_ITERS, seed: int = 0) -> Tuple[Optional[np.ndarray], np.ndarray]:
    """cv2.estimateAffinePartial2D(src, dst, RANSAC): 4-degree-of-freedom similarity + inlier mask, or (None, zeros)."""
    src, dst = np.asarray(src, np.float64).reshape(-1, 2), np.asarray(dst, np.float64).reshape(-1, 2)
    n = len(src)
    if n < 2:
        return None, np.zeros(n, bool)
    rng = np.random.default_rng(seed)
    best_mask, best_count = None, 0
    iters, it = max_iters, 0
    thr2 = threshold * threshold
    while it < iters:
        it += 1
        i, j = rng.choice(n, 2, replace=False)
        if _same_point(src[i], src[j]) or _same_point(dst[i], dst[j]):
            continue
        H = _similarity_from_pairs(src[[i, j]], dst[[i, j]])
        err = ((src @ H[:, :2].T + H[:, 2] - dst) ** 2).sum(1)
        mask = err <= thr2
        cnt = int(mask.sum())
        if cnt > max(best_count, 1):
            best_mask, best_count = mask, cnt
            # RANSAC's adaptive stopping rule: draws needed to see an all-inlier sample with the asked confidence
            w = cnt / n
            denom = np.log(max(1.0 - w * w, 1e-12))
            iters = min(iters, int(np.ceil(np.log(1.0 - confidence) / denom))) if denom < 0 else it
    if best_mask is None or best_count < 2:
        return None, np.zeros(n, bool)
    H = _similarity_from_pairs(src[best_mask], dst[best_mask])          # refit on the consensus set
    return H, best_mask


class GMC:
    """``GMC(method="sparseOptFlow", downscale=2).apply(frame_bgr) -> 2x3`` (float64), all numpy; identity on the first frame, when
    too few points survive, or when ``method`` is None / "none"."""

    def __init__(self, method: Optional[str] = "sparseOptFlow", downscale: int = 2):
        if method in ("none", "None"):
            method = None
        if method not in (None, "sparseOptFlow"):
            raise ValueError(f"GMC method {method!r} is not implemented (sparseOptFlow, the botsort.yaml default, or None)")
        self.method, self.downscale = method, max(1, int(downscale))
        self.prev_frame: Optional[np.ndarray] = None
        self.prev_points: Optional[np.ndarray] = None

    def reset(self) -> None:
        self.prev_frame = self.prev_points = None

    def apply(self, raw_frame: np.ndarray, detections=None) -> np.ndarray:
        H = np.eye(2, 3)
        if self.method is None or raw_frame is None:
            return H
        frame, points = prepare_frame(raw_frame, self.downscale)
        if self.prev_frame is None or self.prev_points is None or self.prev_frame.shape != frame.shape:
            self.prev_frame, self.prev_points = frame.copy(), points
            return H
        nxt, status = calc_optical_flow_pyr_lk(self.prev_frame, frame, self.prev_points)
        p, q = self.prev_points[status], nxt[status]
        if len(p) > 4:
            est, _ = estimate_affine_partial_2d(p, q)
            if est is not None:
                H = est
                H[0, 2] *= self.downscale
                H[1, 2] *= self.downscale
        self.prev_frame, self.prev_points = frame.copy(), points
        return H


def warp_kalman(mean: np.ndarray, cov: np.ndarray, H: np.ndarray, R8: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """``STrack.multi_gmc`` for one track: the rotation/scale block acts on every (x, y)-like pair of the state
    (cx cy | w h | vcx vcy | vw vh), the translation on the centre only; P <- R8 P R8'.  ``R8`` = kron(I4, H[:2, :2]) may be
    handed in by a caller that warps many tracks with one H."""
    if R8 is None:
        R8 = np.kron(np.eye(4), H[:2, :2])
    m = R8 @ mean
    m[:2] += H[:2, 2]
    return m, R8 @ cov @ R8.T
