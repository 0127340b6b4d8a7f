"""The product's tracker core (host C++, csrc/tracker_host.cpp, through the C ABI) against its checker (oracle/tracker_oracle.py, numpy)
on scripted and random clips -- /root/reference/model.py:38-46 (model.track -> boxes.id, boxes.xywhn).

Both are handed the SAME detections and the SAME camera-motion matrices, so everything that differs is the core itself: Kalman
predict / update / warp, the fp32 IoU cost and its score fusion, the Jonker-Volgenant assignment with lap.lapjv's cost_limit
semantics, the two-stage association and the track bookkeeping.  Asserted: the same rows in the same order -- ids, matched
detection index, score and class EXACTLY, box corners to float32 rounding (the C++ filter solves the 4x4 gain system by Cholesky, numpy
by LU: last-bit differences in float64 that a float32 row almost never shows; the bound is 1e-4 px and the exact-match fraction is
printed)."""
import numpy as np
import pytest

from cvsd_amd import tracker as P
from oracle import tracker_oracle as O


def _clip(seed, n_frames=200, n_people=6, w=640, h=480, camera=True):
    """-> list of (detections [n, 6] float32, warp 2x3 or None): people on random walks, missed detections, low-score frames,
    spurious boxes, people who leave and come back, and (camera=True) a camera that pans / rolls a little on some frames."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform([60, 60], [w - 60, h - 60], size=(n_people, 2))
    vel = rng.uniform(-5, 5, size=(n_people, 2))
    size = rng.uniform([24, 60], [60, 150], size=(n_people, 2))
    away = np.zeros(n_people, int)
    out = []
    for f in range(n_frames):
        warp = None
        if camera and f > 0 and rng.random() < 0.5:
            ang, sc = rng.normal(0, 0.004), 1 + rng.normal(0, 0.003)
            t = rng.normal(0, 6.0, size=2)
            warp = np.array([[sc * np.cos(ang), -sc * np.sin(ang), t[0]], [sc * np.sin(ang), sc * np.cos(ang), t[1]]])
            pos = pos @ warp[:, :2].T + warp[:, 2]
        vel += rng.normal(0, 0.6, size=vel.shape)
        pos = pos + vel
        det = []
        for p in range(n_people):
            if away[p] > 0:
                away[p] -= 1
                continue
            if rng.random() < 0.03:
                away[p] = int(rng.integers(3, 45))          # leaves the scene: shorter and longer than track_buffer
                continue
            if rng.random() < 0.08:
                continue                                    # missed detection
            score = rng.uniform(0.3, 0.95) if rng.random() > 0.15 else rng.uniform(0.05, 0.3)
            jit = rng.normal(0, 1.5, size=4)
            cx, cy = pos[p]
            bw, bh = size[p]
            det.append([cx - bw / 2 + jit[0], cy - bh / 2 + jit[1], cx + bw / 2 + jit[2], cy + bh / 2 + jit[3], score, 0])
        for _ in range(rng.poisson(0.4)):                   # spurious boxes
            x, y = rng.uniform(0, w - 40), rng.uniform(0, h - 80)
            det.append([x, y, x + rng.uniform(15, 50), y + rng.uniform(30, 100), rng.uniform(0.05, 0.6), 0])
        rng.shuffle(det)
        out.append((np.asarray(det, np.float32).reshape(-1, 6), warp))
    return out


def _run_both(clip):
    p, o = P.BYTETracker(gmc_method=None), O.BYTETracker(gmc_method=None)
    exact = total = 0
    for f, (det, warp) in enumerate(clip):
        rp, ro = p.update(det, warp=warp), o.update(det, warp=warp)
        assert rp.shape == ro.shape, f"frame {f}: {len(rp)} rows vs the oracle's {len(ro)}"
        np.testing.assert_array_equal(rp[:, 4:], ro[:, 4:], err_msg=f"frame {f}: id / score / cls / idx")
        np.testing.assert_allclose(rp[:, :4], ro[:, :4], rtol=0, atol=1e-4, err_msg=f"frame {f}: boxes")
        exact += int((rp[:, :4] == ro[:, :4]).sum())
        total += rp[:, :4].size
        assert p.frame_id == o.frame_id and p._ids_issued == o._ids_issued
        assert [t.track_id for t in p.lost_stracks] == [t.track_id for t in o.lost_stracks]
        assert [t.track_id for t in p.tracked_stracks] == [t.track_id for t in o.tracked_stracks]
    return exact, total, p._ids_issued


@pytest.mark.parametrize("seed", range(8))
def test_cpp_core_reproduces_the_numpy_tracker_on_random_clips(seed):
    exact, total, ids = _run_both(_clip(seed, camera=seed % 2 == 0, n_people=4 + seed))
    assert total > 400 and ids >= 4 + seed
    print(f"[tracker] clip {seed}: {ids} ids issued, {exact}/{total} box coordinates bit-identical in float32")
    assert exact >= 0.999 * total


def test_cpp_core_on_a_crowd_and_on_degenerate_frames():
    """40 people (assignment problems of 80 x 80 after lap's extension), then frames made of duplicates, zero-area and
    touching boxes, equal scores -- the cases where ties decide"""
    exact, total, ids = _run_both(_clip(100, n_frames=60, n_people=40, w=1920, h=1080))
    assert ids >= 40
    box = [100, 100, 160, 220]
    frames = [
        np.asarray([[*box, 0.9, 0], [*box, 0.9, 0], [*box, 0.9, 0]], np.float32),                 # three identical detections
        np.asarray([[*box, 0.9, 0], [*box, 0.9, 0]], np.float32),
        np.asarray([[100, 100, 100, 220, 0.9, 0], [*box, 0.5, 0], [160, 100, 220, 220, 0.5, 0]], np.float32),   # zero area, touching
        np.zeros((0, 6), np.float32),
        np.asarray([[*box, 0.25, 0], [*box, 0.1, 0], [*box, 0.2499, 0]], np.float32),             # the thresholds themselves
        np.asarray([[*box, 0.9, 0], [300, 100, 360, 220, 0.9, 1]], np.float32),
    ]
    _run_both([(d, None) for d in frames * 4])


def test_lapjv_is_laps_extended_problem():
    """mi355_lapjv(cost, limit) = lap.lapjv(cost, extend_cost=True, cost_limit=limit): (1) the C++ solver and the oracle's pure-Python
    statement return the SAME assignment on matrices full of ties; (2) the assignment is optimal for the extended problem (checked
    against scipy's Hungarian solver on the extended matrix); (3) hand cases: a pair above the limit stays unmatched even when
    Hungarian-then-filter would have paired (and then dropped) it and forced a worse match elsewhere."""
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(7)
    for t in range(400):
        nr, nc = int(rng.integers(0, 12)), int(rng.integers(0, 12))
        c = rng.random((nr, nc))
        if t % 2:
            c = np.round(c * 4) / 4                                  # many equal costs
        if t % 5 == 0:
            c[:] = 1.0                                               # no overlap anywhere: IoU cost 1 in every cell
        lim = float(rng.choice([0.5, 0.7, 0.8]))
        x, y = P.lapjv(c, lim)
        xo, yo = O.lapjv(c, lim)
        np.testing.assert_array_equal(x, xo)
        np.testing.assert_array_equal(y, yo)
        assert all(y[x[i]] == i for i in range(nr) if x[i] >= 0) and all(x[y[j]] == j for j in range(nc) if y[j] >= 0)
        if nr and nc:
            n = nr + nc
            ext = np.full((n, n), lim / 2)
            ext[nr:, nc:] = 0
            ext[:nr, :nc] = c
            ri, ci = linear_sum_assignment(ext)
            mine = sum(c[i, x[i]] for i in range(nr) if x[i] >= 0) + lim / 2 * ((x < 0).sum() + (y < 0).sum())
            assert mine == pytest.approx(ext[ri, ci].sum(), abs=1e-9)
            assert all(c[i, x[i]] <= lim for i in range(nr) if x[i] >= 0)      # a pair above the limit is never made
    # (3): rows = tracks, columns = detections, limit 0.8.  Hungarian's optimum is (0,1) + (1,0) = 0.85 + 0.1; filtering drops (0,1).
    # lap's extended problem leaves track 0 unmatched from the start -- the same here, but the two differ on the next case
    x, y = P.lapjv(np.array([[0.5, 0.85], [0.1, 0.9]]), 0.8)
    assert x.tolist() == [-1, 0] and y.tolist() == [1, -1]
    # Hungarian: (0,0) + (1,1) = 0.3 + 0.95 = 1.25 beats (0,1) + (1,0) = 0.9 + 0.4 = 1.3 -> after the filter only (0,0) is left.
    # Extended problem: (1,0) + both others unmatched = 0.4 + 0.8 = 1.2 vs (0,0) + unmatched = 0.3 + 0.8 = 1.1 -> (0,0) as well;
    x, y = P.lapjv(np.array([[0.3, 0.9], [0.4, 0.95]]), 0.8)
    assert x.tolist() == [0, -1]
    # ... while here Hungarian pairs (0,1) + (1,0) = 0.7 + 0.2 = 0.9 < (0,0) + (1,1) = 0.1 + 0.85 = 0.95 and keeps both pairs,
    # and the extended problem agrees (0.9 < 0.1 + 0.8): two matches
    x, y = P.lapjv(np.array([[0.1, 0.7], [0.2, 0.85]]), 0.8)
    assert x.tolist() == [1, 0]
    with pytest.raises(ValueError):
        P.lapjv(np.zeros(3), 0.8)


def test_kalman_pieces_agree_with_the_numpy_filter():
    rng = np.random.default_rng(3)
    for _ in range(50):
        z = rng.uniform([0, 0, 10, 20], [1000, 800, 200, 400])
        mp, cp = P.KalmanFilterXYWH.initiate(z)
        mo, co = O.KalmanFilterXYWH.initiate(z)
        for step in range(6):
            np.testing.assert_allclose(mp, mo, rtol=1e-12, atol=1e-10)
            np.testing.assert_allclose(cp, co, rtol=1e-11, atol=1e-10)
            mp, cp = P.KalmanFilterXYWH.predict(mp, cp)
            mo, co = O.KalmanFilterXYWH.predict(mo, co)
            if step % 2 == 0:
                H = np.array([[1.01, -0.02, 3.0], [0.02, 1.01, -1.5]])
                from oracle.gmc_oracle import warp_kalman
                mp, cp = P.warp_kalman(mp, cp, H)
                mo, co = warp_kalman(mo, co, H)
            z2 = mo[:4] + rng.normal(0, 2, 4)
            mp, cp = P.KalmanFilterXYWH.update(mp, cp, z2)
            mo, co = O.KalmanFilterXYWH.update(mo, co, z2)
        assert np.allclose(cp, cp.T, atol=1e-9)
