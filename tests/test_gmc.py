"""Known-answer tests of BoT-SORT's global motion compensation (``gmc_method: sparseOptFlow``, the Ultralytics default behind
``/root/reference/model.py:38``).  OpenCV is not available, so every stage is pinned against constructions whose answer is known:
hand-computed luma values, corners of drawn squares, analytically shifted / rotated images, point sets with planted outliers --
and, end to end, a panning camera over static people.  Two implementations go through them: the numpy restatement
(oracle/gmc_oracle.py = ``G``, the checker) and the product's host C++ (cvsd_amd/gmc.py = ``gmc``, csrc/gmc_host.cpp), which
must also equal the restatement stage by stage (plane and corner list bit for bit, Lucas-Kanade points to 1e-3 px)."""
import numpy as np
import pytest

from cvsd_amd import gmc
from oracle import gmc_oracle as G
from cvsd_amd.tracker import BYTETracker, KalmanFilterXYWH


def _texture(h, w, seed=0, smooth=2):
    """A band-limited random texture (sum of random sinusoids) as a FUNCTION of continuous coordinates."""
    rng = np.random.default_rng(seed)
    k = rng.uniform(-0.45, 0.45, size=(40, 2))
    ph = rng.uniform(0, 2 * np.pi, size=40)
    amp = rng.uniform(0.5, 1.0, size=40)

    def f(x, y):
        v = sum(a * np.sin(kx * x + ky * y + p) for a, (kx, ky), p in zip(amp, k, ph))
        return np.clip(127.5 + v * 18.0, 0, 255)
    return f


def _render(f, h, w, H=None):
    """image whose pixel (x, y) shows the texture at the PRE-image of (x, y) under the 2x3 transform H (background moved by H)"""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    if H is not None:
        A = np.vstack([H, [0, 0, 1]])
        Ai = np.linalg.inv(A)
        xs, ys = Ai[0, 0] * xs + Ai[0, 1] * ys + Ai[0, 2], Ai[1, 0] * xs + Ai[1, 1] * ys + Ai[1, 2]
    return np.rint(f(xs, ys)).astype(np.uint8)


def test_bgr_to_gray_is_opencvs_fixed_point_luma():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 100], [255, 255, 255]]], dtype=np.uint8)    # B, G, R
    # (1868 B + 9617 G + 4899 R + 8192) >> 14
    assert G.bgr_to_gray(px).tolist() == [[29, 150, 76, (1868 * 10 + 9617 * 200 + 4899 * 100 + 8192) >> 14, 255]]


def test_half_size_linear_resize_is_the_2x2_block_mean():
    rng = np.random.default_rng(1)
    g = rng.integers(0, 256, size=(12, 16), dtype=np.uint8)
    want = (g.reshape(6, 2, 8, 2).astype(np.int64).sum((1, 3)) + 2) >> 2
    np.testing.assert_array_equal(G.resize_linear(g, 8, 6), want)


def test_shi_tomasi_finds_the_corners_of_drawn_squares_strongest_first():
    img = np.full((60, 80), 20, np.uint8)
    img[10:30, 15:40] = 220          # bright rectangle: corners near (15,10) (39,10) (15,29) (39,29)
    img[40:52, 50:70] = 120          # a weaker one
    pts = G.good_features_to_track(img)
    assert 8 <= len(pts) <= 64
    strong = {(15, 10), (39, 10), (15, 29), (39, 29)}
    for x, y in pts[:4]:             # the four strongest are the bright rectangle's corners (within the 3x3 block)
        assert min(abs(x - cx) + abs(y - cy) for cx, cy in strong) <= 2
    found = [min(np.abs(pts - np.array(c)).sum(1)) for c in [(50, 40), (69, 40), (50, 51), (69, 51)]]
    assert max(found) <= 2           # the weaker rectangle's corners pass the 1 % quality level too
    assert (pts[:, 0] >= 1).all() and (pts[:, 0] <= 78).all()
    assert len(G.good_features_to_track(np.full((40, 40), 7, np.uint8))) == 0      # flat image: no corner


def _smooth_noise(h, w, seed, sigma=2.0):
    """non-periodic texture: white noise blurred by a separable Gaussian, stretched to 8 bits"""
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(h, w))
    r = int(3 * sigma)
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
    k /= k.sum()
    a = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 1, a)
    a = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 0, a)
    a = (a - a.min()) / (a.max() - a.min())
    return np.rint(a * 255).astype(np.uint8)


@pytest.mark.parametrize("shift", [(3.0, -2.0), (1.3, 0.7), (-6.5, 4.25)])
def test_pyramidal_lk_recovers_a_known_subpixel_translation(shift):
    """an analytic texture sampled at shifted coordinates: the true displacement is known to any precision"""
    f = _texture(120, 160, seed=3)
    prev = _render(f, 120, 160)
    cur = _render(f, 120, 160, np.array([[1, 0, shift[0]], [0, 1, shift[1]]], float))
    pts = G.good_features_to_track(prev)
    pts = pts[(pts[:, 0] > 20) & (pts[:, 0] < 140) & (pts[:, 1] > 20) & (pts[:, 1] < 100)][:200]
    nxt, ok = gmc.calc_optical_flow_pyr_lk(prev, cur, pts)
    assert ok.mean() > 0.9
    err = np.abs((nxt - pts)[ok] - np.array(shift))
    assert np.median(err) < 0.05 and np.quantile(err, 0.9) < 0.1


@pytest.mark.parametrize("shift", [(11, 9), (-17, 6), (24, -13)])
def test_pyramidal_lk_follows_large_integer_shifts_through_the_pyramid(shift):
    """two crops of one non-periodic image: displacements far beyond a 21-pixel window's reach at full resolution are found
    on the coarse levels (4 levels: the coarsest sees an eighth of the shift)"""
    # fine + coarse detail: the coarse levels of the pyramid need structure that survives three 2x reductions
    big = ((_smooth_noise(300, 400, seed=7, sigma=2.0).astype(np.int32) + _smooth_noise(300, 400, seed=8, sigma=8.0)) // 2).astype(np.uint8)
    y0, x0 = 60, 80
    prev = big[y0:y0 + 180, x0:x0 + 240]
    cur = big[y0 - shift[1]:y0 - shift[1] + 180, x0 - shift[0]:x0 - shift[0] + 240]        # content moves by +shift
    pts = G.good_features_to_track(prev)
    pts = pts[(pts[:, 0] > 40) & (pts[:, 0] < 200) & (pts[:, 1] > 40) & (pts[:, 1] < 140)][:150]
    nxt, ok = gmc.calc_optical_flow_pyr_lk(prev, cur, pts)
    assert ok.mean() > 0.8              # windows on weak structure fail the minimum-eigenvalue test and are dropped
    err = np.abs((nxt - pts)[ok] - np.array(shift, float))
    assert np.median(err) < 0.02 and np.quantile(err, 0.8) < 0.1


def test_partial_affine_ransac_ignores_outliers_and_refits_on_inliers():
    rng = np.random.default_rng(5)
    src = rng.uniform(0, 300, size=(120, 2))
    ang, sc, t = np.deg2rad(4.0), 1.03, np.array([7.5, -3.25])
    R = sc * np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    dst = src @ R.T + t
    dst[:30] += rng.uniform(20, 60, size=(30, 2))            # a quarter of the matches are wrong (moving people)
    H, inl = G.estimate_affine_partial_2d(src, dst)
    np.testing.assert_allclose(H, np.hstack([R, t[:, None]]), atol=1e-9)
    assert inl[30:].all() and not inl[:30].any()
    noisy = dst + rng.normal(0, 0.2, size=dst.shape)
    H2, _ = G.estimate_affine_partial_2d(src, noisy)
    np.testing.assert_allclose(H2, np.hstack([R, t[:, None]]), atol=0.05)
    assert G.estimate_affine_partial_2d(src[:1], dst[:1])[0] is None


@pytest.mark.parametrize("H", [
    np.array([[1, 0, 8.0], [0, 1, -6.0]]),                                                   # pan
    np.array([[np.cos(0.02), -np.sin(0.02), 3.0], [np.sin(0.02), np.cos(0.02), 1.5]]),       # pan + 1.1 degree roll
    np.array([[1.02, 0, -2.0], [0, 1.02, 4.0]]),                                             # zoom
])
@pytest.mark.parametrize("impl", ["product host C++", "oracle numpy"])
def test_gmc_recovers_the_background_transform_within_a_tenth_of_a_pixel(H, impl):
    """frame pair: the texture moved by H (full-resolution pixels); GMC works at half resolution and scales the translation
    back, as Ultralytics' GMC does (downscale 2).  Error measured as the displacement error over the frame's corners."""
    h, w = 240, 320
    f = _texture(h, w, seed=11)
    g0 = _render(f, h, w)
    g1 = _render(f, h, w, H)
    bgr = lambda g: np.stack([g, g, g], -1)
    m = gmc.GMC() if impl.startswith("product") else G.GMC()
    np.testing.assert_array_equal(m.apply(bgr(g0)), np.eye(2, 3))          # first frame: identity
    got = m.apply(bgr(g1))
    corners = np.array([[0, 0], [w, 0], [0, h], [w, h], [w / 2, h / 2]], float)
    # the estimate lives on the half-size grid: pixel centres map as x_half = (x_full + 0.5) / 2 - 0.5, which leaves a pure
    # translation unchanged after the x2 rescale and a rotation / zoom about a slightly shifted origin -- compare displacements
    want = corners @ H[:, :2].T + H[:, 2]
    have = corners @ got[:, :2].T + got[:, 2]
    # Ultralytics rescales only the translation (H[0,2], H[1,2] *= downscale); the linear part is resolution-independent
    pure_translation = np.array_equal(H[:, :2], np.eye(2))
    # a 21x21 Lucas-Kanade window models translation only: under roll / zoom its estimate carries a small bias, which shows at
    # the frame's far corners (a 7e-4 error of the linear part is 0.2 px at 320 px); a pan is recovered to well under 0.1 px
    assert np.abs(have - want).max() < (0.1 if pure_translation else 0.3), (got, H)
    np.testing.assert_allclose(got[:, :2], H[:, :2], atol=1e-3)
    np.testing.assert_allclose(got[:, 2], H[:, 2], atol=0.1 if pure_translation else 0.25)


def test_warp_of_the_kalman_state_is_strack_multi_gmc():
    mean = np.array([100.0, 50.0, 20.0, 40.0, 1.0, -2.0, 0.5, 0.25])
    cov = np.diag(np.arange(1.0, 9.0))
    H = np.array([[0.0, -1.0, 5.0], [1.0, 0.0, 7.0]])        # quarter turn + translation
    m, c = gmc.warp_kalman(mean, cov, H)
    np.testing.assert_allclose(m, [-50 + 5, 100 + 7, -40, 20, 2, 1, -0.25, 0.5])
    np.testing.assert_allclose(np.diag(c), [2, 1, 4, 3, 6, 5, 8, 7])      # each (x, y) pair's variances swap under the quarter turn
    m2, c2 = gmc.warp_kalman(mean, cov, np.eye(2, 3))
    np.testing.assert_array_equal(m2, mean)
    np.testing.assert_array_equal(c2, cov)


def test_ids_survive_a_panning_camera_with_gmc_and_break_without():
    """Static people filmed by a camera that pans 26 px per frame: every box jumps by more than its own width (24 px), so the
    constant-velocity prediction of a freshly born track (velocity 0) has IoU 0 with the next detection.  With the frame handed
    to update() the background motion moves the predicted boxes along and the ids persist from the first frame on; without
    it the tracker loses every track on frame 2 and issues new ids."""
    h, w, pan = 240, 640, 26
    scene = ((_smooth_noise(h, w + 400, seed=21, sigma=2.0).astype(np.int32) + _smooth_noise(h, w + 400, seed=22, sigma=8.0)) // 2).astype(np.uint8)
    people = [(60.0, 60.0), (200.0, 120.0), (330.0, 80.0)]           # box centres at frame 0, 24 x 60 px boxes

    def frame_and_dets(k):
        g = scene[:, pan * k:pan * k + w]                            # camera moved right by pan*k: the scene moves left
        det = [[cx - pan * k - 12, cy - 30, cx - pan * k + 12, cy + 30, 0.9, 0] for cx, cy in people]
        return np.stack([g, g, g], -1), np.asarray(det, np.float32)

    with_gmc, without = BYTETracker(), BYTETracker(gmc_method=None)
    ids_a, ids_b = [], []
    for k in range(5):
        img, det = frame_and_dets(k)
        ra = with_gmc.update(det, img)
        rb = without.update(det, img)
        ids_a.append(sorted(ra[:, 4].astype(int).tolist()))
        ids_b.append(sorted(rb[:, 4].astype(int).tolist()))
    assert ids_a == [[1, 2, 3]] * 5, ids_a
    assert ids_b[0] == [1, 2, 3] and all(set(x).isdisjoint({1, 2, 3}) for x in ids_b[2:]), ids_b
    # the compensated boxes sit on the detections (the Kalman update then has a zero-velocity innovation)
    last = with_gmc.update(frame_and_dets(5)[1], frame_and_dets(5)[0])
    want = np.array([[cx - pan * 5, cy] for cx, cy in people])
    got = np.stack([(last[:, 0] + last[:, 2]) / 2, (last[:, 1] + last[:, 3]) / 2], 1)
    assert np.abs(np.sort(got, 0) - np.sort(want, 0)).max() < 1.0


def test_update_without_a_frame_is_the_identity_path():
    t1, t2 = BYTETracker(), BYTETracker(gmc_method=None)
    det = np.asarray([[10, 10, 50, 90, 0.9, 0]], np.float32)
    for _ in range(3):
        np.testing.assert_array_equal(t1.update(det), t2.update(det))
    with pytest.raises(ValueError):
        BYTETracker(gmc_method="orb")


@pytest.mark.parametrize("shape,downscale", [((240, 320), 2), ((241, 323), 2), ((96, 128), 1), ((150, 100), 3), ((2, 40), 2), ((40, 3), 2)])
def test_host_cpp_frame_preparation_is_the_numpy_statement_bit_for_bit(shape, downscale):
    """csrc/gmc_host.cpp:mi355_gmc_prepare_host (what GMC(device=None) runs) against oracle/gmc_oracle.py: the gray plane byte for
    byte, the corner list element for element -- odd sizes, no resize, a 1/3 scale, and planes with a 1-pixel dimension (reflect-101
    of a length-1 axis: the guard the advisor asked for, on the host side)."""
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    h, w = shape
    base = _smooth_noise(h + 8, w + 8, seed=h + w, sigma=1.5)[4:4 + h, 4:4 + w]
    frame = np.stack([base, np.roll(base, 1, 0), np.roll(base, 1, 1)], -1).astype(np.uint8)
    frame[rng.integers(0, h, 20), rng.integers(0, w, 20)] = rng.integers(0, 256, (20, 3))
    gray_o, pts_o = G.prepare_frame(frame, downscale)
    gray_p, pts_p = gmc.prepare_frame(frame, downscale, None)
    np.testing.assert_array_equal(gray_p, gray_o)
    np.testing.assert_array_equal(pts_p, pts_o)
    flat = np.full((h, w, 3), 9, np.uint8)
    assert gmc.prepare_frame(flat, downscale, None)[1].shape == (0, 2)


def test_gmc_objects_of_both_implementations_walk_the_same_states():
    """the product's host object (state machine in C++: mi355_gmc_track_*) and the numpy GMC over a short panning clip: the same
    previous plane and corner list after every frame; transforms equal up to the RANSAC generators (translation < 0.05 px); a frame
    of another size resets both; begin() of one frame followed by apply() of another drops the stale step."""
    scene = ((_smooth_noise(200, 500, seed=31, sigma=2.0).astype(np.int32) + _smooth_noise(200, 500, seed=32, sigma=8.0)) // 2).astype(np.uint8)
    frames = [np.stack([scene[:, 7 * k:7 * k + 320]] * 3, -1) for k in range(5)]
    a, b = gmc.GMC(), G.GMC()
    assert a.prev_frame is None and a.prev_points is None
    for k, f in enumerate(frames):
        Ha, Hb = a.apply(f), b.apply(f)
        np.testing.assert_array_equal(a.prev_frame, b.prev_frame)
        np.testing.assert_array_equal(a.prev_points, b.prev_points)
        if k == 0:
            np.testing.assert_array_equal(Ha, np.eye(2, 3))
        else:
            np.testing.assert_allclose(Ha, Hb, atol=0.05)
            assert abs(Ha[0, 2] + 7.0) < 0.2 and abs(Ha[1, 2]) < 0.2
    small = np.ascontiguousarray(frames[0][:100, :150])
    np.testing.assert_array_equal(a.apply(small), np.eye(2, 3))
    np.testing.assert_array_equal(b.apply(small), np.eye(2, 3))
    assert a.prev_frame.shape == (50, 75)
    a.begin(frames[1])                                          # enqueued for one frame ...
    np.testing.assert_array_equal(a.apply(frames[2]), np.eye(2, 3))      # ... collected for another: stale, dropped, state reset
    assert abs(a.apply(frames[3])[0, 2] + 7.0) < 0.2
    a.reset()
    assert a.prev_frame is None


def test_host_cpp_lucas_kanade_is_the_numpy_statement_of_the_algorithm():
    """the product runs csrc/gmc_host.cpp; oracle/gmc_oracle.py states the same algorithm in numpy -- same points kept, same
    positions to float32 rounding"""
    big = ((_smooth_noise(200, 260, seed=3, sigma=2.0).astype(np.int32) + _smooth_noise(200, 260, seed=4, sigma=8.0)) // 2).astype(np.uint8)
    prev, cur = big[20:170, 30:230], big[23:173, 25:225]
    pts = G.good_features_to_track(prev)[:300]
    a, sa = gmc.calc_optical_flow_pyr_lk(prev, cur, pts)
    b, sb = G.calc_optical_flow_pyr_lk(prev, cur, pts)
    assert (sa == sb).mean() > 0.99
    both = sa & sb
    assert both.sum() > 100 and np.abs(a[both] - b[both]).max() < 1e-3
    assert gmc.calc_optical_flow_pyr_lk(prev, cur, np.zeros((0, 2), np.float32))[0].shape == (0, 2)


def test_host_cpp_corner_ordering_and_ransac_are_the_numpy_statements():
    """csrc/gmc_host.cpp: the ordering of the kept corners (ties in raster order) and the RANSAC similarity, against the numpy forms"""
    rng = np.random.default_rng(4)
    eig = rng.integers(0, 12, size=(37, 53)).astype(np.float32) * np.float32(0.25)          # many ties
    ok = (rng.random((37, 53)) < 0.3).astype(np.uint8)
    ys, xs = np.nonzero(ok)
    order = np.argsort(-eig[ys, xs], kind="stable")
    want = np.stack([xs[order], ys[order]], axis=1).astype(np.float32)
    np.testing.assert_array_equal(gmc.order_corners(eig, ok, 10 ** 6), want)
    np.testing.assert_array_equal(gmc.order_corners(eig, ok, 50), want[:50])
    assert gmc.order_corners(eig, np.zeros_like(ok), 50).shape == (0, 2)
    # RANSAC: clean points -> the exact similarity; a quarter of planted outliers -> the planted inlier set and the refit on it
    ang, sc = 0.03, 1.02
    A = np.array([[sc * np.cos(ang), -sc * np.sin(ang), 4.5], [sc * np.sin(ang), sc * np.cos(ang), -2.25]])
    p = rng.uniform(0, 300, size=(200, 2))
    q = p @ A[:, :2].T + A[:, 2]
    H, m = gmc.estimate_affine_partial_2d_host(p, q)
    assert m.all() and np.abs(H - A).max() < 1e-9
    q2 = q.copy()
    bad = rng.choice(200, 50, replace=False)
    q2[bad] += rng.uniform(20, 60, size=(50, 2)) * rng.choice([-1, 1], size=(50, 2))
    H2, m2 = gmc.estimate_affine_partial_2d_host(p, q2)
    Hn, mn = G.estimate_affine_partial_2d(p, q2)
    good = np.ones(200, bool); good[bad] = False
    np.testing.assert_array_equal(m2, good)
    np.testing.assert_array_equal(mn, good)
    assert np.abs(H2 - Hn).max() < 1e-9 and np.abs(H2 - A).max() < 1e-9
    assert gmc.estimate_affine_partial_2d_host(p[:1], q[:1])[0] is None
    # all points identical: no hypothesis survives
    assert gmc.estimate_affine_partial_2d_host(np.zeros((10, 2)), np.zeros((10, 2)))[0] is None


def test_device_path_without_a_gpu_is_a_loud_error_not_a_fallback():
    """GMC(device=k) never falls back to the host routines: without a usable GPU the step raises (and BYTETracker.update lets that
    RuntimeError through -- only degenerate-geometry errors are bypassed, as Ultralytics' tracker does)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    frame = np.zeros((48, 64, 3), np.uint8)
    with pytest.raises(RuntimeError):
        gmc.GMC(device=0).apply(frame)
    with pytest.raises(RuntimeError):
        gmc.calc_optical_flow_pyr_lk(np.zeros((48, 64), np.uint8), np.zeros((48, 64), np.uint8), np.ones((3, 2), np.float32), device=0)
    with pytest.raises(RuntimeError):
        gmc.prepare_frame(frame, 2, device=0)
    t = BYTETracker(gmc_device=0)
    with pytest.raises(RuntimeError):
        t.update(np.array([[10, 10, 30, 40, 0.9, 0]], np.float32), frame)
