"""BoT-SORT's motion compensation on the GPU (csrc/gmc_kernels.hip) against the CPU restatement in oracle/gmc_oracle.py (numpy, ``G``)
-- and, secondarily, against the product's own host C++ -- and the tracker running on it (``model.track`` hands its tracker the
engine's device; the tracker core's checker is oracle/tracker_oracle.py)."""
import time

import numpy as np
import pytest

from cvsd_amd import gmc
from oracle import gmc_oracle as G

pytestmark = pytest.mark.gpu


def _smooth_noise(h, w, seed, sigma=2.0):
    """non-periodic texture: white noise blurred by a separable Gaussian, stretched to 8 bits (as tests/test_gmc.py)"""
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(h, w))
    r = int(3 * sigma)
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
    k /= k.sum()
    a = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 1, a)
    a = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 0, a)
    a = (a - a.min()) / (a.max() - a.min())
    return np.rint(a * 255).astype(np.uint8)


def _pair(h=150, w=200, dy=3, dx=-5, seeds=(3, 4)):
    big = ((_smooth_noise(h + 50, w + 60, seed=seeds[0], sigma=2.0).astype(np.int32) + _smooth_noise(h + 50, w + 60, seed=seeds[1], sigma=8.0)) // 2
           ).astype(np.uint8)
    return np.ascontiguousarray(big[20:20 + h, 30:30 + w]), np.ascontiguousarray(big[20 + dy:20 + dy + h, 30 + dx:30 + dx + w])


@pytest.mark.parametrize("shape,shift", [((150, 200), (3, -5)), ((120, 160), (-7, 9)), ((97, 131), (2, 2)), ((360, 640), (11, -14))])
def test_device_lucas_kanade_is_the_numpy_statement_of_the_algorithm(shape, shift):
    """same points kept, same positions to 1e-3 px (window sums are associated differently: lane-wise, then a butterfly), on even,
    odd and larger planes (the pyramid's reflect-101 borders and odd halvings included)"""
    prev, cur = _pair(shape[0], shape[1], shift[0], shift[1])
    pts = G.good_features_to_track(prev)[:400]
    # corners near the border too: windows that hang over the edge read reflected pixels
    edge = np.array([[1.0, 1.0], [shape[1] - 2.0, 2.0], [3.5, shape[0] - 2.5], [shape[1] - 1.0, shape[0] - 1.0]], np.float32)
    pts = np.concatenate([pts, edge])
    a, sa = gmc.calc_optical_flow_pyr_lk(prev, cur, pts, device=0)
    h, sh = gmc.calc_optical_flow_pyr_lk(prev, cur, pts)
    assert (sa == sh).mean() > 0.995
    both = sa & sh
    assert both.sum() > 100 and np.abs(a[both] - h[both]).max() < 1e-3
    if shape[0] <= 150:                                            # the numpy statement takes seconds per call
        b, sb = G.calc_optical_flow_pyr_lk(prev, cur, pts[:120])
        ok = sa[:120] & sb
        assert (sa[:120] == sb).mean() > 0.99 and np.abs(a[:120][ok] - b[ok]).max() < 1e-3
    # the recovered motion is the planted shift
    d = (a[both] - pts[both])
    assert np.abs(np.median(d[:, 0]) - (-shift[1])) < 0.05 and np.abs(np.median(d[:, 1]) - (-shift[0])) < 0.05


@pytest.mark.parametrize("shape,downscale", [((240, 320), 2), ((241, 323), 2), ((120, 160), 1), ((360, 640), 2), ((90, 121), 3), ((2, 80), 2), ((80, 3), 2)])
def test_device_frame_preparation_gives_numpys_plane_and_corner_list(shape, downscale):
    """luma, INTER_LINEAR resize, Shi-Tomasi corner map, threshold and non-maximum suppression on the GPU: the same gray plane byte
    for byte and the same corners in the same order as the numpy statement in oracle/gmc_oracle.py (odd sizes, no resize, a non-integer
    scale, and planes with a 1-pixel dimension: reflect-101 of a length-1 axis must terminate)"""
    rng = np.random.default_rng(shape[0] + downscale)
    g = _smooth_noise(shape[0], shape[1], seed=11, sigma=1.5)
    frame = np.stack([g, np.roll(g, 3, 1), 255 - g], axis=2)
    frame = np.clip(frame.astype(np.int32) + rng.integers(-6, 7, size=frame.shape), 0, 255).astype(np.uint8)
    gray_h, pts_h = G.prepare_frame(frame, downscale)
    gray_d, pts_d = gmc.prepare_frame(frame, downscale, 0)
    np.testing.assert_array_equal(gray_d, gray_h)
    assert len(pts_h) > 50 or min(shape) < 8
    np.testing.assert_array_equal(pts_d, pts_h)
    gray_c, pts_c = gmc.prepare_frame(frame, downscale, None)             # the product's host C++: the same again
    np.testing.assert_array_equal(gray_c, gray_h)
    np.testing.assert_array_equal(pts_c, pts_h)
    # a flat frame has no corners
    flat = np.full((shape[0], shape[1], 3), 90, np.uint8)
    assert gmc.prepare_frame(flat, downscale, 0)[1].shape == (0, 2)


def test_device_routine_rejects_bad_arguments_and_takes_empty_input():
    prev, cur = _pair()
    assert gmc.calc_optical_flow_pyr_lk(prev, cur, np.zeros((0, 2), np.float32), device=0)[0].shape == (0, 2)
    with pytest.raises(ValueError):
        gmc.calc_optical_flow_pyr_lk(prev, cur, np.ones((3, 2), np.float32), win=23, device=0)      # windows above 21 x 21: host only
    with pytest.raises(ValueError):
        gmc.calc_optical_flow_pyr_lk(prev, cur, np.ones((3, 2), np.float32), device=99)


def test_tracker_on_the_device_keeps_the_ids_of_the_host_tracker():
    """a panning camera over static people: the tracker whose optical flow runs on the GPU returns the ids of the numpy tracker of
    oracle/tracker_oracle.py (which estimates the motion with oracle/gmc_oracle.py) and the boxes and ids of the product's host form"""
    from cvsd_amd.tracker import BYTETracker
    from oracle.tracker_oracle import BYTETracker as OracleTracker
    ora = OracleTracker()
    h, w = 240, 320
    big = ((_smooth_noise(h + 40, w + 400, seed=7, sigma=2.0).astype(np.int32) + _smooth_noise(h + 40, w + 400, seed=8, sigma=6.0)) // 2).astype(np.uint8)
    people = np.array([[60, 60, 100, 180], [150, 40, 190, 170], [240, 80, 275, 200]], np.float32)
    host, dev = BYTETracker(), BYTETracker(gmc_device=0)
    t_host = t_dev = 0.0
    for k in range(12):
        off = 9 * k
        frame = np.repeat(big[20:20 + h, off:off + w, None], 3, axis=2)
        det = np.concatenate([people - [off, 0, off, 0] + 100 * np.array([1, 0, 1, 0]), np.full((3, 1), 0.9, np.float32), np.zeros((3, 1), np.float32)], axis=1
                             ).astype(np.float32)
        t0 = time.perf_counter(); a = host.update(det, frame); t1 = time.perf_counter(); b = dev.update(det, frame); t2 = time.perf_counter()
        t_host += t1 - t0; t_dev += t2 - t1
        c = ora.update(det, frame)
        assert a.shape == b.shape == c.shape
        if len(a):
            np.testing.assert_array_equal(a[:, 4], b[:, 4])                  # track ids
            np.testing.assert_array_equal(c[:, 4], b[:, 4])
            np.testing.assert_allclose(a[:, :4], b[:, :4], atol=2e-2)        # boxes after the compensated Kalman update
            np.testing.assert_allclose(c[:, :4], b[:, :4], atol=0.25)        # the oracle's RANSAC draws from another generator
    print(f"tracker update per frame: host {t_host / 12 * 1e3:.2f} ms, device {t_dev / 12 * 1e3:.2f} ms")


def test_enqueued_steps_give_the_host_warps_and_survive_misuse():
    """GMC.begin enqueues a frame's step, GMC.apply of the same frame object collects it: the warps of a panning clip equal the host
    object's to 1e-3; a step enqueued for ANOTHER frame is discarded (the object starts over) instead of being taken for this one"""
    h, w = 240, 320
    big = ((_smooth_noise(h + 40, w + 200, seed=17, sigma=2.0).astype(np.int32) + _smooth_noise(h + 40, w + 200, seed=18, sigma=6.0)) // 2).astype(np.uint8)
    frames = [np.repeat(big[10:10 + h, 7 * k:7 * k + w, None], 3, axis=2).copy() for k in range(8)]
    host, dev, ora = gmc.GMC(), gmc.GMC(device=0), G.GMC()
    for k, f in enumerate(frames):
        dev.begin(f)
        a, b = host.apply(f), dev.apply(f)
        np.testing.assert_allclose(b, a, atol=1e-3)
        np.testing.assert_allclose(b, ora.apply(f), atol=0.05)                  # numpy statement: same points, its own RANSAC generator
        np.testing.assert_array_equal(dev.prev_frame, ora.prev_frame)
        np.testing.assert_array_equal(dev.prev_points, ora.prev_points)
        if k:
            assert abs(a[0, 2] - (-7.0)) < 0.3 and abs(a[1, 2]) < 0.3           # the camera pans 7 px per frame
    np.testing.assert_array_equal(dev.prev_frame, host.prev_frame)
    np.testing.assert_array_equal(dev.prev_points, host.prev_points)
    # misuse: a step is pending for frames[0], the tracker asks about frames[1]
    dev.begin(frames[0])
    w1 = dev.apply(frames[1])
    np.testing.assert_array_equal(w1, np.eye(2, 3))                             # started over: first frame of a new sequence
    np.testing.assert_allclose(dev.apply(frames[2]), dev_pair(frames[1], frames[2]), atol=1e-3)


def dev_pair(f0, f1):
    g = gmc.GMC()
    g.apply(f0)
    return g.apply(f1)


def test_step_api_order_is_checked():
    import ctypes as C
    from cvsd_amd import _lib
    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.mi355_gmc_create(0, C.byref(h)) == 0
    frame = np.zeros((64, 96, 3), np.uint8)
    gray, eig, ok = np.empty((64, 96), np.uint8), np.empty((64, 96), np.float32), np.empty((64, 96), np.uint8)
    assert lib.mi355_gmc_step_finish(h, gray.ctypes.data, eig.ctypes.data, ok.ctypes.data, None, None) == -1          # nothing pending
    pts = np.ones((4, 2), np.float32)
    args = (frame.ctypes.data, 64, 96, 64, 96, None, None, 0.01)
    assert lib.mi355_gmc_step_begin(h, *args, pts.ctypes.data, 4, 21, 3, 30, 0.01, 1e-4) == -1                         # points, but no previous plane
    assert lib.mi355_gmc_step_begin(h, *args, None, 0, 21, 3, 30, 0.01, 1e-4) == 0
    assert lib.mi355_gmc_step_begin(h, *args, None, 0, 21, 3, 30, 0.01, 1e-4) == -1                                    # one step at a time
    assert lib.mi355_gmc_step_finish(h, gray.ctypes.data, eig.ctypes.data, ok.ctypes.data, None, None) == 0
    assert (gray == 0).all() and not ok.any()
    lib.mi355_gmc_destroy(h)


@pytest.mark.parametrize("shape", [(24, 40), (50, 44), (1080, 1920)])
def test_device_steps_on_tiny_and_large_frames_follow_the_host(shape):
    """frames smaller than the 21 x 21 window (no pyramid level fits, few or no corners) and a 1080p frame (buffer sizing): the GPU
    object returns the host object's warps"""
    h, w = shape
    g = _smooth_noise(h + 12, w + 12, seed=29, sigma=1.5)
    frames = [np.repeat(g[4 + k:4 + k + h, 3 * (k % 3):3 * (k % 3) + w, None], 3, axis=2).copy() for k in range(3)]
    host, dev = gmc.GMC(), gmc.GMC(device=0)
    for f in frames:
        a, b = host.apply(f), dev.apply(f)
        np.testing.assert_allclose(b, a, atol=2e-3)
    np.testing.assert_array_equal(dev.prev_frame, host.prev_frame)
    np.testing.assert_array_equal(dev.prev_points, host.prev_points)


def test_frame_size_change_restarts_the_sequence():
    g = _smooth_noise(300, 400, seed=31, sigma=2.0)
    a = np.repeat(g[:240, :320, None], 3, axis=2).copy()
    b = np.repeat(g[:200, :300, None], 3, axis=2).copy()
    dev = gmc.GMC(device=0)
    dev.apply(a)
    np.testing.assert_array_equal(dev.apply(b), np.eye(2, 3))          # another plane size: first frame of a new sequence
    assert dev.prev_frame.shape == (100, 150)
    host = gmc.GMC()
    host.apply(b)
    b2 = np.repeat(g[2:202, 3:303, None], 3, axis=2).copy()
    np.testing.assert_allclose(dev.apply(b2), host.apply(b2), atol=2e-3)


@pytest.mark.parametrize("shape", [(240, 320), (97, 131), (24, 40)])
def test_batched_steps_are_the_single_steps_bit_for_bit(shape):
    """mi355_gmc_track_batch (all frame preparations of a detector batch as one set of launches, all Lucas-Kanade steps as one launch)
    against the frame-by-frame entry points on the same GPU: the same warps, bit for bit, and the same state afterwards -- across
    two batches of different length, and when single steps and batches alternate on one object"""
    h, w = shape
    g = _smooth_noise(h + 40, w + 120, seed=41, sigma=1.8)
    frames = [np.repeat(g[3 + (k % 4):3 + (k % 4) + h, 5 * k:5 * k + w, None], 3, axis=2).copy() for k in range(13)]
    one, bat, mix = gmc.GMC(device=0), gmc.GMC(device=0), gmc.GMC(device=0)
    want = np.stack([one.apply(f) for f in frames])
    got = np.concatenate([bat.apply_batch(frames[:8]), bat.apply_batch(frames[8:])])
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(bat.prev_frame, one.prev_frame)
    np.testing.assert_array_equal(bat.prev_points, one.prev_points)
    m = [mix.apply(frames[0]), *mix.apply_batch(frames[1:6]), mix.apply(frames[6]), mix.apply(frames[7]), *mix.apply_batch(frames[8:9]), *mix.apply_batch(frames[9:])]
    np.testing.assert_array_equal(np.stack(m), want)
    if min(shape) > 60:
        assert np.abs(want[1:, 0, 2] + 5.0).max() < 0.3                      # the camera pans 5 px per frame
    # the host object runs a batch as single steps
    host = gmc.GMC()
    np.testing.assert_allclose(host.apply_batch(frames[:4]), want[:4], atol=2e-3)
    # another frame size restarts the sequence inside the batched entry point too
    small = [np.ascontiguousarray(f[:h // 2 * 2 - 10, :w - 20]) for f in frames[:3]]
    w2 = bat.apply_batch(small)
    np.testing.assert_array_equal(w2[0], np.eye(2, 3))
    ref = gmc.GMC(device=0)
    np.testing.assert_array_equal(w2, np.stack([ref.apply(f) for f in small]))
    with pytest.raises(ValueError):
        bat.apply_batch([frames[0], small[0]])


def test_pending_frame_is_the_uploaded_frame_and_track_results_do_not_depend_on_sharing_it(monkeypatch):
    """Round 4: ``model.track`` hands the detector pass the copy of the frame its tracker's motion-compensation step has just uploaded
    (``mi355_gmc_pending_frame`` -> ``mi355_yolo_infer_device``).  (a) that device copy is the frame, byte for byte, valid while the step is
    pending and through its collection on the worker thread; (b) the loop's results -- boxes, ids, scores -- are those of the loop that
    uploads the frame a second time, and of the loop with the step collected on the calling thread (MI355_GMC_ASYNC=0 is read once
    per process, so that arm is the sharing knob only)."""
    import torch
    from cvsd_amd import YOLO
    from tools import synth
    h, w = 240, 320
    big = ((_smooth_noise(h + 40, w + 200, seed=11, sigma=2.0).astype(np.int32) + _smooth_noise(h + 40, w + 200, seed=12, sigma=6.0)) // 2).astype(np.uint8)
    frames = [np.ascontiguousarray(np.repeat(big[20:20 + h, 7 * k:7 * k + w, None], 3, axis=2)) for k in range(10)]
    g = gmc.GMC(device=0)
    g.begin(frames[0])
    ptr, ph, pw = g.pending_device_frame()
    assert (ph, pw) == (h, w)
    back = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
    import ctypes as C
    from cvsd_amd import _lib
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(back.data_ptr()), C.c_void_p(ptr), C.c_size_t(h * w * 3), C.c_int(3)) == 0      # device -> device
    np.testing.assert_array_equal(back.cpu().numpy(), frames[0])
    g.apply(frames[0])
    assert g.pending_device_frame() is None                                   # nothing pending any more

    _, sd = synth.synthetic_checkpoint("yolov8n", seed=0)
    outs = []
    for share in ("1", "0"):
        monkeypatch.setenv("MI355_TRACK_SHARED_FRAME", share)
        model = YOLO.from_state_dict("yolov8n", sd, device=0, batch_chunk=1)
        rows = []
        for f in frames:
            r = model.track(f, persist=True, conf=0.05)[0]
            rows.append((r.boxes.data.numpy().copy(), None if r.boxes.id is None else r.boxes.id.numpy().copy()))
        outs.append(rows)
    for (b1, i1), (b0, i0) in zip(*outs):
        np.testing.assert_array_equal(b1, b0)
        assert (i1 is None) == (i0 is None) and (i1 is None or np.array_equal(i1, i0))
