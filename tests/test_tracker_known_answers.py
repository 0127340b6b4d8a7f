"""Known-answer tests of the host tracker behind ``YOLO.track`` (SURVEY.md 8(a) a12 / 8(f) rank 2;
``/root/reference/model.py:38,45,60-64``).  Every expected value below is derived by hand in the docstring of its test
from the published BoT-SORT / ByteTrack definitions (constant-velocity XYWH Kalman filter with std weights 1/20 and
1/160, two-stage IoU association with score fusion, track_buffer 30) -- none is produced by running the tracker.

Global motion compensation (``gmc_method: sparseOptFlow`` of botsort.yaml) has tests of its own (tests/test_gmc.py); here update()
is called without a frame, or on a static scene, where it is the identity.  The CSV test at the bottom states that explicitly: a
static person keeps exactly the measured box centre.
"""
import numpy as np
import pytest
import torch

from cvsd_amd import tracker as product_tracker
from oracle import tracker_oracle


@pytest.fixture(params=["product: host C++ (csrc/tracker_host.cpp)", "oracle: numpy (oracle/tracker_oracle.py)"])
def impl(request):
    """every hand-derived answer below is asked of BOTH implementations: the product's C++ core behind the C ABI and the numpy
    restatement that checks it"""
    return product_tracker if request.param.startswith("product") else tracker_oracle


def _det(*boxes):
    """rows x1,y1,x2,y2,conf,cls"""
    return np.asarray(boxes, dtype=np.float32).reshape(-1, 6)


def test_kalman_initiate_predict_update_by_hand(impl):
    """measurement z0 = (cx 100, cy 200, w 40, h 80).
    initiate: mean = (z0, 0,0,0,0); std = (2w/20, 2h/20, 2w/20, 2h/20, 10w/160, 10h/160, 10w/160, 10h/160)
              = (4, 8, 4, 8, 2.5, 5, 2.5, 5)  ->  P0 = diag(16, 64, 16, 64, 6.25, 25, 6.25, 25).
    predict (dt 1): mean unchanged (zero velocity); Q = diag((w/20)^2, (h/20)^2, (w/20)^2, (h/20)^2, (w/160)^2, ...)
              = diag(4, 16, 4, 16, 0.0625, 0.25, 0.0625, 0.25);  P1 = F P0 F' + Q:
              P1[x,x] = 16 + 6.25 + 4 = 26.25, P1[x,vx] = 6.25, P1[vx,vx] = 6.25 + 0.0625 = 6.3125
              P1[y,y] = 64 + 25 + 16 = 105,    P1[y,vy] = 25,   P1[vy,vy] = 25.25.
    update with z1 = (104, 200, 40, 80): R = diag(4, 16, 4, 16); S[x] = 26.25 + 4 = 30.25;
              K[x] = 26.25 / 30.25, K[vx] = 6.25 / 30.25; innovation 4 ->
              x = 100 + 4 * 26.25 / 30.25 = 103.47107..., vx = 25 / 30.25 = 0.826446...
              P2[x,x] = 26.25 - 26.25^2 / 30.25 = 3.47107..., P2[x,vx] = 6.25 - 26.25 * 6.25 / 30.25 = 0.826446...,
              P2[vx,vx] = 6.3125 - 6.25^2 / 30.25 = 5.021178...
    """
    kf = impl.KalmanFilterXYWH()
    mean, cov = kf.initiate(np.array([100.0, 200.0, 40.0, 80.0]))
    np.testing.assert_array_equal(mean, [100, 200, 40, 80, 0, 0, 0, 0])
    np.testing.assert_allclose(cov, np.diag([16, 64, 16, 64, 6.25, 25, 6.25, 25]), rtol=0, atol=1e-12)
    mean, cov = kf.predict(mean, cov)
    np.testing.assert_array_equal(mean, [100, 200, 40, 80, 0, 0, 0, 0])
    want = np.diag([26.25, 105, 26.25, 105, 6.3125, 25.25, 6.3125, 25.25])
    for i, v in enumerate([6.25, 25, 6.25, 25]):
        want[i, i + 4] = want[i + 4, i] = v
    np.testing.assert_allclose(cov, want, rtol=0, atol=1e-12)
    mean, cov = kf.update(mean, cov, np.array([104.0, 200.0, 40.0, 80.0]))
    np.testing.assert_allclose(mean, [100 + 4 * 26.25 / 30.25, 200, 40, 80, 25 / 30.25, 0, 0, 0], rtol=0, atol=1e-10)
    assert cov[0, 0] == pytest.approx(26.25 - 26.25 ** 2 / 30.25, abs=1e-10)
    assert cov[0, 4] == pytest.approx(6.25 - 26.25 * 6.25 / 30.25, abs=1e-10)
    assert cov[4, 4] == pytest.approx(6.3125 - 6.25 ** 2 / 30.25, abs=1e-10)
    # y, w, h had zero innovation: their means stay, their covariances shrink by the same formula
    assert cov[1, 1] == pytest.approx(105 - 105 ** 2 / 121, abs=1e-10)


def test_first_frame_tracks_are_reported_at_once_later_ones_after_one_confirmation(impl):
    """ByteTrack: a track activated on frame 1 is is_activated at once; one born later is 'unconfirmed' until it is
    matched on the following frame (threshold 0.7) -- so person B, first seen on frame 2, is first REPORTED on frame 3.
    Row layout: x1,y1,x2,y2,id,score,cls,idx."""
    tr = impl.BYTETracker()
    a = (10, 10, 50, 90, 0.9, 0)
    b = (200, 40, 260, 160, 0.8, 0)
    out = tr.update(_det(a))
    assert out.shape == (1, 8) and out[0, 4] == 1 and out[0, 7] == 0
    np.testing.assert_allclose(out[0, :4], a[:4], atol=1e-4)            # state == first measurement
    out = tr.update(_det(a, b))
    assert out[:, 4].tolist() == [1.0]                                   # B exists (id 2) but is unconfirmed
    out = tr.update(_det(b, a))                                          # detection order must not matter
    assert sorted(out[:, 4].tolist()) == [1.0, 2.0]
    by_id = {int(r[4]): r for r in out}
    assert by_id[1][7] == 1 and by_id[2][7] == 0                         # idx = row of THIS frame's detections
    assert by_id[2][5] == pytest.approx(0.8) and by_id[1][5] == pytest.approx(0.9)


def test_two_people_crossing_keep_their_ids(impl):
    """A walks right (+6 px/frame), B walks left (-6 px/frame) on rows 30 px apart; they cross around frame 11.  Each
    track's constant-velocity prediction lands on its own detection (IoU ~ 1) and 12 px per frame away from the other's
    continuation, so the first association keeps A = id 1 moving right and B = id 2 moving left through the crossing."""
    tr = impl.BYTETracker()
    for f in range(24):
        xa, xb = 100 + 6 * f, 220 - 6 * f
        out = tr.update(_det((xa, 100, xa + 40, 180, 0.9, 0), (xb, 130, xb + 40, 210, 0.85, 0)))
        assert out.shape == (2, 8)
        by_id = {int(r[4]): r for r in out}
        assert set(by_id) == {1, 2}
        assert abs(by_id[1][0] - xa) < 3.0 and abs(by_id[1][1] - 100) < 1.0      # Kalman state follows A's measurements
        assert abs(by_id[2][0] - xb) < 3.0 and abs(by_id[2][1] - 130) < 1.0
    assert tr._ids_issued == 2


def test_low_score_detection_is_used_by_the_second_association_only(impl):
    """Scores in (track_low_thresh 0.1, track_high_thresh 0.25) skip the first association: they can CONTINUE a tracked
    person (second stage, plain IoU distance <= 0.5) but never start a track (new_track_thresh 0.25)."""
    tr = impl.BYTETracker()
    box = (50, 60, 110, 200)
    tr.update(_det((*box, 0.9, 0)))
    out = tr.update(_det((*box, 0.15, 0), (300, 60, 360, 200, 0.15, 0)))
    assert out.shape == (1, 8) and out[0, 4] == 1 and out[0, 5] == pytest.approx(0.15) and out[0, 7] == 0
    out = tr.update(_det((*box, 0.9, 0)))
    assert out[:, 4].tolist() == [1.0] and tr._ids_issued == 1           # the weak far-away box never became a track
    # a score <= 0.1 is ignored by both stages: the track goes lost, nothing is reported
    assert tr.update(_det((*box, 0.1, 0))).shape == (0, 8)


def test_lost_track_is_refound_inside_the_buffer_and_retired_after_it(impl):
    """track_buffer 30 at 30 fps: a lost track is marked Removed once frame_id - end_frame > 30.  A static person seen on
    frames 1-3 and again on frame 14 (10 empty frames: 4..13) is re-activated with the SAME id.  With 32 empty frames
    (4..35) the track is marked Removed on frame 34 (34 - 3 = 31 > 30) and has left the candidate pool by frame 36, so the
    person gets a NEW id.  byte_tracker.py subtracts the removed list from the lost list BEFORE appending the frame's own
    removals, so a track retired on frame 34 is still offered to the association of frame 35 -- with exactly 31 empty
    frames the old id comes back (upstream's bookkeeping order, kept).  Empty frames count: the tracker steps on every
    frame (trackers/track.py:on_predict_postprocess_end)."""
    box = (120, 80, 180, 220, 0.9, 0)
    empty = np.zeros((0, 6), np.float32)
    tr = impl.BYTETracker()
    for _ in range(3):
        assert tr.update(_det(box))[:, 4].tolist() == [1.0]
    for _ in range(10):
        assert tr.update(empty).shape == (0, 8)
    out = tr.update(_det(box))
    assert out[:, 4].tolist() == [1.0] and tr.frame_id == 14
    np.testing.assert_allclose(out[0, :4], box[:4], atol=0.5)

    tr = impl.BYTETracker()
    for _ in range(3):
        tr.update(_det(box))
    for k in range(32):
        tr.update(empty)
    assert not tr.lost_stracks                                            # marked on frame 34, gone from the pool on 35
    out = tr.update(_det(box))                                            # frame 36: a new, unconfirmed track
    assert out.shape == (0, 8)
    out = tr.update(_det(box))
    assert out[:, 4].tolist() == [2.0]

    tr = impl.BYTETracker()
    for _ in range(3):
        tr.update(_det(box))
    for k in range(31):
        tr.update(empty)
    assert tr.update(_det(box))[:, 4].tolist() == [1.0]                   # frame 35: the one-frame grace of the list order


def test_two_trackers_do_not_share_ids(impl):
    """ids belong to the tracker instance: a sweep / bridge tracker created while model.track(persist=True) is alive
    must not reset or advance the latter's counter."""
    t1 = impl.BYTETracker()
    t1.update(_det((10, 10, 50, 90, 0.9, 0)))
    t2 = impl.BYTETracker()
    t2.update(_det((10, 10, 50, 90, 0.9, 0), (100, 10, 150, 90, 0.9, 0)))
    out = t1.update(_det((10, 10, 50, 90, 0.9, 0), (300, 10, 350, 90, 0.9, 0)))
    assert out[:, 4].tolist() == [1.0]
    out = t1.update(_det((10, 10, 50, 90, 0.9, 0), (300, 10, 350, 90, 0.9, 0)))
    assert sorted(out[:, 4].tolist()) == [1.0, 2.0]                      # not 3: t2's two ids are its own


class _FakeModel:
    """YOLO.track with predict() replaced by a script of detections (the engine itself is covered by the GPU tests)."""

    def __init__(self, script, shape=(240, 320)):
        from cvsd_amd.engine import YOLO
        self.m = YOLO.__new__(YOLO)
        self.m._tracker = None
        self.script, self.shape, self.k = script, shape, 0
        self.m.predict = self._predict

    def _predict(self, batch, conf=None, **kw):
        from cvsd_amd.results import Results
        assert conf == 0.1                                               # model.track predicts at conf 0.1
        d = self.script[self.k]
        self.k += 1
        return [Results(None, "f.jpg", {0: "person"}, boxes=torch.as_tensor(d, dtype=torch.float32).reshape(-1, 6),
                        orig_shape=self.shape)]


def test_track_steps_on_empty_frames_and_reports_is_track_false():
    """model.py:45: a frame without tracks has boxes.is_track False and is dropped by the caller; the tracker has
    stepped on it all the same (32 empty frames retire id 1, see above)."""
    box = [120, 80, 180, 220, 0.9, 0]
    script = [[box]] * 3 + [[]] * 32 + [[box]] * 2
    fm = _FakeModel(script)
    frame = np.zeros((240, 320, 3), np.uint8)
    ids = []
    for _ in script:
        b = fm.m.track(frame, persist=True, show=False, classes=[0], verbose=False)[0].boxes
        ids.append([float(x.id) for x in b] if b.is_track else None)
    assert ids[:3] == [[1.0]] * 3 and ids[3:35] == [None] * 32
    assert ids[35] is None and ids[36] == [2.0]                          # new track: unconfirmed, then reported as id 2
    assert fm.m._tracker.frame_id == len(script)


def test_track_rows_are_clipped_to_the_frame_and_gmc_is_identity():
    """Results.update clips the Kalman-state boxes (utils/ops.py:clip_boxes) before xywhn is read for the CSV
    (model.py:61-64): a person overhanging the left/top edge by 10 px in a 320x240 frame has
    x1,y1 = 0 -> xywhn = ((0+50)/2/320, (0+90)/2/240, 50/320, 90/240).  GMC is the identity: a static camera and a
    static person give back exactly the measured box on every frame."""
    from cvsd_amd.tracker_csv import Tracker
    box = [-10, -10, 50, 90, 0.9, 0]
    fm = _FakeModel([[box]] * 3)
    t = Tracker(model=fm.m)
    for _ in range(3):
        boxes = t.get_boxes(np.zeros((240, 320, 3), np.uint8))
        assert boxes.is_track and len(boxes) == 1
        np.testing.assert_allclose(boxes.xyxy.numpy(), [[0, 0, 50, 90]], atol=1e-4)
        row = t.rows_for(boxes, 7, 3, "Shoplifting", "Shoplifting001_x264.mp4")[0]
        assert row.person == 1.0
        np.testing.assert_allclose([row.left, row.top, row.width, row.height],
                                   [25 / 320, 45 / 240, 50 / 320, 90 / 240], atol=1e-6)
        assert all(0.0 <= v <= 1.0 for v in (row.left, row.top, row.width, row.height))
