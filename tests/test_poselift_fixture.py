"""The PoseLift bridge against the REFERENCE'S OWN loader.

tests/golden/poselift_fixture.npz holds (a) a bridge dict produced by cvsd_amd.poselift_bridge.video_to_poselift from
canonical-oracle pose detections and (b) what /root/reference/shopformer/data/poselift_dataset.py:PoseLiftDataset returned
for the pickle tree of that dict (generated in the build container by tests/golden/make_poselift_fixture.py; only outputs
are stored).  Here, without the reference: the stored dict must survive the writer unchanged, and the restated loader of
tests/_poselift_windows.py must reproduce the reference loader's tensors exactly -- which licenses its use on the GPU box."""
import os
import pickle

import numpy as np
import pytest

from _poselift_windows import unflatten, windows

FIX = os.path.join(os.path.dirname(__file__), "golden", "poselift_fixture.npz")


@pytest.fixture(scope="module")
def fix():
    return np.load(FIX)


def test_fixture_has_windows_of_tracked_people(fix):
    assert int(fix["train_xy_n"]) == int(fix["test_xy_n"]) > 0
    n_frames, h, w = (int(v) for v in fix["meta"][:3])
    assert sorted(fix["frame_keys"].tolist()) == list(range(n_frames))            # one entry per frame, 0-based
    assert (fix["row_pid"] >= 1).all()
    b = fix["row_bbox"]                                                            # x, y, w, h inside the frame (clipped track boxes)
    assert (b[:, 0] >= 0).all() and (b[:, 1] >= 0).all() and (b[:, 0] + b[:, 2] <= w + 1e-3).all() and (b[:, 1] + b[:, 3] <= h + 1e-3).all()
    assert fix["train_xy_x"].shape[1:] == (2, 12, 17) and fix["train_xyc_x"].shape[1:] == (3, 12, 17)


@pytest.mark.parametrize("split,inc", [("train", False), ("train", True), ("test", False), ("test", True)])
def test_restated_loader_equals_the_reference_loader(fix, split, inc):
    data = unflatten(fix["frame_keys"], fix["row_frame"], fix["row_pid"], fix["row_bbox"], fix["row_kpts"])
    x, y = windows(data, seq_len=int(fix["meta"][6]), stride=int(fix["meta"][7]), include_confidence=inc,
                   frame_labels=fix["gt"] if split == "test" else None)
    key = f"{split}_{'xyc' if inc else 'xy'}"
    assert len(x) == int(fix[key + "_n"])
    np.testing.assert_array_equal(y, fix[key + "_y"])
    np.testing.assert_array_equal(x, fix[key + "_x"])                              # same float32 bits as the reference class
    if split == "test":
        assert set(y.tolist()) == {0, 1}                                           # the majority-label rule saw both classes


@pytest.mark.parametrize("split,inc", [("train", False), ("train", True), ("test", False), ("test", True)])
def test_restated_loader_equals_the_second_reference_loader_with_its_neck_keypoint(fix, split, inc):
    """/root/reference/shopformer_2/data/poselift_dataset.py (num_keypoints=18: COCO-17 + the synthetic neck of :57-91) on the same pickle
    tree: a second reference-held pin of the bridge's output format"""
    data = unflatten(fix["frame_keys"], fix["row_frame"], fix["row_pid"], fix["row_bbox"], fix["row_kpts"])
    x, y = windows(data, seq_len=int(fix["meta"][6]), stride=int(fix["meta"][7]), include_confidence=inc,
                   frame_labels=fix["gt"] if split == "test" else None, num_keypoints=18)
    key = f"s2_{split}_{'xyc' if inc else 'xy'}"
    assert len(x) == int(fix[key + "_n"]) > 0 and x.shape[1:] == (3 if inc else 2, 12, 18)
    np.testing.assert_array_equal(y, fix[key + "_y"])
    np.testing.assert_array_equal(x, fix[key + "_x"])
    # the neck really is the shoulders' midpoint wherever both shoulders are present (raw pixels, before normalisation)
    k = fix["row_kpts"]
    both = (np.abs(k[:, 5, :2]).sum(1) > 0) & (np.abs(k[:, 6, :2]).sum(1) > 0)
    assert both.any()


def test_writer_round_trip_is_lossless(fix, tmp_path):
    """PoseLiftWriter.add_frame / save reproduce the stored dict from tracker rows + keypoints (xywh boxes, float32)"""
    from cvsd_amd.poselift_bridge import PoseLiftWriter
    data = unflatten(fix["frame_keys"], fix["row_frame"], fix["row_pid"], fix["row_bbox"], fix["row_kpts"])
    w = PoseLiftWriter()
    for f, people in data.items():
        rows = np.asarray([[b[0], b[1], b[0] + b[2], b[1] + b[3], pid, 0.9, 0, 0] for pid, (b, _) in people.items()], np.float32).reshape(-1, 8)
        w.add_frame(f, rows, np.asarray([k for _, k in people.values()], np.float32).reshape(-1, 17, 3))
    p = tmp_path / "cam1_0001.pkl"
    w.save(str(p))
    back = pickle.load(open(p, "rb"))
    assert list(back) == list(data)
    for f in data:
        assert list(back[f]) == list(data[f])
        for pid in data[f]:
            np.testing.assert_allclose(back[f][pid][0], data[f][pid][0], rtol=0, atol=2e-5)   # x2 - x1 re-derived in fp32
            np.testing.assert_array_equal(back[f][pid][1], data[f][pid][1])
