import io
import os

import numpy as np
import pytest
import torch

from cvsd_amd.results import Boxes, Keypoints, Results
from cvsd_amd.tracker_csv import ANOMALIES, BBox, Tracker, write_rows


def test_boxes_surface_used_by_reference():
    """model.py:45,56-67: is_track, iteration -> 1-row Boxes, float(box.id), box.xywhn[0][k]"""
    data = torch.tensor([[100.0, 50.0, 300.0, 250.0, 7.0, 0.9, 0.0], [0.0, 0.0, 64.0, 48.0, 9.0, 0.5, 0.0]])
    b = Boxes(data, (240, 320))
    assert b.is_track and len(b) == 2
    rows = [box for box in b]
    assert len(rows) == 2 and all(isinstance(r, Boxes) and r.data.shape == (1, 7) for r in rows)
    assert float(rows[0].id) == 7.0
    np.testing.assert_allclose(rows[0].xywhn[0].numpy(), [200 / 320, 150 / 240, 200 / 320, 200 / 240], rtol=1e-7)
    assert b.conf.tolist() == [0.8999999761581421, 0.5] and b.cls.tolist() == [0.0, 0.0]
    nb = Boxes(data[:, [0, 1, 2, 3, 5, 6]], (240, 320))
    assert not nb.is_track and nb.id is None
    np.testing.assert_allclose(nb.xyxyn[1].numpy(), [0, 0, 0.2, 0.2], rtol=1e-7)
    assert isinstance(nb.numpy().data, np.ndarray) and len(Boxes(torch.zeros(0, 6), (4, 4))) == 0


def test_keypoints_low_conf_zeroed():
    k = torch.tensor([[[10.0, 20.0, 0.9], [30.0, 40.0, 0.4]]])
    kp = Keypoints(k, (100, 200))
    assert kp.xy.tolist() == [[[10.0, 20.0], [0.0, 0.0]]] and kp.has_visible
    np.testing.assert_allclose(kp.xyn[0, 0].numpy(), [0.05, 0.2])
    assert kp.conf.tolist() == [[0.8999999761581421, 0.4000000059604645]]


def test_results_index_and_update():
    r = Results(None, "x.jpg", {0: "person"}, boxes=torch.rand(3, 6), keypoints=torch.rand(3, 17, 3), orig_shape=(48, 64))
    sub = r[[2, 0]]
    assert len(sub) == 2 and sub.keypoints.data.shape == (2, 17, 3)
    sub.update(boxes=torch.rand(2, 7))
    assert sub.boxes.is_track


GOLDEN_CSV = (
    "7,Shoplifting003_x264.mp4,12,3.0,0.625,0.625,0.625,0.8333333134651184,True,Shoplifting\r\n"
    "7,Shoplifting003_x264.mp4,12,5.0,0.10000000149011612,0.10000000149011612,0.20000000298023224,0.20000000298023224,True,Shoplifting\r\n"
)


class _FakeModel:
    def __init__(self, rows):
        self.rows = rows

    def track(self, frame, persist=True, show=False, classes=None, verbose=False):
        assert persist and classes == [0] and not show and not verbose          # the reference's literal call
        data = torch.tensor(self.rows, dtype=torch.float32).reshape(-1, 7 if self.rows and len(self.rows[0]) == 7 else 6)
        return [Results(frame, "f", {0: "person"}, boxes=data, orig_shape=frame.shape[:2])]


def test_csv_rows_byte_for_byte(tmp_path):
    """dataclass-csv semantics: astuple -> csv.writer (excel dialect), no header, append; left/top = box centre"""
    frame = np.zeros((240, 320, 3), np.uint8)
    m = _FakeModel([[100.0, 50.0, 300.0, 250.0, 3.0, 0.9, 0.0], [0.0, 0.0, 64.0, 48.0, 5.0, 0.5, 0.0]])
    t = Tracker(model=m, out_dir=str(tmp_path))
    rows = t.save_to_dataset(frame, 7, 12.0, "Shoplifting", "Shoplifting003_x264.mp4")
    assert isinstance(rows[0], BBox) and rows[0].frame == 12 and rows[0].is_anomaly
    p = tmp_path / "ucf-crime_dataset.csv"
    assert p.read_bytes().decode() == GOLDEN_CSV
    t.save_to_dataset(frame, 7, 13.0, "Shoplifting", "Shoplifting003_x264.mp4")          # append, never truncate
    assert p.read_bytes().decode().count("\r\n") == 4
    t.save_to_dataset(frame, 8, 1.0, "Shopping", "Shopping001_x264.mp4")                  # normal label -> other file
    assert (tmp_path / "ucf-crime_dataset-normal.csv").read_bytes().decode().startswith("8,Shopping001_x264.mp4,1,3.0,")
    assert "Shopping" not in ANOMALIES and "Shoplifting" in ANOMALIES


def test_untracked_frame_is_dropped(tmp_path):
    t = Tracker(model=_FakeModel([[1.0, 2.0, 3.0, 4.0, 0.9, 0.0]]), out_dir=str(tmp_path))
    assert t.save_to_dataset(np.zeros((8, 8, 3), np.uint8), 1, 1.0, "Shoplifting", "x") is None
    assert not os.listdir(tmp_path)
    with pytest.raises(TypeError):
        write_rows(str(tmp_path / "x.csv"), [("not", "a", "BBox")])


def test_preprocess_driver_semantics(tmp_path):
    """clip counter counts skipped lines; frame number is 1-based; unopenable clips are skipped"""
    from cvsd_amd import preprocess_driver as P
    (tmp_path / "Shoplifting").mkdir()
    np.save(tmp_path / "Shoplifting" / "Shoplifting003_x264.npy", np.zeros((3, 8, 8, 3), np.uint8))
    lst = tmp_path / "list.txt"
    lst.write_text("Abuse/Abuse001_x264.mp4\nShoplifting/Shoplifting003_x264.mp4\nShoplifting/Missing_x264.mp4")
    calls = []

    class T:
        def save_to_dataset(self, frame, i, n, label, name):
            calls.append((i, n, label, name, frame.shape))
    logs = []
    n = P.run(T(), str(lst), str(tmp_path) + "/", capture=P.NpyCapture, log=logs.append)
    assert n == 3
    assert calls == [(2, float(k), "Shoplifting", "Shoplifting003_x264.mp4", (8, 8, 3)) for k in (1, 2, 3)]
    assert any("Failed to load video" in m for m in logs) and any("Skipping, Abuse" in m for m in logs)


def test_tracker_assigns_stable_ids():
    from cvsd_amd.tracker import BYTETracker
    tr = BYTETracker()
    det = np.array([[10, 10, 50, 90, 0.9, 0], [200, 40, 260, 160, 0.8, 0]], np.float32)
    out1 = tr.update(det)
    assert out1.shape == (2, 8) and sorted(out1[:, 4].tolist()) == [1.0, 2.0]
    ids = {tuple(np.round(r[:2] / 50)): r[4] for r in out1}
    for step in range(1, 5):
        moved = det.copy()
        moved[:, [0, 2]] += 3 * step
        out = tr.update(moved[::-1].copy())                      # detection order must not matter
        assert out.shape == (2, 8)
        for r in out:
            assert ids[tuple(np.round((r[:2] - 3 * step) / 50))] == r[4]
    assert tr.update(np.zeros((0, 6), np.float32)).shape == (0, 8)


def test_poselift_bridge_layout(tmp_path):
    """{frame_num: {person_id: [bbox, kpts(17,3)]}} exactly as shopformer/data/poselift_dataset.py:256-295 parses it"""
    import pickle
    from cvsd_amd.poselift_bridge import PoseLiftWriter
    w = PoseLiftWriter()
    rows = np.array([[10, 20, 50, 100, 3, 0.9, 0, 0], [200, 40, 260, 160, 7, 0.8, 0, 1]], np.float32)
    kp = np.random.default_rng(0).random((2, 17, 3)).astype(np.float32)
    for f in range(12):
        w.add_frame(f, rows, kp + f)
    w.add_frame(12, np.zeros((0, 8), np.float32), np.zeros((0, 17, 3), np.float32))
    p = tmp_path / "Shoplifting003.pkl"
    w.save(str(p))
    data = pickle.load(open(p, "rb"))
    assert sorted(data) == list(range(13)) and sorted(data[0]) == [3, 7] and data[12] == {}
    bbox, k = data[5][7]
    assert bbox.tolist() == [200.0, 40.0, 60.0, 120.0] and k.shape == (17, 3)
    # the reference loader's own acceptance rules (poselift_dataset.py:266-295)
    for frame_num, frame_data in data.items():
        for pid, person in frame_data.items():
            assert isinstance(person, (list, tuple)) and len(person) >= 2
            assert not np.isnan(np.array(person[1])).any()
    per_person = {pid: sorted(f for f, d in data.items() if pid in d) for pid in (3, 7)}
    assert all(len(v) >= 12 for v in per_person.values())          # seq_len 12 windows exist


def test_image_directory_capture_decodes_bgr_frames(tmp_path):
    """the host front-end without OpenCV: a clip as a directory of frame images (Pillow decode), cv2.VideoCapture semantics
    (preprocess.py:31-44): BGR uint8 frames, 1-based CAP_PROP_POS_FRAMES after read(), False at the end, closed when missing"""
    from PIL import Image
    from cvsd_amd import preprocess_driver as P
    clip = tmp_path / "Shoplifting" / "Shoplifting001_x264"
    clip.mkdir(parents=True)
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, size=(3, 24, 32, 3), dtype=np.uint8)            # BGR
    for k, f in enumerate(frames):
        Image.fromarray(f[..., ::-1]).save(clip / f"{k + 1:04d}.png")             # PNG is lossless: exact round trip
    cap = P.open_capture(str(tmp_path / "Shoplifting" / "Shoplifting001_x264.mp4"))
    assert isinstance(cap, P.ImageDirCapture) and cap.isOpened()
    for k in range(3):
        ok, f = cap.read()
        assert ok and f.dtype == np.uint8 and f.flags["C_CONTIGUOUS"] and cap.get(P.CAP_PROP_POS_FRAMES) == k + 1
        np.testing.assert_array_equal(f, frames[k])
    assert cap.read() == (False, None)
    cap.release()
    assert not P.open_capture(str(tmp_path / "Shoplifting" / "missing.mp4")).isOpened()
    jpg = tmp_path / "Shopping" / "Shopping001_x264"
    jpg.mkdir(parents=True)
    Image.fromarray(frames[0][..., ::-1]).save(jpg / "0001.jpg", quality=95)
    ok, f = P.open_capture(str(jpg)).read()
    assert ok and f.shape == (24, 32, 3)                                          # JPEG: lossy, shape and layout only


def test_image_directory_capture_orders_unpadded_frame_names_numerically(tmp_path):
    """frame dumps named 1.jpg .. 12.jpg (no zero padding) must be read as 1, 2, ..., 12 -- a plain string sort would give
    1, 10, 11, 12, 2, ... and label the CSV rows with the wrong CAP_PROP_POS_FRAMES"""
    from PIL import Image
    from cvsd_amd import preprocess_driver as P
    clip = tmp_path / "clip"
    clip.mkdir()
    for k in range(1, 13):
        Image.fromarray(np.full((8, 8, 3), k, dtype=np.uint8)).save(clip / f"{k}.png")
    cap = P.open_capture(str(clip))
    got = []
    while True:
        ok, f = cap.read()
        if not ok:
            break
        got.append(int(f[0, 0, 0]))
    assert got == list(range(1, 13))
    mixed = tmp_path / "mixed"
    mixed.mkdir()
    for name, v in (("frame_10.png", 10), ("frame_9.png", 9), ("frame_100.png", 100), ("frame_0011.png", 11)):
        Image.fromarray(np.full((8, 8, 3), v, dtype=np.uint8)).save(mixed / name)
    cap = P.open_capture(str(mixed))
    assert [int(cap.read()[1][0, 0, 0]) for _ in range(4)] == [9, 10, 11, 100]
