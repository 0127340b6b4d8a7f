"""BASELINE config 0 and the dataset sweep with the REAL engine: clips -> detection -> tracker -> the reference's CSV files."""
import csv
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dataset(root, n_frames=6):
    from tools import synth
    lines = ["Abuse/Abuse001_x264.mp4"]
    for k, label in enumerate(("Shoplifting", "Shopping")):
        os.makedirs(os.path.join(root, label), exist_ok=True)
        name = f"{label}00{k + 1}_x264"
        # a CLIP (one drifting scene), not unrelated frames: the tracker compensates camera motion between consecutive frames
        np.save(os.path.join(root, label, name + ".npy"), synth.synthetic_clip(n_frames, 240, 320, seed=60 + 10 * k))      # seeds on which the synthetic detector tracks a "person"
        lines.append(f"{label}/{name}.mp4")
    lst = os.path.join(root, "list.txt")
    with open(lst, "w") as f:
        f.write("\n".join(lines))
    return lst


def _read(path, missing_ok=False):
    if missing_ok and not os.path.exists(path):
        return []
    with open(path, "rb") as f:
        raw = f.read()
    assert raw.endswith(b"\r\n") and b"\n" not in raw.replace(b"\r\n", b"")           # csv.writer's excel dialect
    return list(csv.reader(raw.decode().splitlines()))


def test_preprocess_loop_writes_the_reference_csv(v8n, tmp_path):
    """preprocess.py:15-53 + model.py:36-81 with the engine in place of Ultralytics: one Shoplifting and one Shopping clip"""
    from cvsd_amd import YOLO
    from cvsd_amd import preprocess_driver as P
    from cvsd_amd.tracker_csv import Tracker
    root = str(tmp_path / "data")
    os.makedirs(root)
    lst = _dataset(root)
    out = str(tmp_path / "out")
    os.makedirs(out)
    t = Tracker(model=YOLO.from_state_dict("yolov8n", v8n[1]), out_dir=out)
    n = P.run(t, lst, root + "/", capture=P.NpyCapture, log=lambda *_: None)
    assert n == 12
    rows = _read(os.path.join(out, "ucf-crime_dataset.csv"))
    assert rows, "the synthetic detector tracks something on these frames"
    for r in rows:
        clip, name, frame, person, left, top, width, height, is_anomaly, anomaly = r
        assert clip == "2" and name == "Shoplifting001_x264.mp4" and 1 <= int(frame) <= 6      # list line 2, 1-based frames
        assert float(person) >= 1 and is_anomaly == "True" and anomaly == "Shoplifting"
        assert all(0.0 <= float(v) <= 1.0 for v in (left, top, width, height))                 # xywhn
    normal = _read(os.path.join(out, "ucf-crime_dataset-normal.csv"))
    assert normal and all(r[0] == "3" and r[8] == "False" and r[9] == "Shopping" for r in normal)


def test_sweep_with_the_engine_matches_the_sequential_loop(v8n, tmp_path):
    """cvsd_amd.sweep (batched detection, per-clip tracker) against the frame-by-frame loop: same detections per frame,
    hence the same rows (ids are per clip in both when the loop's tracker is reset per clip)"""
    from cvsd_amd import YOLO
    from cvsd_amd import preprocess_driver as P
    from cvsd_amd.sweep import sweep
    root = str(tmp_path / "data")
    os.makedirs(root)
    lst = _dataset(root, n_frames=6)
    m = YOLO.from_state_dict("yolov8n", v8n[1])
    out = str(tmp_path / "sweep")
    os.makedirs(out)
    written = sweep(m, lst, root + "/", out_dir=out, batch=4, capture=P.NpyCapture, log=lambda *_: None)
    rows = _read(os.path.join(out, "ucf-crime_dataset.csv"), True) + _read(os.path.join(out, "ucf-crime_dataset-normal.csv"), True)
    assert written == len(rows) > 0
    # reference-style loop, tracker reset at each clip
    want = []
    for i, label, name, rel in [(2, "Shoplifting", "Shoplifting001_x264.mp4", "Shoplifting/Shoplifting001_x264"),
                                (3, "Shopping", "Shopping002_x264.mp4", "Shopping/Shopping002_x264")]:
        m._tracker = None
        clip = np.load(os.path.join(root, rel + ".npy"))
        for k, frame in enumerate(clip, start=1):
            b = m.track(frame, persist=True, show=False, classes=[0], verbose=False)[0].boxes
            if b.is_track:
                for box in b:
                    want.append((str(i), name, str(k), float(box.id), [float(v) for v in box.xywhn[0]]))
    assert len(want) == len(rows)
    for r, w in zip(rows, want):
        assert (r[0], r[1], r[2]) == w[:3]
        np.testing.assert_allclose([float(v) for v in r[4:8]], w[4], rtol=0, atol=1e-6)


def test_pose_clip_to_poselift_pickle(v8n_pose, tmp_path):
    """config 2's consumer format: frames -> YOLOv8n-pose engine (batched) -> host tracker -> the per-video pickle
    {frame: {person: [bbox, keypoints(17,3)]}} that shopformer/data/poselift_dataset.py:256-295 reads"""
    import pickle
    from cvsd_amd import YOLO
    from cvsd_amd.poselift_bridge import video_to_poselift
    from tools import synth
    m = YOLO.from_state_dict("yolov8n-pose", v8n_pose[1])
    frames = synth.synthetic_clip(10, 240, 320, seed=77)          # consecutive frames of one scene (GMC + tracker need a video)
    out = str(tmp_path / "Shoplifting001.pkl")
    data = video_to_poselift(m, list(frames), out_path=out, conf=0.25, batch=4)
    with open(out, "rb") as f:
        assert pickle.load(f).keys() == data.keys()
    assert sorted(data) == list(range(10))                                  # one entry per frame, 0-based like PoseLift
    persons = [p for fr in data.values() for p in fr.items()]
    assert persons, "the synthetic pose model tracks somebody on these frames"
    for pid, (bbox, kp) in persons:
        assert isinstance(pid, int) and pid >= 1
        assert bbox.shape == (4,) and bbox[2] > 0 and bbox[3] > 0           # x, y, w, h in pixels of the 320x240 frame
        assert kp.shape == (17, 3) and np.isfinite(kp).all() and (kp[:, 2] >= 0).all() and (kp[:, 2] <= 1).all()
    # ids persist: some person is present in at least two consecutive frames
    assert any(set(data[k]) & set(data[k + 1]) for k in range(9))


def test_engine_to_poselift_reproduces_the_reference_loader_fixture(v8n_pose):
    """SURVEY 8(f) rank 1, pinned by the reference itself: tests/golden/poselift_fixture.npz holds the bridge dict of a
    20-frame synthetic clip (canonical-oracle detections) and what /root/reference/shopformer/data/poselift_dataset.py:
    PoseLiftDataset made of its pickle tree (generated in the build container, tests/golden/make_poselift_fixture.py).
    The real engine through the same bridge must give the same dict bit for bit, hence the same training windows."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from _poselift_windows import unflatten, windows
    from cvsd_amd import YOLO
    from cvsd_amd.poselift_bridge import video_to_poselift
    from tools import synth
    fix = np.load(os.path.join(os.path.dirname(__file__), "golden", "poselift_fixture.npz"))
    n, h, w, seed, imgsz, batch, seq_len, stride = (int(v) for v in fix["meta"])
    m = YOLO.from_state_dict("yolov8n-pose", v8n_pose[1])
    data = video_to_poselift(m, list(synth.synthetic_clip(n, h, w, seed=seed)), conf=float(fix["conf"]), batch=batch, imgsz=imgsz)
    want = unflatten(fix["frame_keys"], fix["row_frame"], fix["row_pid"], fix["row_bbox"], fix["row_kpts"])
    assert list(data) == list(want)
    for f in want:
        assert list(data[f]) == list(want[f]), f"frame {f}: person ids differ"
        for pid in want[f]:
            np.testing.assert_array_equal(data[f][pid][0], want[f][pid][0])
            np.testing.assert_array_equal(data[f][pid][1], want[f][pid][1])
    for split, labels in (("train", None), ("test", fix["gt"])):
        x, y = windows(data, seq_len=seq_len, stride=stride, include_confidence=True, frame_labels=labels)
        np.testing.assert_array_equal(x, fix[f"{split}_xyc_x"])
        np.testing.assert_array_equal(y, fix[f"{split}_xyc_y"])
        # ... and the windows the second reference loader (shopformer_2, 18 keypoints with the synthetic neck) made of the same tree
        x18, y18 = windows(data, seq_len=seq_len, stride=stride, include_confidence=True, frame_labels=labels, num_keypoints=18)
        np.testing.assert_array_equal(x18, fix[f"s2_{split}_xyc_x"])
        np.testing.assert_array_equal(y18, fix[f"s2_{split}_xyc_y"])


def test_plan_file_tune_path_and_load_path(v8n, tmp_path, monkeypatch):
    """the persisted launch-plan choices: a fresh directory makes the first model TUNE (stopwatch) and write the file, a second
    model of the same image LOADS it (same plan hash, no rewrite); a file from a different planner fingerprint is ignored"""
    import glob
    import torch
    from cvsd_amd import YOLO
    from tools import synth
    monkeypatch.setenv("MI355_PLAN_CACHE", str(tmp_path))
    frames = torch.from_numpy(synth.synthetic_frames(2, 160, 160, seed=4)).cuda()
    m1 = YOLO.from_state_dict("yolov8n", v8n[1], batch_chunk=2)
    r1 = m1.predict(frames, imgsz=160)
    i1 = m1.plan_info()
    files = glob.glob(str(tmp_path / "*.plan"))
    assert i1["plan_source"] == "tuned" and len(files) == 1 and i1["launches_per_pass"] > 20 and i1["activation_bytes"] > 0
    stamp = os.stat(files[0]).st_mtime_ns
    m2 = YOLO.from_state_dict("yolov8n", v8n[1], batch_chunk=2)
    r2 = m2.predict(frames, imgsz=160)
    i2 = m2.plan_info()
    assert i2["plan_source"] == "cache" and i2["plan_hash"] == i1["plan_hash"] and os.stat(files[0]).st_mtime_ns == stamp
    for a, b in zip(r1, r2):
        np.testing.assert_array_equal(a.boxes.data.numpy(), b.boxes.data.numpy())
    # a SHIPPED plan directory (mi355_opts.plan_dir; the package's plans/ by default) is looked up before the machine's cache and is
    # never written to: the same launches (same hash), source "file"
    import shutil
    shipped = tmp_path / "shipped"
    shipped.mkdir()
    shutil.copy(files[0], shipped)
    monkeypatch.setenv("MI355_PLAN_CACHE", str(tmp_path / "empty_cache"))
    m4 = YOLO.from_state_dict("yolov8n", v8n[1], batch_chunk=2, plan_dir=str(shipped))
    r4 = m4.predict(frames, imgsz=160)
    i4 = m4.plan_info()
    assert i4["plan_source"] == "file" and i4["plan_hash"] == i1["plan_hash"] and not glob.glob(str(tmp_path / "empty_cache" / "*.plan"))
    for a, b in zip(r1, r4):
        np.testing.assert_array_equal(a.boxes.data.numpy(), b.boxes.data.numpy())
    monkeypatch.setenv("MI355_PLAN_CACHE", str(tmp_path))
    # a stale file (other fingerprint: another build, GPU or knob setting) is not trusted
    lines = open(files[0]).read().split("\n")
    head = lines[0].split()
    head[2] = f"{int(head[2], 16) ^ 1:x}"
    open(files[0], "w").write("\n".join([" ".join(head)] + lines[1:]))
    m3 = YOLO.from_state_dict("yolov8n", v8n[1], batch_chunk=2)
    m3.predict(frames, imgsz=160)
    assert m3.plan_info()["plan_source"] == "tuned"
