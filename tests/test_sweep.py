"""Clip-sharded dataset sweep: the 2-rank result (gloo) is byte-identical to the single-process one."""
import os
import socket

import numpy as np
import pytest
import torch

from cvsd_amd.results import Results


class FakePoseModel:
    """Deterministic stand-in for the engine: two 'persons' whose boxes are a function of the frame's first pixels."""
    task = "detect"

    def predict(self, frames, conf=0.25, classes=None, **kw):
        assert classes == [0] and conf == 0.1
        out = []
        for f in np.asarray(frames):
            h, w = f.shape[:2]
            a, b = float(f[0, 0, 0]), float(f[0, 1, 0])
            rows = [[10 + a, 20, 60 + a, 120, 0.9, 0.0], [150, 30 + b, 210, 150 + b, 0.8, 0.0]]
            if f[0, 2, 0] % 5 == 0:
                rows = rows[:1]
            out.append(Results(f, "x", {0: "person"}, boxes=torch.tensor(rows, dtype=torch.float32), orig_shape=(h, w)))
        return out


def _make_dataset(root):
    rng = np.random.default_rng(0)
    lines = ["Abuse/Abuse001_x264.mp4"]
    for label, n_clips in (("Shoplifting", 3), ("Shopping", 2)):
        os.makedirs(os.path.join(root, label), exist_ok=True)
        for k in range(n_clips):
            name = f"{label}{k:03d}_x264"
            t = int(rng.integers(5, 12))
            clip = np.zeros((t, 240, 320, 3), np.uint8)
            clip[:, 0, 0, 0] = np.arange(t) * 2 + k             # slow horizontal motion
            clip[:, 0, 1, 0] = np.arange(t) + 3 * k
            clip[:, 0, 2, 0] = rng.integers(0, 20, t)
            np.save(os.path.join(root, label, name + ".npy"), clip)
            lines.append(f"{label}/{name}.mp4")
    lines.insert(3, "Shoplifting/Missing_x264.mp4")              # cannot be opened: skipped with a message
    lst = os.path.join(root, "list.txt")
    with open(lst, "w") as f:
        f.write("\n".join(lines))
    return lst


def _run(rank, world, port, root, out_dir):
    import torch.distributed as dist
    from cvsd_amd import preprocess_driver as P
    from cvsd_amd.sweep import sweep
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        sweep(FakePoseModel(), os.path.join(root, "list.txt"), root + "/", out_dir=out_dir, batch=4, capture=P.NpyCapture,
              log=lambda *_: None)
    finally:
        if world > 1:
            dist.destroy_process_group()


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sweep_two_ranks_equals_one(tmp_path):
    import torch.multiprocessing as mp
    root = str(tmp_path / "data")
    os.makedirs(root)
    _make_dataset(root)
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    _run(0, 1, 0, root, one)
    ctx = mp.get_context("spawn")
    port = _port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, root, two)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for fn in ("ucf-crime_dataset.csv", "ucf-crime_dataset-normal.csv"):
        a = open(os.path.join(one, fn), "rb").read()
        assert a == open(os.path.join(two, fn), "rb").read() and len(a) > 0
    rows = open(os.path.join(one, "ucf-crime_dataset.csv"), "rb").read().decode().split("\r\n")[:-1]
    clips = [int(r.split(",")[0]) for r in rows]
    assert clips == sorted(clips) and set(clips) == {2, 3, 5}     # list line numbers; 1 = Abuse (skipped), 4 = missing file
    ids = [float(r.split(",")[3]) for r in rows]
    first = {c: min(i for i, cc in zip(ids, clips) if cc == c) for c in set(clips)}
    assert first[2] < first[3] < first[5]                          # person ids keep growing from clip to clip
    assert rows[0].split(",")[1] == "Shoplifting000_x264.mp4" and rows[0].endswith(",True,Shoplifting")


def test_clips_are_balanced_by_frame_count_not_dealt_round_robin():
    """'balance by frame count'.  UCF-Crime-like clip lengths (50 ... 2000 frames, log-uniform, 145 clips = the Shoplifting
    + Shopping lines of Anomaly_Train.txt): the longest-processing-time assignment keeps the busiest rank within 15 % of the mean load
    at 2 and at 8 ranks, where dealing clips k % world does not; unknown lengths count as average clips; every clip has one owner."""
    from cvsd_amd.sweep import assign_clips
    rng = np.random.default_rng(5)
    for trial in range(20):
        n = 145 if trial % 2 == 0 else int(rng.integers(9, 60))
        lengths = np.exp(rng.uniform(np.log(50), np.log(2000), size=n)).astype(int).tolist()
        for world in (2, 8):
            owner = assign_clips(lengths, world)
            assert len(owner) == n and set(owner) <= set(range(world))
            load = np.bincount(owner, weights=lengths, minlength=world)
            bound = 1.15 if n >= 8 * world else 1.0 + max(lengths) / load.mean()     # few clips: one clip can exceed the mean by itself
            assert load.max() / load.mean() <= bound, (trial, world, load)
            if n == 145 and world == 8:
                rr = np.bincount(np.arange(n) % world, weights=lengths, minlength=world)
                assert load.max() <= rr.max()
    assert assign_clips([100, -1, 100, 0], 2) in ([0, 1, 1, 0], [0, 1, 0, 1], [0, 0, 1, 1], [0, 1, 1, 0])
    assert assign_clips([], 4) == [] and assign_clips([5, 5, 5], 1) == [0, 0, 0]
    assert assign_clips([10, 2000, 30], 2) == [1, 0, 1]                               # the long clip alone, the short ones together


def test_sweep_of_long_and_short_clips_two_ranks_equals_one(tmp_path):
    """clip lengths 3 ... 60 (a 20x spread), batch 4 (tails of 1, 2, 3 frames: the power-of-two padding), two ranks over gloo with the
    frame-balanced assignment and per-clip messages to rank 0: the CSV bytes of the one-process run"""
    import torch.multiprocessing as mp
    root = str(tmp_path / "data")
    os.makedirs(root)
    rng = np.random.default_rng(1)
    lines = []
    for label, lens in (("Shoplifting", [60, 3, 17, 5]), ("Shopping", [9, 41, 6])):
        os.makedirs(os.path.join(root, label), exist_ok=True)
        for k, t in enumerate(lens):
            name = f"{label}{k:03d}_x264"
            clip = np.zeros((t, 240, 320, 3), np.uint8)
            clip[:, 0, 0, 0] = (np.arange(t) * 2 + k) % 200
            clip[:, 0, 1, 0] = (np.arange(t) + 3 * k) % 60
            clip[:, 0, 2, 0] = rng.integers(0, 20, t)
            np.save(os.path.join(root, label, name + ".npy"), clip)
            lines.append(f"{label}/{name}.mp4")
    with open(os.path.join(root, "list.txt"), "w") as f:
        f.write("\n".join(lines))
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    _run(0, 1, 0, root, one)
    ctx = mp.get_context("spawn")
    port = _port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, root, two)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    for fn in ("ucf-crime_dataset.csv", "ucf-crime_dataset-normal.csv"):
        a = open(os.path.join(one, fn), "rb").read()
        assert a == open(os.path.join(two, fn), "rb").read() and len(a) > 0


def test_vectorised_rows_are_the_per_box_reads_of_the_reference_loop():
    """sweep.track_rows_xywhn against `for box in boxes: float(box.id), float(box.xywhn[0][k])` (model.py:56-64) on Results.update's
    clipped boxes: the same float64 values"""
    from cvsd_amd.results import Boxes, clip_boxes
    from cvsd_amd.sweep import pad_bucket, track_rows_xywhn
    rng = np.random.default_rng(2)
    tracks = rng.uniform(-40, 360, size=(50, 8)).astype(np.float32)
    tracks[:, 2:4] = tracks[:, :2] + rng.uniform(1, 120, size=(50, 2)).astype(np.float32)
    tracks[:, 4] = np.arange(1, 51)
    got = track_rows_xywhn(tracks, 17.0, (240, 320))
    b = Boxes(clip_boxes(torch.as_tensor(tracks[:, :-1].copy(), dtype=torch.float32), (240, 320)), (240, 320))
    want = [[17.0, float(box.id), *(float(box.xywhn[0][k]) for k in range(4))] for box in b]
    np.testing.assert_array_equal(got, np.asarray(want, np.float64))
    assert [pad_bucket(n, 64) for n in (1, 2, 3, 5, 10, 33, 63, 64)] == [1, 2, 4, 8, 16, 64, 64, 64] and pad_bucket(5, 4) == 4
