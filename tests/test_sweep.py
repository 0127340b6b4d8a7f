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
