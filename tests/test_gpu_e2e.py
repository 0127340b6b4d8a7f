"""GPU parity of the whole hot path (frames -> rows) against the CPU oracle, through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# Two oracles (DESIGN.md "parity"):
#  * oracle/det.py  -- canonical operation order: the GPU path must match it BIT FOR BIT (tolerance 0), which makes
#    "identical box/class indices" exact and every coordinate identical (well inside north_star's 1e-3).
#  * oracle/yolo_oracle.py -- torch CPU fp32 (oneDNN picks the summation order per host): the GPU path and the
#    canonical oracle both sit within fp32 re-association noise of it.  Through ~60 sequential fp32 convs that noise is
#    ~1e-5 relative on the head logits, i.e. up to ~3e-2 px on a stride-32 DFL box (torch itself differs from a
#    float64 run by 6e-3..1.5e-2 px depending on the host), so 1e-3 cannot be asserted against THIS oracle.
RAW_BOX_TOL = 5e-2       # vs torch oracle, pre-NMS tensor, max over all anchors
RAW_SCORE_TOL = 1e-3


def _model(name, ckpt):
    from cvsd_amd import YOLO
    prog, sd = ckpt
    return YOLO.from_state_dict(name, sd)


@pytest.mark.parametrize("name", ["yolov8n", "yolov8n-pose"])
def test_raw_head_small(name):
    from oracle import yolo_oracle as O
    from tools import synth
    ckpt = synth.synthetic_checkpoint(name, seed=0)
    m = _model(name, ckpt)
    frames = synth.synthetic_frames(3, 64, 96, seed=11)
    got = m.raw_head(frames, imgsz=96)
    om = O.OracleModel(name, ckpt[1])
    want = om.forward(O.preprocess(list(frames), 96)).numpy()
    assert got.shape == want.shape
    nc = om.nc
    assert np.abs(got[:, :4] - want[:, :4]).max() < RAW_BOX_TOL
    assert np.abs(got[:, 4:4 + nc] - want[:, 4:4 + nc]).max() < RAW_SCORE_TOL
    if om.pose:
        assert np.abs(got[:, 4 + nc:] - want[:, 4 + nc:]).max() < RAW_BOX_TOL


def _assert_rows_identical(res, want, pose):
    """bit-exact post-NMS rows: same anchors in the same order, same boxes, scores, classes, keypoints"""
    for r, w in zip(res, want):
        np.testing.assert_array_equal(r.anchor_idx, w["anchor_idx"].numpy())
        np.testing.assert_array_equal(r.boxes.data.numpy(), w["boxes"].numpy())
        if pose and len(r.anchor_idx):
            # Keypoints() zeroes x,y where conf < 0.5 (results.py); apply the same rule to the oracle rows
            k = w["kpts"].clone()
            k[..., :2][k[..., 2] < 0.5] = 0
            np.testing.assert_array_equal(r.keypoints.data.numpy(), k.numpy())


@pytest.mark.parametrize("name,n,size", [("yolov8n", 3, 640), ("yolov8n", 1, 640), ("yolov8n-pose", 4, 640), ("yolov8n-pose", 2, 320)])
def test_bit_exact_vs_canonical_order_oracle(name, n, size):
    from oracle import det
    from tools import synth
    ckpt = synth.synthetic_checkpoint(name, seed=0)
    m = _model(name, ckpt)
    dm = det.DetOracleModel(name, ckpt[1])
    frames = synth.synthetic_frames(n, size, size, seed=21)
    want, pred = det.predict(dm, list(frames), conf=0.25, imgsz=size)
    np.testing.assert_array_equal(m.raw_head(frames, imgsz=size), pred.numpy())       # pre-NMS head tensor
    res = m.predict(frames, conf=0.25, imgsz=size)
    assert sum(len(r) for r in res) > 0
    _assert_rows_identical(res, want, dm.pose)


def test_bit_exact_ucf_crime_shape_with_resize():
    """320x240 clip frames: cv2-style resize to 640x480 + rect letterbox, conf 0.1 and classes=[0] as at model.py:38"""
    from oracle import det
    from tools import synth
    ckpt = synth.synthetic_checkpoint("yolov8n-pose", seed=0)
    m = _model("yolov8n-pose", ckpt)
    dm = det.DetOracleModel("yolov8n-pose", ckpt[1])
    frames = synth.synthetic_frames(3, 240, 320, seed=8)
    want, _ = det.predict(dm, list(frames), conf=0.1, classes=[0])
    res = m.predict(frames, conf=0.1, classes=[0])
    assert res[0].orig_shape == (240, 320)
    _assert_rows_identical(res, want, True)


# float64 decision margins below which a post-NMS divergence between two fp32 implementations is fp32 noise (the same levels
# tests/test_gpu_precision.py uses against the float64 run itself)
MARGIN_NOISE = {"conf threshold": 5e-4, "score order": 5e-4, "iou threshold": 5e-3}


def _compare_predictions(m, om, name, sd, frames, conf, classes=None, max_det=300, imgsz=640):
    """engine rows vs the torch oracle's rows.  A frame whose kept-anchor list differs is NOT skipped: the first divergence
    must sit on a float64 decision margin (score - conf, IoU - 0.7, score order) below fp32 noise -- two fp32 summation
    orders may legitimately decide such a case differently -- or the test fails."""
    from oracle import yolo_oracle as O
    from tools import precision as P
    res = m.predict(frames, conf=conf, classes=classes, max_det=max_det, imgsz=imgsz)
    want, _ = O.predict(om, list(frames), conf=conf, classes=classes, max_det=max_det, imgsz=imgsz)
    stats = {"frames": len(frames), "rows": 0, "max_box_err": 0.0, "max_kpt_err": 0.0, "index_mismatch_frames": 0, "explained": []}
    for i, (r, w) in enumerate(zip(res, want)):
        ga, wa = r.anchor_idx, w["anchor_idx"].numpy()
        if len(ga) != len(wa) or not np.array_equal(ga, wa):
            stats["index_mismatch_frames"] += 1
            ref64 = P.f64_head(name, sd, frames[i:i + 1], imgsz)[0]
            pos, margin, kind = P.first_divergence_margin(ref64, wa.tolist(), ga.tolist(), m.nc, conf, 0.7)
            assert margin < MARGIN_NOISE[kind], (f"{name} frame {i}: kept anchors differ from the torch oracle at rank {pos} although the "
                                                 f"float64 {kind} margin there is {margin:.2e} (not fp32 noise)")
            stats["explained"].append((i, pos, kind, float(f"{margin:.2e}")))
            continue
        stats["rows"] += len(ga)
        if len(ga) == 0:
            continue
        gb, wb = r.boxes.data.numpy(), w["boxes"].numpy()
        assert np.array_equal(gb[:, 5], wb[:, 5])                      # identical class indices
        stats["max_box_err"] = max(stats["max_box_err"], float(np.abs(gb[:, :4] - wb[:, :4]).max()))
        assert np.abs(gb[:, 4] - wb[:, 4]).max() < RAW_SCORE_TOL
        if om.pose:
            k = w["kpts"].clone()
            vis = (k[..., 2] >= 0.5) & torch.from_numpy(r.keypoints.data.numpy()[..., 2] >= 0.5)   # Keypoints() zeroes the rest
            d = np.abs(r.keypoints.data.numpy()[..., :2] - k.numpy()[..., :2])[vis.numpy()]
            if d.size:
                stats["max_kpt_err"] = max(stats["max_kpt_err"], float(d.max()))
    return stats


@pytest.mark.parametrize("name,n", [("yolov8n", 2), ("yolov8n-pose", 4)])
def test_predict_640_matches_oracle(name, n):
    from oracle import yolo_oracle as O
    from tools import synth
    ckpt = synth.synthetic_checkpoint(name, seed=0)
    m = _model(name, ckpt)
    om = O.OracleModel(name, ckpt[1])
    frames = synth.synthetic_frames(n, 640, 640, seed=5)
    st = _compare_predictions(m, om, name, ckpt[1], frames, conf=0.25)
    print(name, st)
    # vs the torch oracle only re-association noise is allowed: every index divergence was explained above by its float64
    # margin (the canonical-order test pins indices exactly); at least one frame must have compared row by row
    assert st["rows"] > 0 and st["index_mismatch_frames"] < n
    assert st["max_box_err"] < RAW_BOX_TOL, st
    assert st["max_kpt_err"] < RAW_BOX_TOL, st


def test_predict_ucf_crime_shape_person_class(v8n):
    """320x240 clips (UCF-Crime) -> rect letterbox 480x640 with a resize; classes=[0] as at model.py:38"""
    from oracle import yolo_oracle as O
    from tools import synth
    m = _model("yolov8n", v8n)
    om = O.OracleModel("yolov8n", v8n[1])
    frames = synth.synthetic_frames(3, 240, 320, seed=2)
    st = _compare_predictions(m, om, "yolov8n", v8n[1], frames, conf=0.1, classes=None)
    print(st)
    assert st["max_box_err"] < RAW_BOX_TOL, st
    res = m.predict(frames, conf=0.1, classes=[0])
    for r in res:
        assert (r.boxes.cls == 0).all()


def test_batch_larger_than_chunk_and_single_frame(v8n):
    """batch looping inside the engine gives the same rows as frame-by-frame calls"""
    from cvsd_amd import YOLO
    from tools import synth
    m = YOLO.from_state_dict("yolov8n", v8n[1], batch_chunk=4)
    frames = synth.synthetic_frames(6, 128, 128, seed=9)
    all_at_once = m.predict(frames, conf=0.1, imgsz=128)
    for i, r in enumerate(all_at_once):
        one = m.predict(frames[i], conf=0.1, imgsz=128)[0]
        np.testing.assert_array_equal(r.boxes.data.numpy(), one.boxes.data.numpy())
        np.testing.assert_array_equal(r.anchor_idx, one.anchor_idx)


def test_rows_written_straight_to_host_equal_the_copied_rows(v8n_pose):
    """Calls of at most 16 frames in one chunk: the greedy NMS kernel writes rows and counts into pinned host memory itself
    (engine.hip: direct_host); larger calls compact on the GPU and copy.  Same frames, both routes: identical rows, keypoint words
    included, on either side of the 16-frame boundary."""
    from cvsd_amd import YOLO
    from tools import synth
    m = YOLO.from_state_dict("yolov8n-pose", v8n_pose[1], batch_chunk=32)
    frames = synth.synthetic_frames(17, 96, 128, seed=21)
    copied = m.predict(frames, conf=0.05, imgsz=128)               # 17 frames: compaction + copies
    direct = m.predict(frames[:16], conf=0.05, imgsz=128)          # 16 frames: straight to host
    singles = [m.predict(frames[i], conf=0.05, imgsz=128)[0] for i in (0, 7, 16)]
    assert sum(len(r.anchor_idx) for r in copied) > 0
    for a, b in zip(copied[:16], direct):
        np.testing.assert_array_equal(a.anchor_idx, b.anchor_idx)
        np.testing.assert_array_equal(a.boxes.data.numpy(), b.boxes.data.numpy())
        np.testing.assert_array_equal(a.keypoints.data.numpy(), b.keypoints.data.numpy())
    for i, one in zip((0, 7, 16), singles):
        np.testing.assert_array_equal(copied[i].boxes.data.numpy(), one.boxes.data.numpy())
        np.testing.assert_array_equal(copied[i].keypoints.data.numpy(), one.keypoints.data.numpy())


def test_device_resident_input(v8n):
    m = _model("yolov8n", v8n)
    from tools import synth
    frames = synth.synthetic_frames(2, 128, 128, seed=4)
    host = m.predict(frames, conf=0.1, imgsz=128)
    dev = m.predict(torch.from_numpy(frames).cuda(), conf=0.1, imgsz=128)
    for a, b in zip(host, dev):
        np.testing.assert_array_equal(a.boxes.data.numpy(), b.boxes.data.numpy())


def test_errors(v8n, tmp_path):
    from cvsd_amd import YOLO
    with pytest.raises(FileNotFoundError):
        YOLO(str(tmp_path / "missing.mi355w"))
    bad = tmp_path / "bad.mi355w"
    bad.write_bytes(b"not a weight file")
    with pytest.raises(ValueError):
        YOLO(str(bad))
    m = _model("yolov8n", v8n)
    with pytest.raises(ValueError):
        m.predict(np.zeros((4, 4), np.uint8))
    empty = m.predict(np.zeros((64, 64, 3), np.uint8), conf=1.0, imgsz=64)[0]      # scores are <= 1 and the test is strict
    assert len(empty.boxes) == 0 and not empty.boxes.is_track


@pytest.mark.parametrize("name", ["yolov8n", "yolov8n-pose"])
def test_engine_matches_committed_golden_vectors(name):
    """tests/golden/golden_v1.npz (written in the build container by tests/golden/make_golden.py from the canonical-order
    oracle): head tensor of 2 small frames and the post-NMS rows of 2 frames at 640x640 -- bit for bit."""
    import os
    from tools import synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
    ckpt = synth.synthetic_checkpoint(name, seed=0)
    m = _model(name, ckpt)
    np.testing.assert_array_equal(m.raw_head(synth.synthetic_frames(2, 64, 96, seed=11), imgsz=96), g[f"head_64x96/{name}"])
    res = m.predict(synth.synthetic_frames(2, 640, 640, seed=21), conf=0.25)
    for i, r in enumerate(res):
        np.testing.assert_array_equal(r.anchor_idx, g[f"rows_640/{name}/{i}/anchors"])
        np.testing.assert_array_equal(r.boxes.data.numpy(), g[f"rows_640/{name}/{i}/boxes"])
        if name.endswith("pose"):
            k = g[f"rows_640/{name}/{i}/kpts"].copy()
            k[..., :2][k[..., 2] < 0.5] = 0
            np.testing.assert_array_equal(r.keypoints.data.numpy(), k)


def test_track_call_of_the_reference(v8n):
    """model.track(frame, persist=True, show=False, classes=[0], verbose=False) -> boxes with ids (model.py:38-45)"""
    from tools import synth
    m = _model("yolov8n", v8n)
    frames = synth.synthetic_frames(3, 240, 320, seed=8)
    seen = 0
    for f in frames:
        boxes = m.track(f, persist=True, show=False, classes=None, verbose=False)[0].boxes
        if boxes.is_track:
            seen += 1
            assert boxes.data.shape[1] == 7 and float(boxes[0].id) >= 1 and boxes.xywhn.shape[1] == 4
    assert seen >= 1


@pytest.mark.parametrize("name,n,h,w,imgsz", [
    ("yolov8s-pose", 2, 320, 320, 320),     # config 4's model
    ("yolov8m", 1, 256, 320, 320),          # config 5's model: widths 48/96/192/384/576, deeper C2f
    ("yolov5nu", 2, 320, 320, 320),         # the reference's literal family (yolov5mu.pt): C3 blocks + 6x6 stem
    ("yolov5mu", 1, 240, 320, 640),         # the reference's literal model AND call: YOLO("yolov5mu.pt") on a 320x240 UCF-Crime
                                            # frame, batch 1 (/root/reference/model.py:18,38): resize to 480x640, C3 x2/4/6/2
    ("yolov8n", 1, 1280, 1280, 1280),       # 33600 anchors: > 32768 sort keys, global-memory bitonic path
    ("yolov8s-pose", 8, 640, 640, 640),     # BASELINE config 4 as stated: its model, its per-GPU batch, full resolution
    ("yolov8m", 1, 640, 640, 640),          # config 5's model at full 640 resolution (fp32 engine)
    ("yolov8m", 2, 1280, 1280, 1280),       # config 5's model, frame size and per-GPU batch in the canonical fp32 arithmetic
])
def test_bit_exact_other_models_and_sizes(name, n, h, w, imgsz):
    from oracle import det
    from tools import synth
    ckpt = synth.synthetic_checkpoint(name, seed=0)
    m = _model(name, ckpt)
    dm = det.DetOracleModel(name, ckpt[1])
    frames = synth.synthetic_frames(n, h, w, seed=31)
    want, pred = det.predict(dm, list(frames), conf=0.25, imgsz=imgsz)
    np.testing.assert_array_equal(m.raw_head(frames, imgsz=imgsz), pred.numpy())
    _assert_rows_identical(m.predict(frames, conf=0.25, imgsz=imgsz), want, dm.pose)
    # low threshold: thousands of candidates per image through sort + greedy NMS, max_det cap
    want, _ = det.predict(dm, list(frames[:1]), conf=0.001, imgsz=imgsz)
    _assert_rows_identical(m.predict(frames[:1], conf=0.001, imgsz=imgsz), want, dm.pose)


def test_shape_switching_reuses_tuning_and_stays_exact(v8n):
    """alternating batch sizes / frame sizes: same rows every time (launch plans are cached per shape, and every
    plan gives the same bits anyway)"""
    from tools import synth
    m = _model("yolov8n", v8n)
    a = synth.synthetic_frames(5, 128, 160, seed=1)
    b = synth.synthetic_frames(2, 96, 96, seed=2)
    ra = [r.boxes.data.numpy().copy() for r in m.predict(a, conf=0.1, imgsz=160)]
    rb = [r.boxes.data.numpy().copy() for r in m.predict(b, conf=0.1, imgsz=96)]
    seen = {}
    for it in range(3):
        for got, want in zip(m.predict(a, conf=0.1, imgsz=160), ra):
            np.testing.assert_array_equal(got.boxes.data.numpy(), want)
        for got, want in zip(m.predict(b, conf=0.1, imgsz=96), rb):
            np.testing.assert_array_equal(got.boxes.data.numpy(), want)
        np.testing.assert_array_equal(m.predict(a[:1], conf=0.1, imgsz=160)[0].boxes.data.numpy(), ra[0])
        # a shape seen before is served from this process's memory (never re-timed): same plan hash, source "memory"
        info = m.plan_info()                       # of the (1 frame, 128x160) shape, first met in the first round
        assert info["plan_source"] in (("memory",) if it else ("tuned", "file", "cache"))
        assert seen.setdefault("hash", info["plan_hash"]) == info["plan_hash"]


def test_full_size_batch_properties(v8n_pose):
    """BASELINE config 3 at full size (YOLOv8n-pose, batch 32, 640x640): size-independent properties.  Rows of a frame
    do not depend on the batch it travels in, on its position, or on the engine's chunking; two frames are
    additionally pinned bit for bit against the canonical-order oracle."""
    from cvsd_amd import YOLO
    from oracle import det
    from tools import synth
    frames = synth.synthetic_frames(32, 640, 640, seed=77)
    m = _model("yolov8n-pose", v8n_pose)
    full = m.predict(frames, conf=0.25)
    assert len(full) == 32 and sum(len(r) for r in full) > 100
    again = m.predict(frames, conf=0.25)                                            # idempotent
    rev = m.predict(frames[::-1].copy(), conf=0.25)                                 # permutation-equivariant
    small = YOLO.from_state_dict("yolov8n-pose", v8n_pose[1], batch_chunk=5)        # 32 = 6 chunks of 5 + a tail of 2
    chunked = small.predict(frames, conf=0.25)
    for i, r in enumerate(full):
        for other in (again[i], rev[31 - i], chunked[i]):
            np.testing.assert_array_equal(r.boxes.data.numpy(), other.boxes.data.numpy())
            np.testing.assert_array_equal(r.keypoints.data.numpy(), other.keypoints.data.numpy())
            np.testing.assert_array_equal(r.anchor_idx, other.anchor_idx)
    for i in (0, 17):
        alone = m.predict(frames[i], conf=0.25)[0]
        np.testing.assert_array_equal(full[i].boxes.data.numpy(), alone.boxes.data.numpy())
    want, _ = det.predict(det.DetOracleModel("yolov8n-pose", v8n_pose[1]), [frames[3], frames[30]], conf=0.25)
    _assert_rows_identical([full[3], full[30]], want, True)
    # boxes are inside the image, rows are confidence-sorted, at most max_det of them
    for r in full:
        b = r.boxes.data.numpy()
        assert len(b) <= 300 and (np.diff(b[:, 4]) <= 0).all()
        assert (b[:, :4] >= 0).all() and (b[:, [0, 2]] <= 640).all() and (b[:, [1, 3]] <= 640).all()


@pytest.mark.parametrize("name,n,size", [("yolov8n", 4, 320), ("yolov8n-pose", 1, 640), ("yolov5nu", 2, 320)])
def test_grouped_launches_do_not_change_a_bit_and_do_not_vary(name, n, size, monkeypatch):
    """the latency-bound regime's step schedule + grouped launches (independent convs of one DAG step as one grid) against
    the plain program-order schedule of the same engine: identical head tensors, identical on every repetition, fewer launches"""
    from cvsd_amd import YOLO
    from tools import synth
    ckpt = synth.synthetic_checkpoint(name, seed=0)
    frames = synth.synthetic_frames(n, size, size, seed=9)
    monkeypatch.setenv("MI355_GROUPS", "0")
    plain = YOLO.from_state_dict(name, ckpt[1], batch_chunk=n)
    ref = plain.raw_head(frames, imgsz=size)
    rows_ref = plain.predict(frames, imgsz=size)
    monkeypatch.setenv("MI355_GROUPS", "1")
    m = YOLO.from_state_dict(name, ckpt[1], batch_chunk=n)
    for _ in range(4):
        np.testing.assert_array_equal(m.raw_head(frames, imgsz=size), ref)
        for a, b in zip(m.predict(frames, imgsz=size), rows_ref):
            np.testing.assert_array_equal(a.boxes.data.numpy(), b.boxes.data.numpy())
    assert m.plan_info()["launches_per_pass"] < plain.plan_info()["launches_per_pass"]
