"""GPU tests of the half=True path (fp16 storage, fp32 arithmetic; BASELINE config 5) through the C ABI.

This mode has no CPU reference run (Ultralytics refuses half on CPU) and is not bit-exact by contract: the MFMA f16
instruction sums a 32-channel block in an order the ISA does not specify, and every stored activation is rounded to
fp16 once.  What IS pinned:
  * one conv against a float64 evaluation of the same fp16-rounded operands: within half an fp16 ulp of rounding plus
    fp32 accumulation noise (fp32 output: 1e-5 relative) -- for every launch plan;
  * the whole net against the oracle's restatement of the same storage contract (oracle/yolo_oracle.py, half=True):
    the engine must be as close to the oracle as two CPU evaluations of the oracle with different fp32 summation
    orders are to each other (self-calibrating tolerance, see _head_noise_and_error);
  * the fp32 engine on the same frames: fp16 rounding noise only.
Fixed tolerances below are measured values on MI355X with a safety margin, written here per north_star.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F16_EPS = 2.0 ** -11                     # half an fp16 ulp, relative
# post-NMS rows of the half engine vs the fp32 engine, matched by source anchor (synthetic weights: wide DFL
# distributions, the worst case for fp16; measured median 0.05 px, worst matched row ~3 px)
# Measured on MI355X over the round-3 runs: median 0.05-0.43 px, worst matched row 3.0-13.1 px, worst score difference 0.035-0.096.
# Bounds = at most 1.5x the worst value seen (bench.py reports the measured figures of the config-5 entries in its JSON line).
# Round 4: the spread of those runs was NOT the launch plans -- every candidate plan of a half=True conv gives the same bits
# (tools/f16_plan_equality.py: 0 of 34-149 plans differ on 9 shapes; test_conv_f16_bits_do_not_depend_on_the_launch_plan below): the
# kernels all add the (k-block, tap) products of one output in the same order, whatever the tile shape, chunking or fusion.  It was the
# store-data hazard fixed at the end of round 3 (common.h:buffer_store_b128), which dropped 16-byte stores in some plans some of the time.
ROW_BOX_MEDIAN_TOL_VS_FP32 = 0.65        # px
ROW_BOX_MAX_TOL_VS_FP32 = 20.0           # px
ROW_SCORE_TOL_VS_FP32 = 0.145


def _f16(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def _ref_conv(x, w, b, stride, silu, res):
    """float64 conv of the fp16-rounded operands, NHWC"""
    import torch
    import torch.nn.functional as F
    xt = torch.from_numpy(_f16(x)).double().permute(0, 3, 1, 2)
    wt = torch.from_numpy(_f16(w)).double()
    y = F.conv2d(xt, wt, torch.from_numpy(b).double(), stride=stride, padding=w.shape[2] // 2)
    if silu:
        y = y * torch.sigmoid(y)
    if res is not None:
        y = y + torch.from_numpy(_f16(res)).double().permute(0, 3, 1, 2)
    return y.permute(0, 2, 3, 1).numpy()


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, silu, residual
    (2, 20, 24, 32, 32, 3, 1, True, True),
    (1, 40, 40, 64, 128, 3, 2, True, False),
    (2, 16, 16, 48, 64, 1, 1, True, False),
    (1, 17, 23, 51, 51, 3, 1, True, False),       # pose keypoint branch: ragged cin / cout, odd image size
    (1, 9, 11, 51, 51, 1, 1, False, False),
    (3, 8, 8, 16, 16, 3, 1, True, True),          # cin 16: half of every 32-channel k-block is padding
    (1, 32, 32, 192, 80, 3, 1, True, False),
    (2, 10, 10, 144, 96, 1, 1, True, False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_f16_every_plan_against_float64(case):
    from cvsd_amd import ops
    n, h, w, cin, cout, k, stride, silu, residual = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, k, k), dtype=np.float32) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    res = rng.standard_normal((n, h // stride, w // stride, cout), dtype=np.float32) if residual else None
    want = _ref_conv(x, wt, b, stride, silu, res)
    y0, n_plans = ops.conv2d(x, wt, b, stride=stride, silu=silu, residual=res, half=True, return_n_plans=True)
    assert n_plans >= 1
    first = None
    for plan in range(n_plans):
        y = ops.conv2d(x, wt, b, stride=stride, silu=silu, residual=res, half=True, plan=plan)
        err = np.abs(y - want)
        assert (err <= 1.01 * F16_EPS * np.abs(want) + 2e-5).all(), (plan, float(err.max()))
        # fp32 output (head finals): no fp16 rounding at all
        if not silu and not residual:
            y32 = ops.conv2d(x, wt, b, stride=stride, silu=silu, half=True, out_f32=True, plan=plan)
            assert np.abs(y32 - want).max() <= 1e-5 * max(1.0, float(np.abs(want).max()))
        # the k-block sum is one instruction and the step order is plan-independent: every plan gives the same bits
        if first is None:
            first = y
        else:
            np.testing.assert_array_equal(y, first)


FUSED_CASES = [
    # n, h, w, cin, c1, stride, c2, silu2, out_f32
    (2, 32, 32, 48, 96, 2, 96, True, False),       # YOLOv8m model.1 -> model.2.cv1
    (1, 24, 40, 96, 192, 2, 192, True, False),     # model.3 -> model.4.cv1
    (2, 20, 20, 64, 64, 1, 64, False, True),       # head box branch [1] -> [2] (fp32 logits)
    (1, 16, 24, 192, 192, 1, 80, False, True),     # head class branch: 12 cout tiles in one block, ragged second stage
    (1, 17, 19, 51, 51, 1, 51, False, True),       # pose keypoint branch: ragged everything, odd image
    (2, 16, 16, 32, 80, 1, 48, True, False),       # c1 = 80: the LDS image is padded to 96 channels
]


@pytest.mark.parametrize("case", [(2, 40, 40, 192, 192, 3, 1, True, True), (1, 80, 80, 48, 96, 3, 2, True, False), (2, 40, 40, 576, 192, 1, 1, True, False),
                                  (1, 17, 23, 51, 51, 3, 1, True, False), (2, 32, 32, 64, 64, 3, 1, True, False), (1, 24, 24, 1152, 576, 1, 1, False, False)])
def test_conv_f16_bits_do_not_depend_on_the_launch_plan(case):
    """the half=True mode is not bit-exact against a CPU run (the f16 MFMA's internal order is unspecified) but it IS reproducible: every
    candidate launch plan -- LDS-staged, streaming and pipelined pointwise kernels, 64- and 128-pixel wave tiles, every chunking -- returns
    the same bits, so what the autotuner's stopwatch picks cannot change a result, on any machine (advisor, round 3)"""
    from cvsd_amd import ops
    n, h, w, cin, cout, k, stride, silu, residual = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, k, k), dtype=np.float32) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    res = rng.standard_normal((n, h // stride, w // stride, cout), dtype=np.float32) if residual else None
    y0, n_plans = ops.conv2d(x, wt, b, stride=stride, silu=silu, residual=res, half=True, return_n_plans=True)
    assert n_plans >= 8
    for plan in range(1, n_plans):
        y = ops.conv2d(x, wt, b, stride=stride, silu=silu, residual=res, half=True, plan=plan)
        assert np.array_equal(y, y0), f"plan {plan} of {n_plans} differs from plan 0 by up to {np.abs(y - y0).max():.3e}"


def test_half_engine_is_reproducible_across_independently_tuned_instances(v8n, tmp_path, monkeypatch):
    """two engines of one model that each time their own launch plans (separate plan caches) return the same head tensor and rows"""
    from cvsd_amd import YOLO
    from tools import synth
    frames = synth.synthetic_frames(3, 640, 640, seed=77)
    outs = []
    for k in range(2):
        monkeypatch.setenv("MI355_PLAN_CACHE", str(tmp_path / f"cache{k}"))
        m = YOLO.from_state_dict("yolov8n", v8n[1], half=True, batch_chunk=3, plan_dir="")
        outs.append((m.raw_head(frames), [r.boxes.data.numpy() for r in m.predict(frames)], m.plan_info()))
    assert outs[0][2]["plan_source"] == outs[1][2]["plan_source"] == "tuned"
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("case", FUSED_CASES)
def test_fused_conv3x3_conv1x1_f16_equals_the_two_launches(case):
    """every fused launch plan == conv3x3 (stored as fp16) followed by conv1x1, bit for bit: the k-block sum is one MFMA and the
    block order is the same, so fusing moves no rounding"""
    from cvsd_amd import ops
    n, h, w, cin, c1, stride, c2, silu2, out_f32 = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    w1 = (rng.standard_normal((c1, cin, 3, 3), dtype=np.float32) / np.sqrt(cin * 9)).astype(np.float32)
    b1 = rng.standard_normal(c1).astype(np.float32) * 0.1
    w2 = (rng.standard_normal((c2, c1, 1, 1), dtype=np.float32) / np.sqrt(c1)).astype(np.float32)
    b2 = rng.standard_normal(c2).astype(np.float32) * 0.1
    mid = ops.conv2d(x, w1, b1, stride=stride, silu=True, half=True)
    want = ops.conv2d(mid, w2, b2, stride=1, silu=silu2, half=True, out_f32=out_f32)
    y0, n_plans = ops.conv2d_fused(x, w1, b1, w2, b2, stride=stride, silu2=silu2, half=True, out_f32=out_f32, return_n_plans=True)
    assert n_plans >= 1
    for plan in range(n_plans):
        y = ops.conv2d_fused(x, w1, b1, w2, b2, stride=stride, silu2=silu2, half=True, out_f32=out_f32, plan=plan)
        np.testing.assert_array_equal(y.view(np.uint32), want.view(np.uint32), err_msg=f"plan {plan}")


def _half_model(name, ckpt, **kw):
    from cvsd_amd import YOLO
    return YOLO.from_state_dict(name, ckpt[1], half=True, **kw)


def _q(d):
    d = np.abs(d).ravel()
    return np.array([np.median(d), np.quantile(d, 0.99), d.max()])


def _head_noise_and_error(name, n, size, seed=31):
    """(intrinsic noise of the storage contract, error of the GPU engine), each as (median, p99, max) per column group.

    The contract fixes WHERE values are rounded to fp16, not the order of the fp32 sums in between; two CPU evaluations
    of it that only differ in that order (oneDNN vs torch's native conv) already disagree wherever a sum lands next to
    an fp16 rounding boundary, and the disagreement is amplified through the remaining layers.  That distance is the
    yardstick: the engine has to sit as close to the oracle as the oracle sits to itself."""
    import torch
    from oracle import yolo_oracle as O
    from tools import synth
    ckpt = synth.synthetic_checkpoint(name, seed=0)
    m = _half_model(name, ckpt)
    frames = synth.synthetic_frames(n, size, size, seed=seed)
    got = m.raw_head(frames, imgsz=size)
    om = O.OracleModel(name, ckpt[1], half=True)
    x = O.preprocess(list(frames), size)
    a = om.forward(x).numpy()
    with torch.backends.mkldnn.flags(enabled=False):
        b = om.forward(x).numpy()
    assert got.shape == a.shape
    nc = om.nc
    groups = {"box": slice(0, 4), "score": slice(4, 4 + nc)}
    if om.pose:
        groups["kpt"] = slice(4 + nc, None)
    return {g: (_q(a[:, sl] - b[:, sl]), _q(got[:, sl] - a[:, sl])) for g, sl in groups.items()}


@pytest.mark.parametrize("name,n,size", [("yolov8n", 2, 640), ("yolov8n-pose", 2, 640), ("yolov8s-pose", 1, 320),
                                         ("yolov8m", 1, 320)])
def test_raw_head_within_the_contracts_own_noise(name, n, size):
    for group, (noise, err) in _head_noise_and_error(name, n, size).items():
        # measured on MI355X: err / noise = 0.95 .. 1.05 at the median and p99, 0.6 .. 1.5 at the max (one worst anchor)
        assert err[0] <= 1.5 * noise[0] + 1e-6, (group, "median", err, noise)
        assert err[1] <= 1.5 * noise[1] + 1e-5, (group, "p99", err, noise)
        assert err[2] <= 4.0 * noise[2] + 1e-4, (group, "max", err, noise)


def _match_rows(a, b):
    """rows of two Results matched by source anchor -> (pairs, only_a, only_b)"""
    ia = {int(x): i for i, x in enumerate(a.anchor_idx)}
    ib = {int(x): i for i, x in enumerate(b.anchor_idx)}
    common = sorted(set(ia) & set(ib))
    return [(ia[k], ib[k]) for k in common], len(ia) - len(common), len(ib) - len(common)


@pytest.mark.parametrize("name", ["yolov8n", "yolov8n-pose"])
def test_predict_half_close_to_fp32_engine(name):
    """model(frames, half=True) on an fp32-constructed model: same facade call as Ultralytics; rows stay close to fp32"""
    from cvsd_amd import YOLO
    from tools import synth
    ckpt = synth.synthetic_checkpoint(name, seed=0)
    m = YOLO.from_state_dict(name, ckpt[1])
    frames = synth.synthetic_frames(4, 640, 640, seed=41)
    r32 = m.predict(frames, conf=0.25)
    r16 = m.predict(frames, conf=0.25, half=True)
    assert sum(len(r) for r in r32) > 0
    matched = total = cls_flips = 0
    box_err = []
    for a, b in zip(r32, r16):
        pairs, only_a, only_b = _match_rows(a, b)
        total += len(pairs) + only_a + only_b
        matched += len(pairs)
        for i, j in pairs:
            da, db = a.boxes.data.numpy()[i], b.boxes.data.numpy()[j]
            box_err.append(np.abs(da[:4] - db[:4]).max())
            assert abs(da[4] - db[4]) < ROW_SCORE_TOL_VS_FP32
            cls_flips += int(da[5] != db[5])        # near-tied class scores of the synthetic head may swap their argmax
    print(f"[half] {name}: rows matched {matched}/{total}, class flips {cls_flips}, box |delta| median {np.median(box_err):.3e} max {max(box_err):.3e} px")
    assert np.median(box_err) < ROW_BOX_MEDIAN_TOL_VS_FP32 and max(box_err) < ROW_BOX_MAX_TOL_VS_FP32, (np.median(box_err), max(box_err))
    # detections whose score sits at the threshold or whose IoU with a neighbour sits at 0.7 may flip; the bulk may not
    assert matched >= 0.85 * total, (matched, total)
    assert cls_flips <= 0.05 * matched, (cls_flips, matched)
    # the fp32 engine is untouched by the half one living in the same object
    r32b = m.predict(frames, conf=0.25, half=False)
    for a, b in zip(r32, r32b):
        np.testing.assert_array_equal(a.boxes.data.numpy(), b.boxes.data.numpy())


def test_half_engine_batches_chunks_and_device_frames(v8n):
    import torch
    from tools import synth
    m = _half_model("yolov8n", v8n, batch_chunk=3)
    frames = synth.synthetic_frames(7, 320, 320, seed=5)
    whole = m.predict(frames, conf=0.25, imgsz=320)                       # 3 chunks, the last one ragged
    single = [m.predict(frames[i:i + 1], conf=0.25, imgsz=320)[0] for i in range(7)]
    dev = m.predict(torch.from_numpy(frames).cuda(), conf=0.25, imgsz=320)
    for a, b, c in zip(whole, single, dev):
        np.testing.assert_array_equal(a.boxes.data.numpy(), b.boxes.data.numpy())     # batch-size independent bits
        np.testing.assert_array_equal(a.boxes.data.numpy(), c.boxes.data.numpy())


def test_config5_yolov8m_1280_half():
    """BASELINE config 5 at a test-sized batch: YOLOv8m, 1280x1280, half -- same self-calibrating bound."""
    res = _head_noise_and_error("yolov8m", 1, 1280, seed=3)
    for group, (noise, err) in res.items():
        assert np.isfinite(err).all()
        assert err[0] <= 1.5 * noise[0] + 1e-6, (group, "median", err, noise)
        assert err[1] <= 1.5 * noise[1] + 1e-5, (group, "p99", err, noise)
        assert err[2] <= 4.0 * noise[2] + 1e-4, (group, "max", err, noise)


def test_head_final_conv_every_plan_repeated_has_no_dropped_store_data():
    """Regression (round 3): a head's final conv -- pointwise, no activation, fp32 output -- computes the next tile's bias add two
    instructions behind a 16-byte buffer store.  On gfx950 the store's data registers may not be rewritten by the next vector
    instruction even when the store takes its offset from an SGPR (the form the compiler's hazard recogniser exempts), and 20 of the
    95 launch plans of this shape stored zeros / the next tile's bits in dword 0 of pixel lanes 12-15, in a few per cent of the
    launches.  common.h:buffer_store_b128 inserts the wait states; every plan, three times, against a float64 reference."""
    from cvsd_amd import ops
    n, h, w, cin, cout = 1, 160, 160, 192, 80
    rng = np.random.default_rng(5)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, 1, 1), dtype=np.float32) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    want = np.einsum("nhwc,oc->nhwo", x.astype(np.float16).astype(np.float64), wt.astype(np.float16).astype(np.float64)[:, :, 0, 0]) + b
    _, n_plans = ops.conv2d(x, wt, b, stride=1, silu=False, half=True, out_f32=True, return_n_plans=True)
    tol = 1e-4 * max(1.0, float(np.abs(want).max()))
    for rep in range(3):
        for plan in range(n_plans):
            y = ops.conv2d(x, wt, b, stride=1, silu=False, half=True, out_f32=True, plan=plan)
            bad = np.argwhere(np.abs(y - want) > tol)
            assert len(bad) == 0, (rep, plan, len(bad), bad[:4].tolist())


def test_config5_at_its_stated_batch_of_two_equals_frame_by_frame():
    """BASELINE config 5 as stated: YOLOv8m, 1280x1280, half, TWO frames per GPU -- the batch-2 launch plans (other tiles, fused
    pairs, merged head convs) give the rows of the one-frame calls whose head tensor the test above bounds, bit for bit."""
    from tools import synth
    ckpt = synth.synthetic_checkpoint("yolov8m", seed=0)
    m = _half_model("yolov8m", ckpt, batch_chunk=2)
    frames = synth.synthetic_frames(2, 1280, 1280, seed=3)
    both = m.predict(frames, conf=0.25, imgsz=1280)
    for i in range(2):
        one = m.predict(frames[i], conf=0.25, imgsz=1280)[0]
        np.testing.assert_array_equal(both[i].anchor_idx, one.anchor_idx)
        np.testing.assert_array_equal(both[i].boxes.data.numpy(), one.boxes.data.numpy())
    assert sum(len(r.anchor_idx) for r in both) > 0


def test_reference_call_sites_with_half(v8n_pose):
    """the reference's own calls on the half engine: .track(frame, persist=True, classes=[0]) (model.py:38) and
    model(frame).keypoints on a UCF-Crime-shaped frame (resize + letterbox path)"""
    from tools import synth
    m = _half_model("yolov8n-pose", v8n_pose)
    frames = synth.synthetic_frames(3, 240, 320, seed=8)
    seen = 0
    for f in frames:
        res = m.track(f, persist=True, show=False, classes=[0], verbose=False)[0]
        assert res.orig_shape == (240, 320)
        if res.boxes.is_track:
            seen += 1
            assert res.boxes.data.shape[1] == 7 and float(res.boxes[0].id) >= 1
            xywhn = res.boxes.xywhn.numpy()
            assert (xywhn >= 0).all() and (xywhn <= 1).all()
    assert seen >= 1
    r = m(frames[0], conf=0.1)[0]
    assert r.keypoints.data.shape[1:] == (17, 3) and len(r.keypoints) == len(r.boxes)


@pytest.mark.parametrize("n,h,w,cu,cs,cout", [(2, 40, 40, 576, 384, 384), (1, 80, 80, 384, 192, 192), (2, 20, 24, 64, 32, 96), (1, 24, 20, 128, 72, 80)])
def test_half_pointwise_conv_reading_through_the_upsample_every_plan(n, h, w, cu, cs, cout):
    """Round 4: the neck's Upsample -> Concat -> C2f.cv1 in half mode with the upsample fused into the conv's read side -- the pipelined kernel and
    the LDS-weights kernel (conv1x1_lwx_f16<.., UP>: whole 64-channel X-chunks staged from the half-resolution map).  Every plan must give
    the bits of the plain half-mode conv on the materialised concatenation (same fp16-rounded operands, same accumulation order); the concat
    buffer's up channels are poisoned with NaN by the op entry."""
    from cvsd_amd import ops
    rng = np.random.default_rng(cu + cs + cout + h)
    xh = rng.standard_normal((n, h // 2, w // 2, cu), dtype=np.float32)
    xs = rng.standard_normal((n, h, w, cs), dtype=np.float32)
    wt = (rng.standard_normal((cout, cu + cs, 1, 1)) / np.sqrt(cu + cs)).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    cat = np.concatenate([xh.repeat(2, axis=1).repeat(2, axis=2), xs], axis=3)
    ref = ops.conv2d(cat, wt, b, stride=1, silu=True, half=True)
    y, n_plans = ops.conv1x1_upcat(xh, xs, wt, b, silu=True, half=True, return_n_plans=True)
    np.testing.assert_array_equal(y, ref)
    assert n_plans >= 2
    for plan in range(1, n_plans):
        np.testing.assert_array_equal(ops.conv1x1_upcat(xh, xs, wt, b, silu=True, half=True, plan=plan), ref, err_msg=f"plan {plan} of {n_plans}")


STEM_CASES = [
    # n, h, w, cout, k
    (2, 64, 96, 16, 3),           # yolov8n
    (1, 96, 64, 48, 3),           # yolov8m (config 5)
    (1, 34, 36, 32, 3),           # ragged tiles: 17 x 18 output pixels
    (1, 34, 38, 32, 3),           # a width that is not a multiple of 4: general kernel only
    (1, 32, 32, 80, 3),           # yolov8x
    (1, 32, 64, 18, 3),           # cout that is not a multiple of four: the element-wise tail of the store
    (1, 64, 64, 48, 6),           # yolov5u's 6 x 6 stem: general kernel only
]


@pytest.mark.parametrize("case", STEM_CASES)
def test_half_stem_both_kernels_against_float64(case):
    """The stem under half=True (engine/predictor.py:preprocess with half -> model.0): (u8 / 255) and the weights rounded to fp16,
    products exact, fp32 sum, one rounding on the store.  The k 3 / stride 2 kernel sums the 27 products in a different order
    from the general one (misc_kernels.hip:stem3s2_u8_h); both must sit within half an fp16 ulp plus fp32 accumulation noise of
    a float64 evaluation of the same rounded operands."""
    import torch
    import torch.nn.functional as F
    from cvsd_amd import ops
    n, h, w, cout, k = case
    rng = np.random.default_rng(sum(case))
    img = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
    img[0, :3, :5] = 0
    img[0, -2:, -7:] = 255
    wt = (rng.standard_normal((cout, 3, k, k)) * 0.4).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.2).astype(np.float32)
    x = _f16(img[..., ::-1].astype(np.float32) / np.float32(255.0))
    y = F.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(_f16(wt)).double(), torch.from_numpy(b).double(),
                 stride=2, padding=2 if k == 6 else 1)
    ref = (y * torch.sigmoid(y)).permute(0, 2, 3, 1).numpy()
    mag = np.abs(ref) + 1e-3
    got = {}
    lean = k == 3 and w % 4 == 0
    for variant in ((0, 1, 2) if lean else (0, 1)):
        out = ops.stem(img, wt, b, stride=2, half=True, variant=variant)
        assert out.dtype == np.float16 and out.shape == ref.shape
        err = np.abs(out.astype(np.float64) - ref) / mag
        assert err.max() < F16_EPS * 1.02 + 2e-5, f"variant {variant}: {err.max():.3e}"
        got[variant] = out
    if lean:
        # the two kernels agree except where the different summation order moves a value across an fp16 rounding boundary
        d = got[1].astype(np.float32) != got[2].astype(np.float32)
        assert d.mean() < 2e-3
    with pytest.raises(Exception):
        ops.stem(img, wt, b, stride=2, half=True, variant=3 if lean else 2)
