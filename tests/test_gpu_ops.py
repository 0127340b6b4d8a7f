"""GPU parity of single HIP kernels against the oracle / plain torch fp32, through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_conv(x, w, b, stride, silu, res):
    y = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w), torch.from_numpy(b), stride=stride,
                 padding=w.shape[2] // 2)
    if silu:
        y = F.silu(y)
    y = y.permute(0, 2, 3, 1).contiguous().numpy()
    return y + res if res is not None else y


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, silu, residual
    (2, 20, 20, 16, 16, 3, 1, True, True),
    (1, 40, 40, 32, 64, 3, 2, True, False),
    (2, 16, 24, 64, 80, 3, 1, True, False),
    (1, 80, 80, 64, 64, 3, 1, True, True),
    (2, 20, 20, 256, 128, 1, 1, True, False),
    (1, 40, 40, 384, 128, 1, 1, True, False),
    (3, 20, 20, 51, 51, 3, 1, True, False),       # pose kpt branch: channel counts not multiples of 4
    (2, 20, 20, 51, 51, 1, 1, False, False),
    (2, 40, 40, 64, 1, 1, 1, False, False),       # pose cls output (nc = 1)
    (1, 20, 20, 80, 80, 1, 1, False, False),
    (1, 32, 32, 48, 96, 3, 2, True, False),       # v8m widths
    (1, 20, 20, 128, 256, 3, 2, True, False),
    (1, 8, 8, 576, 576, 3, 1, True, True),
    (5, 12, 20, 32, 32, 3, 1, True, False),       # ragged: tile does not divide the map
    (1, 160, 160, 16, 32, 3, 2, True, False),
]


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride,silu,residual", CONV_CASES)
def test_conv2d_matches_torch(n, h, w, cin, cout, k, stride, silu, residual):
    from cvsd_amd import ops
    rng = np.random.default_rng(cin * 1000 + cout + k)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((n, h // stride, w // stride, cout), dtype=np.float32) if residual else None
    y = ops.conv2d(x, wt, b, stride=stride, silu=silu, residual=res)
    ref = _ref_conv(x, wt, b, stride, silu, res)
    assert y.shape == ref.shape
    # fp32 with a different summation order: tolerance 2e-5 absolute on O(1) outputs
    np.testing.assert_allclose(y, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride,silu,residual", CONV_CASES)
def test_conv2d_bit_exact_vs_canonical_order_oracle(n, h, w, cin, cout, k, stride, silu, residual):
    """The fp32 MFMA is a k-ordered fma chain: against the C oracle that states the same order (and the same
    libm-free exp) the conv output must be identical bit for bit."""
    from cvsd_amd import ops
    from oracle import det
    rng = np.random.default_rng(cin * 1000 + cout + k + 1)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((n, h // stride, w // stride, cout), dtype=np.float32) if residual else None
    y = ops.conv2d(x, wt, b, stride=stride, silu=silu, residual=res)
    ref = det.conv2d(x, wt, b, stride=stride, act=silu, residual=res)
    np.testing.assert_array_equal(y, ref)


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride", [
    (2, 24, 40, 64, 64, 3, 1), (1, 40, 40, 80, 80, 3, 1), (2, 16, 16, 128, 256, 3, 2), (3, 20, 20, 384, 128, 1, 1),
    (1, 32, 48, 32, 16, 3, 1), (2, 20, 20, 51, 51, 3, 1), (1, 64, 64, 16, 32, 3, 2), (2, 12, 12, 192, 96, 1, 1),
    # batch-1 shapes: latency-bound, so the candidates include the small wave tiles and the split-K kernel (v6)
    (1, 20, 20, 256, 80, 3, 1), (1, 20, 20, 128, 128, 3, 2), (1, 10, 14, 320, 48, 3, 1), (1, 40, 40, 64, 64, 3, 1),
])
def test_every_launch_plan_gives_the_same_bits(n, h, w, cin, cout, k, stride):
    """All candidate plans (tile shapes, wave arrangements and wave-tile sizes, LDS-staged / streamed / pipelined / split-K
    kernels, staged channel counts) must reproduce the canonical-order oracle bit for bit: the autotuner may pick any."""
    from cvsd_amd import ops
    from oracle import det
    rng = np.random.default_rng(cin + cout + k + h)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((n, h // stride, w // stride, cout), dtype=np.float32)
    ref = det.conv2d(x, wt, b, stride=stride, act=True, residual=res)
    y, n_plans = ops.conv2d(x, wt, b, stride=stride, silu=True, residual=res, plan=0, return_n_plans=True)
    np.testing.assert_array_equal(y, ref)
    assert n_plans >= 4
    for plan in range(1, n_plans):
        np.testing.assert_array_equal(ops.conv2d(x, wt, b, stride=stride, silu=True, residual=res, plan=plan), ref,
                                      err_msg=f"plan {plan} of {n_plans}")


@pytest.mark.parametrize("n,h,w,cu,cs,cout", [(1, 40, 40, 256, 128, 128), (1, 80, 80, 128, 64, 64), (2, 20, 24, 32, 16, 48), (3, 12, 10, 48, 51, 80),
                                              (16, 40, 40, 256, 128, 128)])
def test_pointwise_conv_reading_through_the_upsample_every_plan(n, h, w, cu, cs, cout):
    """The neck's Upsample -> Concat -> C2f.cv1 (ultralytics yolov8.yaml head, layers 10-12 / 13-15) as the engine runs it: the conv reads
    the up channels from the half-resolution map at (y >> 1, x >> 1); the concat buffer's up channels are never written (the op entry
    poisons them with NaN).  Every plan -- the pipelined kernels and, round 4, the LDS-free streaming kernel that latency-bound passes
    use -- against the canonical-order oracle on the materialised concatenation, bit for bit."""
    from cvsd_amd import ops
    from oracle import det
    rng = np.random.default_rng(cu + cs + cout + h)
    xh = rng.standard_normal((n, h // 2, w // 2, cu), dtype=np.float32)
    xs = rng.standard_normal((n, h, w, cs), dtype=np.float32)
    wt = (rng.standard_normal((cout, cu + cs, 1, 1)) / np.sqrt(cu + cs)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    cat = np.concatenate([xh.repeat(2, axis=1).repeat(2, axis=2), xs], axis=3)
    ref = det.conv2d(cat, wt, b, stride=1, act=True, residual=None)
    y, n_plans = ops.conv1x1_upcat(xh, xs, wt, b, silu=True, return_n_plans=True)
    np.testing.assert_array_equal(y, ref)
    assert n_plans >= 2
    for plan in range(1, n_plans):
        np.testing.assert_array_equal(ops.conv1x1_upcat(xh, xs, wt, b, silu=True, plan=plan), ref, err_msg=f"plan {plan} of {n_plans}")


@pytest.mark.parametrize("n,h,w,cin,cout", [(1, 80, 80, 64, 80), (1, 160, 160, 64, 64), (2, 40, 40, 128, 80)])
def test_head_final_conv_without_activation_every_plan_repeated(n, h, w, cin, cout):
    """Regression (round 3, common.h:buffer_store_b128): a head's final conv -- pointwise, NO activation -- computes the next tile's
    bias add right behind a 16-byte buffer store; on gfx950 that rewrote the store's data registers too early in a few per cent of
    the launches of some plans (zeros / the next tile's bits in dword 0 of pixel lanes 12-15; first seen as the streaming member of
    a grouped launch).  Every plan, three times, bit for bit against the canonical-order oracle."""
    from cvsd_amd import ops
    from oracle import det
    rng = np.random.default_rng(cin + cout + h)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = det.conv2d(x, wt, b, stride=1, act=False, residual=None)
    _, n_plans = ops.conv2d(x, wt, b, stride=1, silu=False, plan=0, return_n_plans=True)
    for rep in range(3):
        for plan in range(n_plans):
            np.testing.assert_array_equal(ops.conv2d(x, wt, b, stride=1, silu=False, plan=plan), ref, err_msg=f"rep {rep} plan {plan} of {n_plans}")


@pytest.mark.parametrize("n,h,w,cin,c1,stride,c2,silu2", [
    (2, 32, 32, 16, 32, 2, 32, True),        # model.1 -> model.2.cv1
    (1, 40, 40, 64, 64, 1, 64, False),       # cv2[i][1] -> cv2[i][2] (box logits, no activation)
    (1, 24, 40, 80, 80, 1, 80, False),       # cv3[i][1] -> cv3[i][2]: 5 cout tiles, 80 classes
    (2, 20, 20, 51, 51, 1, 51, False),       # cv4[i][1] -> cv4[i][2]: ragged channel counts on both convs
    (1, 20, 20, 64, 64, 1, 1, False),        # pose cls output (nc = 1)
    (1, 16, 16, 128, 256, 2, 256, True),     # model.7 -> model.8.cv1: two staged chunks, 16 cout tiles
    (1, 20, 20, 64, 128, 2, 128, True),      # batch-1 shape: small wave tiles among the candidates
])
def test_fused_conv_pair_equals_the_two_convs_for_every_plan(n, h, w, cin, c1, stride, c2, silu2):
    """Conv3x3+SiLU -> Conv1x1 as one launch (first conv's output image stays in LDS) against the canonical-order oracle's
    two separate convs: identical bits for every candidate launch plan (tile shapes, cout groups, wave-tile sizes)."""
    from cvsd_amd import ops
    from oracle import det
    rng = np.random.default_rng(cin * 7 + c1 + c2 + stride)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    w1 = (rng.standard_normal((c1, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b1 = rng.standard_normal(c1).astype(np.float32)
    w2 = (rng.standard_normal((c2, c1, 1, 1)) / np.sqrt(c1)).astype(np.float32)
    b2 = rng.standard_normal(c2).astype(np.float32)
    ref = det.conv2d(det.conv2d(x, w1, b1, stride=stride, act=True), w2, b2, stride=1, act=silu2)
    y, n_plans = ops.conv2d_fused(x, w1, b1, w2, b2, stride=stride, silu2=silu2, plan=0, return_n_plans=True)
    np.testing.assert_array_equal(y, ref)
    assert n_plans >= 2
    for plan in range(1, n_plans):
        np.testing.assert_array_equal(ops.conv2d_fused(x, w1, b1, w2, b2, stride=stride, silu2=silu2, plan=plan), ref,
                                      err_msg=f"fused plan {plan} of {n_plans}")


def test_silu_epilogue_is_the_oracles_ieee_division_over_a_dense_sweep():
    """The kernels' SiLU divides with a reciprocal-refinement sequence (detmath.h:det_div_ge1: 8 instructions, no range scaling or
    fix-up); the oracle divides with C's `/`.  An identity 1x1 conv (weight 1, bias 0) makes the conv epilogue evaluate silu(x)
    on any x: 4M values -- a dense grid over [-100, 100], the clamp seams, tiny / huge magnitudes, random bit patterns of finite
    floats -- must give the oracle's bits, i.e. the correctly rounded quotient.  (-0.0 is not in the sweep: a conv output is
    `(+0 + partial sums) + bias` and cannot be -0; the refinement sequence would return +0 for it.)"""
    from cvsd_amd import ops
    from oracle import det
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2 ** 32, 1_500_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    bits = bits[np.isfinite(bits)]
    x = np.concatenate([
        np.linspace(-100.0, 100.0, 2_000_001),
        np.linspace(-87.5, -86.5, 200_001), np.linspace(86.5, 88.5, 200_001),
        rng.normal(0, 4, 500_000), rng.normal(0, 1e-3, 100_000), rng.normal(0, 1e4, 100_000),
        [0.0, 1e-45, -1e-45, 1.1754944e-38, -1.1754944e-38, 3.0e38, -3.0e38, 87.0, -87.0, 87.25, -87.25, 88.0, -88.0],
        bits.astype(np.float64)]).astype(np.float32)
    n = (x.size // 1024) * 1024
    x = np.ascontiguousarray(x[:n])
    want = np.empty_like(x)
    det.lib().det_silu_array(x.ctypes.data, want.ctypes.data, x.size)
    got = ops.conv2d(x.reshape(1, n // 1024, 1024, 1), np.ones((1, 1, 1, 1), np.float32), np.zeros(1, np.float32), stride=1, silu=True)
    got = got.reshape(-1)
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
    assert bad.size == 0, (bad.size, x[bad[:5]], got[bad[:5]], want[bad[:5]])


def test_conv2d_asymmetric_identity():
    """A = I check with an asymmetric operand (catches a transposed MFMA fragment map)."""
    from cvsd_amd import ops
    cin = cout = 32
    x = np.arange(2 * 4 * 4 * cin, dtype=np.float32).reshape(2, 4, 4, cin) % 97
    w = np.zeros((cout, cin, 1, 1), np.float32)
    for o in range(cout):
        w[o, (o * 7 + 3) % cin, 0, 0] = 1.0          # a permutation, not symmetric
    y = ops.conv2d(x, w, np.zeros(cout, np.float32), silu=False)
    ref = x[..., [(o * 7 + 3) % cin for o in range(cout)]]
    np.testing.assert_array_equal(y, ref)


@pytest.mark.parametrize("k,cout,h,w", [(3, 16, 64, 96), (3, 48, 32, 32), (6, 16, 64, 64), (3, 32, 640, 640)])
def test_stem_matches_torch(k, cout, h, w):
    from cvsd_amd import ops
    rng = np.random.default_rng(k + cout)
    img = rng.integers(0, 256, size=(2, h, w, 3), dtype=np.uint8)
    wt = (rng.standard_normal((cout, 3, k, k)) / np.sqrt(3 * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    y = ops.stem(img, wt, b, stride=2)
    x = torch.from_numpy(np.ascontiguousarray(img[..., ::-1].transpose(0, 3, 1, 2))).float()
    x /= 255
    ref = F.silu(F.conv2d(x, torch.from_numpy(wt), torch.from_numpy(b), stride=2, padding=2 if k == 6 else 1))
    np.testing.assert_allclose(y, ref.permute(0, 2, 3, 1).numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("k,cout,h,w", [(3, 16, 64, 96), (3, 48, 32, 32), (6, 16, 64, 64), (3, 32, 640, 640), (3, 16, 34, 36), (3, 18, 34, 38),
                                       (3, 80, 32, 64)])
def test_stem_bit_exact_vs_canonical_order_oracle(k, cout, h, w, monkeypatch):
    """Both stem kernels (misc_kernels.hip: stem3s2_u8_f32 for k 3 / stride 2 on widths that are multiples of 4, stem_mfma_u8 for the
    rest and under MI355_STEM_LEAN=0) against the oracle's fma chain; ragged tiles, a width of 38 and a ragged cout among the shapes."""
    from cvsd_amd import ops
    from oracle import det
    rng = np.random.default_rng(k + cout + 7)
    img = rng.integers(0, 256, size=(2, h, w, 3), dtype=np.uint8)
    wt = (rng.standard_normal((cout, 3, k, k)) / np.sqrt(3 * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    want = det.stem(img, wt, b, stride=2)
    np.testing.assert_array_equal(ops.stem(img, wt, b, stride=2), want)
    monkeypatch.setenv("MI355_STEM_LEAN", "0")
    np.testing.assert_array_equal(ops.stem(img, wt, b, stride=2), want)


@pytest.mark.parametrize("h,w", [(240, 320), (720, 1280), (480, 640), (100, 37), (640, 640), (1080, 1920)])
def test_letterbox_bit_exact(h, w):
    """integer work: bit-exact against the oracle's cv2.resize/LetterBox restatement"""
    from cvsd_amd import ops
    from oracle import yolo_oracle as O
    rng = np.random.default_rng(h + w)
    img = rng.integers(0, 256, size=(2, h, w, 3), dtype=np.uint8)
    out = ops.letterbox(img, 640)
    ref = np.stack([O.letterbox(f, (640, 640)) for f in img])
    assert out.shape == ref.shape
    np.testing.assert_array_equal(out, ref)


def _random_pred(rng, n, nc, extra, a, frac=0.05):
    pred = np.zeros((n, 4 + nc + extra, a), np.float32)
    pred[:, 0] = rng.uniform(0, 640, (n, a))
    pred[:, 1] = rng.uniform(0, 640, (n, a))
    pred[:, 2] = rng.uniform(8, 300, (n, a))
    pred[:, 3] = rng.uniform(8, 300, (n, a))
    sc = rng.uniform(0, 0.2, (n, nc, a)).astype(np.float32)
    hot = rng.random((n, a)) < frac
    cls = rng.integers(0, nc, (n, a))
    for i in range(n):
        idx = np.nonzero(hot[i])[0]
        sc[i, cls[i, idx], idx] = rng.uniform(0.2, 1.0, len(idx))
    pred[:, 4:4 + nc] = sc
    if extra:
        pred[:, 4 + nc:] = rng.standard_normal((n, extra, a))
    return pred


@pytest.mark.parametrize("nc,extra,a,frac,classes,max_det", [
    (80, 0, 8400, 0.05, None, 300),
    (1, 51, 8400, 0.10, None, 300),
    (80, 0, 8400, 0.30, [0, 3, 17], 300),
    (3, 0, 2100, 1.00, None, 300),        # every anchor is a candidate; max_det truncation
    (80, 0, 8400, 0.0, None, 300),        # no candidates at all
    (2, 4, 525, 0.5, [1], 50),
])
def test_nms_matches_oracle(nc, extra, a, frac, classes, max_det):
    from cvsd_amd import ops
    from oracle import yolo_oracle as O
    rng = np.random.default_rng(nc + extra + a)
    pred = _random_pred(rng, 3, nc, extra, a, frac)
    got = ops.nms(pred, nc, conf=0.25, iou=0.7, classes=classes, max_det=max_det)
    want, idxs = O.non_max_suppression(torch.from_numpy(pred), 0.25, 0.7, classes=classes, max_det=max_det, nc=nc,
                                       return_idxs=True)
    for (rows, anchors), wr, wi in zip(got, want, idxs):
        assert len(rows) == len(wr)
        np.testing.assert_array_equal(anchors, wi.numpy())           # identical box indices, identical order
        np.testing.assert_array_equal(rows, wr.numpy())              # same fp32 op order -> bit-exact rows


@pytest.mark.parametrize("n,frac", [(2, 0.03), (2, 0.3), (1, 0.9)])
def test_nms_on_a_1280_map_uses_the_multi_launch_sort(n, frac):
    """33,600 anchors (a 1280x1280 input): the candidate sort runs as a sequence of wide launches (collect, chunk-local stages,
    global + LDS merge steps) instead of one block per image; 1k / 10k / 30k candidates = one partial chunk / 4 chunks / 8 chunks"""
    from cvsd_amd import ops
    from oracle import yolo_oracle as O
    rng = np.random.default_rng(int(frac * 100))
    pred = _random_pred(rng, n, 80, 0, 33600, frac)
    got = ops.nms(pred, 80, conf=0.25, iou=0.7, max_det=300)
    want, idxs = O.non_max_suppression(torch.from_numpy(pred), 0.25, 0.7, max_det=300, nc=80, return_idxs=True)
    for (rows, anchors), wr, wi in zip(got, want, idxs):
        np.testing.assert_array_equal(anchors, wi.numpy())
        np.testing.assert_array_equal(rows, wr.numpy())


def test_nms_score_ties_are_stable():
    """equal scores keep ascending anchor order (torch's stable sort)"""
    from cvsd_amd import ops
    from oracle import yolo_oracle as O
    a = 512
    pred = np.zeros((1, 5, a), np.float32)
    pred[0, 0] = (np.arange(a) % 32) * 20 + 10
    pred[0, 1] = (np.arange(a) // 32) * 40 + 10
    pred[0, 2:4] = 30
    pred[0, 4] = 0.5
    (rows, anchors), = ops.nms(pred, 1)
    want, idx = O.non_max_suppression(torch.from_numpy(pred), nc=1, return_idxs=True)
    np.testing.assert_array_equal(anchors, idx[0].numpy())


GROUP_CASES = [
    # n, h, w, cin | conv a: cout, k, stride, fused 1x1 cout (0 = none) | conv b: cout, k, stride
    (1, 40, 40, 64, 64, 3, 1, 0, 80, 3, 1),       # two head-branch 3x3 convs (LDS-staged and split-K instances)
    (4, 20, 20, 64, 64, 3, 1, 64, 80, 3, 1),      # Conv3x3 -> Conv1x1 fused member beside a plain 3x3 member
    (2, 40, 40, 32, 64, 3, 2, 0, 51, 3, 1),       # stride-2 conv beside a ragged-cout (51) conv
    (1, 80, 80, 16, 16, 3, 1, 0, 32, 3, 2),       # short K
]


@pytest.mark.parametrize("case", GROUP_CASES)
def test_grouped_launch_gives_each_member_the_bits_of_its_own_launch(case):
    """conv_f32_group.hip: two independent convs as ONE grid, every pair of menu plans (sampled), repeated: each member's output
    equals its stand-alone launch bit for bit and does not vary between launches"""
    from cvsd_amd import ops
    n, h, w, cin, ca, ka, sa, c2, cb, kb, sb = case
    rng = np.random.default_rng(7)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wa = (rng.standard_normal((ca, cin, ka, ka)) / np.sqrt(cin * ka * ka)).astype(np.float32)
    wb = (rng.standard_normal((cb, cin, kb, kb)) / np.sqrt(cin * kb * kb)).astype(np.float32)
    ba, bb = rng.standard_normal(ca).astype(np.float32) * 0.1, rng.standard_normal(cb).astype(np.float32) * 0.1
    w2 = b2 = None
    if c2:
        w2 = (rng.standard_normal((c2, ca, 1, 1)) / np.sqrt(ca)).astype(np.float32)
        b2 = rng.standard_normal(c2).astype(np.float32) * 0.1
        ra = ops.conv2d_fused(x, wa, ba, w2, b2, stride=sa)
    else:
        ra = ops.conv2d(x, wa, ba, stride=sa)
    rb = ops.conv2d(x, wb, bb, stride=sb)
    _, _, na, nb = ops.conv2d_group(x, wa, ba, wb, bb, sa, sb, w2a=w2, b2a=b2)
    assert na > 0 and nb > 0
    for rep in range(2):
        for pa in range(0, na, max(1, na // 8)):
            for pb in range(0, nb, max(1, nb // 8)):
                ya, yb, _, _ = ops.conv2d_group(x, wa, ba, wb, bb, sa, sb, pa, pb, w2a=w2, b2a=b2)
                assert np.array_equal(ya, ra) and np.array_equal(yb, rb), (case, pa, pb, rep)


C2F_TAIL_CASES = [
    # n, h, w, cin, c1 (3x3 cout), lead channels, c2 (1x1 cout), residual
    (2, 24, 20, 16, 16, 32, 32, True),        # model.2 of YOLOv8n: cat(ys) = 48 channels, the last 16 stay in LDS
    (1, 20, 20, 64, 64, 128, 128, True),      # model.6-like
    (2, 17, 13, 32, 32, 64, 64, False),       # neck C2f (no shortcut), odd image size
    (1, 9, 11, 48, 48, 96, 51, True),         # YOLOv8m widths, ragged pointwise cout
    (3, 8, 8, 16, 16, 48, 32, True),          # two Bottlenecks in front (lead = 3 slices)
]


@pytest.mark.parametrize("case", C2F_TAIL_CASES)
def test_c2f_tail_fused_launch_equals_the_two_convs(case):
    """Bottleneck.cv2 (3x3 + residual) -> C2f.cv2 (1x1 over cat(ys)) as one launch: the pointwise stage reads the earlier concat
    slices from global memory and the last one from the block's LDS image -- bit for bit the two separate launches, every plan"""
    from cvsd_amd import ops
    n, h, w, cin, c1, L, c2, res = case
    rng = np.random.default_rng(11)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    lead = rng.standard_normal((n, h, w, L)).astype(np.float32)
    w1 = (rng.standard_normal((c1, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    w2 = (rng.standard_normal((c2, L + c1, 1, 1)) / np.sqrt(L + c1)).astype(np.float32)
    b1, b2 = rng.standard_normal(c1).astype(np.float32) * 0.1, rng.standard_normal(c2).astype(np.float32) * 0.1
    r = rng.standard_normal((n, h, w, c1)).astype(np.float32) if res else None
    y1 = ops.conv2d(x, w1, b1, residual=r)
    want = ops.conv2d(np.concatenate([lead, y1], axis=-1), w2, b2)
    got, npl = ops.c2f_tail(x, w1, b1, lead, w2, b2, residual=r, return_n_plans=True)
    assert npl >= 1
    np.testing.assert_array_equal(got, want)
    for k in range(1, npl):
        np.testing.assert_array_equal(ops.c2f_tail(x, w1, b1, lead, w2, b2, residual=r, plan=k), want)
