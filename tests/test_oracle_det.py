"""The canonical-order C oracle: pinned by the golden vectors and bounded against the torch oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import det
from oracle import yolo_oracle as O
from tools import synth

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))


def test_det_expf_accuracy_and_golden():
    x = np.linspace(-87, 88, 200001).astype(np.float32)
    e = det.expf(x)
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(e - ref) / ref) < 1.2e-7                     # <= ~1.5 ulp
    assert det.expf(np.array([0.0], np.float32))[0] == 1.0
    assert det.expf(np.array([-200.0], np.float32))[0] == 0.0 and np.isinf(det.expf(np.array([100.0], np.float32))[0])
    np.testing.assert_array_equal(det.expf(G["expf/x"]), G["expf/y"])


def _ulp_err(got32: np.ndarray, exact64: np.ndarray) -> np.ndarray:
    """|got - exact| in units of the fp32 ulp AT the exact value (subnormal range: the fixed 2^-149 spacing)"""
    ulp = np.maximum(np.spacing(np.abs(exact64).astype(np.float32)).astype(np.float64), 2.0 ** -149)
    return np.abs(got32.astype(np.float64) - exact64) / ulp


def test_det_expf_within_2ulp_of_libm_over_the_whole_range():
    """det_expf is shared by the GPU kernels and the C oracle, so a flaw in it would pass every bit-exact test.
    Pin it against libm's exp (float64, then the distance is measured in fp32 ulps): every fp32 argument on a dense grid of
    the normal-output range [-87.3, 88.7] plus 2M random ones -- <= 2 ulp (measured 1.4); the gradual-underflow range down to
    -104 within 1 ulp of the subnormal spacing + the double rounding of the two-step scaling; exact saturation outside."""
    rng = np.random.default_rng(0)
    x = np.concatenate([np.linspace(-87.3, 88.7, 1_000_001), rng.uniform(-87.3, 88.7, 2_000_000),
                        rng.normal(0, 4, 1_000_000).clip(-87.3, 88.7)]).astype(np.float32)
    err = _ulp_err(det.expf(x), np.exp(x.astype(np.float64)))
    assert err.max() <= 2.0, err.max()
    assert np.quantile(err, 0.999) <= 1.0
    xs = np.linspace(-104.0, -87.3, 400_001).astype(np.float32)           # subnormal results
    assert _ulp_err(det.expf(xs), np.exp(xs.astype(np.float64))).max() <= 2.0
    big = np.array([88.72, 89.0, 100.0, 1e30, np.inf], np.float32)
    assert np.isinf(det.expf(big)[1:]).all() and np.isfinite(det.expf(big)[0])
    small = np.array([-104.0, -150.0, -1e30, -np.inf], np.float32)
    assert (det.expf(small)[1:] == 0).all()
    # monotone non-decreasing on the grid (a polynomial/reduction seam would show as a dip)
    g = det.expf(np.linspace(-20, 20, 2_000_001).astype(np.float32))
    assert (np.diff(g.astype(np.float64)) >= 0).all()


def test_det_silu_and_sigmoid_against_float64():
    """silu(v) = v / (1 + exp(-v)), sigmoid(v) = 1 / (1 + exp(-v)) with the canonical exp and IEEE division: within 3 ulp of
    the float64 value wherever the result is a normal float (exp <= 2 ulp, the add and the division half an ulp each)."""
    rng = np.random.default_rng(1)
    v = np.concatenate([np.linspace(-80, 80, 400_001), rng.normal(0, 3, 1_000_000)]).astype(np.float32)
    y = np.empty_like(v)
    det.lib().det_silu_array(np.ascontiguousarray(v).ctypes.data, y.ctypes.data, v.size)
    v64 = v.astype(np.float64)
    assert _ulp_err(y, v64 / (1.0 + np.exp(-v64))).max() <= 3.0
    sg = (np.float32(1.0) / (np.float32(1.0) + det.expf(-v))).astype(np.float32)
    assert _ulp_err(sg, 1.0 / (1.0 + np.exp(-v64))).max() <= 3.0


@pytest.mark.parametrize("tag", ["conv3x3_s1_res", "conv3x3_s2", "conv1x1_51"])
def test_conv_golden_and_vs_torch(tag):
    x, w, b = G[f"{tag}/x"], G[f"{tag}/w"], G[f"{tag}/b"]
    k, s, act = G[f"{tag}/meta"]
    res = G[f"{tag}/res"] if f"{tag}/res" in G else None
    y = det.conv2d(x, w, b, stride=int(s), act=bool(act), residual=res)
    np.testing.assert_array_equal(y, G[f"{tag}/y"])                    # bit-exact on every host
    ref = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w), torch.from_numpy(b), stride=int(s), padding=int(k) // 2)
    if act:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1).numpy()
    if res is not None:
        ref = ref + res
    np.testing.assert_allclose(y, ref, rtol=2e-5, atol=2e-5)           # same function as torch's conv


@pytest.mark.parametrize("name", ["yolov8n", "yolov8n-pose"])
def test_full_net_golden_head(name):
    prog, sd = synth.synthetic_checkpoint(name, seed=0)
    dm = det.DetOracleModel(name, sd)
    frames = synth.synthetic_frames(2, 64, 96, seed=11)
    _, pred = det.predict(dm, list(frames), conf=0.25, imgsz=96)
    np.testing.assert_array_equal(pred.numpy(), G[f"head_64x96/{name}"])
    # and it is the same function as the torch oracle, up to fp32 re-association noise
    want = O.OracleModel(name, sd).forward(O.preprocess(list(frames), 96)).numpy()
    assert np.abs(pred.numpy() - want).max() < 2e-2 and np.abs(pred.numpy() - want).mean() < 1e-4


def test_det_vs_torch_oracle_640_statistics():
    """documents the noise floor any fp32 implementation has against torch's CPU kernels"""
    name = "yolov8n"
    prog, sd = synth.synthetic_checkpoint(name, seed=0)
    frames = synth.synthetic_frames(1, 640, 640, seed=21)
    rd, pd_ = det.predict(det.DetOracleModel(name, sd), list(frames))
    ro, po = O.predict(O.OracleModel(name, sd), list(frames))
    d = np.abs(pd_.numpy() - po.numpy())
    assert d[:, :4].max() < 5e-2 and d[:, :4].mean() < 5e-4 and d[:, 4:].max() < 1e-3
    np.testing.assert_array_equal(rd[0]["boxes"].numpy(), G[f"rows_640/{name}/0/boxes"])
    np.testing.assert_array_equal(rd[0]["anchor_idx"].numpy(), G[f"rows_640/{name}/0/anchors"])


@pytest.mark.parametrize("tag", ["lb_30x40", "lb_48x48", "lb_100x37"])
def test_letterbox_golden(tag):
    np.testing.assert_array_equal(O.letterbox(G[f"{tag}/in"], (64, 64)), G[f"{tag}/out"])


def test_sigmoid_window_used_by_the_decode_kernel():
    """decode_kernel_v2 evaluates the canonical sigmoid only for class logits >= thr(max logit) (post_kernels.hip) and
    must still return the reference's (max score, FIRST argmax).  Property behind it, on the canonical det_expf:
    logits outside the window never reach the maximum's score; then the windowed scan equals the full scan."""
    from oracle import det
    rng = np.random.default_rng(5)

    def sig(v):
        v = np.asarray(v, dtype=np.float32)
        return (np.float32(1.0) / (np.float32(1.0) + det.expf(-v))).astype(np.float32)

    def thr(m):
        return np.float32(10.9) if m > 11 else (np.float32(-np.inf) if m < -80 else np.float32(m - np.float32(0.01)))

    # strictness just outside the window, over the whole non-saturated range
    m = rng.uniform(-80, 11, 200000).astype(np.float32)
    gap = (np.float32(0.01) + rng.uniform(0, 0.02, m.size).astype(np.float32))
    lo = (m - gap).astype(np.float32)
    lo = np.minimum(lo, np.nextafter((m - np.float32(0.01)).astype(np.float32), np.float32(-np.inf)))   # strictly below the window
    assert (sig(lo) < sig(m)).all()
    # windowed scan == full scan on logit vectors of every flavour (ties, saturation, underflow)
    for scale, shift in [(1, 0), (3, -4), (0.01, 5), (6, 10), (10, -90), (0.001, 0), (20, 0)]:
        L = (rng.standard_normal((3000, 80)) * scale + shift).astype(np.float32)
        L[::7, 5] = L[::7, 40]                                 # exact ties
        S = sig(L)
        want_best, want_idx = S.max(1), S.argmax(1)            # argmax = first occurrence, like the reference's `>` scan
        for i in range(len(L)):
            t = thr(L[i].max())
            cand = np.nonzero(L[i] >= t)[0]
            s = S[i, cand]
            assert s.max() == want_best[i] and cand[s.argmax()] == want_idx[i]
