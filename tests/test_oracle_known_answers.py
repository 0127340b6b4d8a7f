"""Hand-computable known answers pinning the oracle."""
import numpy as np
import torch

from oracle import yolo_oracle as O


def test_letterbox_geometry():
    # 240x320 (UCF-Crime) -> scale 2 -> 480x640, no padding
    assert O.letterbox_geometry(240, 320) == ((640, 480), 0, 0, 0, 0)
    # 720x1280 -> r = 0.5 -> 360x640; dh = 280 % 32 = 24 -> 12 + 12 -> 384x640
    assert O.letterbox_geometry(720, 1280) == ((640, 360), 12, 12, 0, 0)
    # odd padding: dh = 25 -> 12.5 -> top round(12.4)=12, bottom round(12.6)=13
    new_unpad, top, bottom, left, right = O.letterbox_geometry(615, 640)
    assert (new_unpad, top, bottom) == ((640, 615), 12, 13)
    f = np.zeros((720, 1280, 3), np.uint8)
    out = O.letterbox(f)
    assert out.shape == (384, 640, 3) and (out[:12] == 114).all() and (out[12:372] == 0).all() and (out[372:] == 114).all()


def test_resize_linear_u8_known_values():
    # 2x upscale of a 1x2 image [0, 100]: cv2 INTER_LINEAR gives [0, 25, 75, 100]
    img = np.array([[[0, 0, 0], [100, 100, 100]]], np.uint8)
    out = O.resize_linear_u8(img, 4, 1)
    assert out[0, :, 0].tolist() == [0, 25, 75, 100]
    const = np.full((7, 5, 3), 77, np.uint8)
    assert (O.resize_linear_u8(const, 13, 9) == 77).all()


def test_anchors_and_dfl_one_hot():
    """anchor grid first/last points (0.5,0.5)/(19.5,19.5); DFL of a one-hot logit = its bin index"""
    from tools import synth
    prog, sd = synth.synthetic_checkpoint("yolov8n", seed=0)
    om = O.OracleModel("yolov8n", sd)

    class Fake(O.OracleModel):
        def _seq3(self, f, prefix):
            n, _, h, w = f.shape
            if ".cv2." in prefix:                     # box logits: side s one-hot at bin 3*s+1
                t = torch.full((n, 64, h, w), -50.0)
                for s in range(4):
                    t[:, 16 * s + 3 * s + 1] = 50.0
                return t
            return torch.zeros((n, self.nc, h, w))
    fk = Fake("yolov8n", sd)
    feats = [torch.zeros(1, 1, 80, 80), torch.zeros(1, 1, 40, 40), torch.zeros(1, 1, 20, 20)]
    y = fk.Head(feats, "model.22")
    assert y.shape == (1, 84, 8400)
    # distances l,t,r,b = 1,4,7,10 -> cx = ax + (r-l)/2 = 0.5+3, w = l+r = 8 (x stride)
    np.testing.assert_allclose(y[0, :4, 0].numpy(), [(0.5 + 3) * 8, (0.5 + 3) * 8, 8 * 8, 14 * 8], rtol=1e-6)
    np.testing.assert_allclose(y[0, :4, 8399].numpy(), [(19.5 + 3) * 32, (19.5 + 3) * 32, 8 * 32, 14 * 32], rtol=1e-6)
    assert torch.allclose(y[0, 4:], torch.full((80, 8400), 0.5))


def test_nms_three_boxes():
    # A (0.9) and B (0.8) overlap with IoU 0.8 > 0.7 -> B suppressed; C (0.7) disjoint -> kept; D below conf
    pred = torch.zeros(1, 5, 4)
    boxes_xyxy = torch.tensor([[0, 0, 100, 100], [0, 0, 100, 80], [200, 200, 260, 260], [0, 0, 10, 10]], dtype=torch.float32)
    pred[0, 0] = (boxes_xyxy[:, 0] + boxes_xyxy[:, 2]) / 2
    pred[0, 1] = (boxes_xyxy[:, 1] + boxes_xyxy[:, 3]) / 2
    pred[0, 2] = boxes_xyxy[:, 2] - boxes_xyxy[:, 0]
    pred[0, 3] = boxes_xyxy[:, 3] - boxes_xyxy[:, 1]
    pred[0, 4] = torch.tensor([0.9, 0.8, 0.7, 0.2])
    out, idx = O.non_max_suppression(pred, 0.25, 0.7, nc=1, return_idxs=True)
    assert idx[0].tolist() == [0, 2]
    np.testing.assert_allclose(out[0][:, :4].numpy(), boxes_xyxy[[0, 2]].numpy())
    out, idx = O.non_max_suppression(pred, 0.25, 0.85, nc=1, return_idxs=True)      # IoU 0.8 <= 0.85 -> all three
    assert idx[0].tolist() == [0, 1, 2]
    # different classes never suppress each other (class offset 7680)
    pred2 = torch.zeros(1, 6, 2)
    pred2[0, :4, 0] = torch.tensor([50.0, 50, 100, 100])
    pred2[0, :4, 1] = torch.tensor([50.0, 50, 100, 100])
    pred2[0, 4, 0], pred2[0, 5, 1] = 0.9, 0.8
    out, idx = O.non_max_suppression(pred2, 0.25, 0.7, nc=2, return_idxs=True)
    assert idx[0].tolist() == [0, 1] and out[0][:, 5].tolist() == [0.0, 1.0]
    out = O.non_max_suppression(pred2, 0.25, 0.7, nc=2, classes=[1])
    assert out[0][:, 5].tolist() == [1.0]


def test_scale_back_and_xywhn():
    # 720x1280 frame letterboxed to 384x640: gain 0.5, pad_y 12
    b = torch.tensor([[100.0, 112.0, 300.0, 212.0]])
    out = O.scale_boxes((384, 640), b.clone(), (720, 1280))
    assert out.tolist() == [[200.0, 200.0, 600.0, 400.0]]
    assert O.scale_boxes((384, 640), torch.tensor([[-5.0, 0.0, 700.0, 500.0]]), (720, 1280)).tolist() == [[0.0, 0.0, 1280.0, 720.0]]
    k = O.scale_coords((384, 640), torch.tensor([[[100.0, 112.0, 0.9]]]), (720, 1280))
    assert k.tolist() == [[[200.0, 200.0, 0.8999999761581421]]]
    xywhn = O.boxes_xywhn(torch.tensor([[200.0, 200.0, 600.0, 400.0]]), (720, 1280))
    np.testing.assert_allclose(xywhn.numpy(), [[400 / 1280, 300 / 720, 400 / 1280, 200 / 720]], rtol=1e-7)


def test_half_storage_mode_of_the_oracle(v8n):
    """oracle half=True (the engine's half contract): the weights of every conv (the stem's too: model.half()), the /255 input
    (im.half()) and every stored activation are fp16 values; the head logits stay fp32; results stay close to fp32; fp32 mode is
    untouched by the option."""
    import torch
    from oracle import yolo_oracle as O
    from tools import synth
    sd = v8n[1]
    frames = synth.synthetic_frames(1, 96, 96, seed=2)
    x = O.preprocess(list(frames), 96)
    o32, o16 = O.OracleModel("yolov8n", sd), O.OracleModel("yolov8n", sd, half=True)
    w16, _ = o16._fused_conv("model.2.cv1")
    assert torch.equal(w16, w16.half().float())                       # fp16-representable weights
    w_stem16, _ = o16._fused_conv("model.0")
    w_stem32, _ = o32._fused_conv("model.0")
    assert torch.equal(w_stem16, w_stem32.half().float()) and not torch.equal(w_stem16, w_stem32)   # the stem's weights are fp16 too
    feats = o16.forward(x, return_features=True)
    for f in feats:
        assert torch.equal(f, f.half().float())                       # stored activations are fp16 values
    a, b = o32.forward(x), o16.forward(x)
    assert a.shape == b.shape and not torch.equal(a, b)
    assert (a[:, :4] - b[:, :4]).abs().median() < 0.1                 # fp16 rounding noise only (px)
    assert (a[:, 4:] - b[:, 4:]).abs().max() < 0.05
    # the residual add is rounded once, after the add (Bottleneck with shortcut)
    y = o16.Bottleneck(o16._store(torch.randn(1, 16, 8, 8)), "model.2.m.0", True)
    assert torch.equal(y, y.half().float())
