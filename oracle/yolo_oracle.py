"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file.  It is a plain torch-CPU fp32 restatement of the per-frame YOLO path that
``/root/reference/model.py:18,38`` reaches inside the un-vendored dependency
``ultralytics==8.3.225`` (``/root/reference/requirements.txt:121``): letterbox -> fused
Conv/C2f/C3/SPPF backbone+neck -> Detect/Pose head (DFL, anchors) -> NMS -> scale-back ->
Boxes/Keypoints math.  SURVEY.md Appendix A is the spec it follows; each function names the
Ultralytics module it restates.

PARITY UNPINNED: ultralytics, torchvision and cv2 are absent from this container and the
reference ships no tests, golden vectors or weights for this path, so this
oracle cannot be checked against the real package here.  What pins it instead: exact fused
parameter counts / GFLOPs of the public model cards (tests/test_graph.py), hand-computed
known answers (tests/test_oracle_known_answers.py), and the committed golden vectors.

It deliberately shares no code with ``computer-vision-shoplifting-detection_amd``: it keeps
its own yaml tables, BN fold, anchors, NMS and scale-back.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- yaml (A.2)
V8_SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768),
             "l": (1.00, 1.00, 512), "x": (1.00, 1.25, 512)}
V5_SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 1024),
             "l": (1.00, 1.00, 1024), "x": (1.33, 1.25, 1024)}
V8_YAML = [
    [-1, 1, "Conv", [64, 3, 2]], [-1, 1, "Conv", [128, 3, 2]], [-1, 3, "C2f", [128, True]],
    [-1, 1, "Conv", [256, 3, 2]], [-1, 6, "C2f", [256, True]], [-1, 1, "Conv", [512, 3, 2]],
    [-1, 6, "C2f", [512, True]], [-1, 1, "Conv", [1024, 3, 2]], [-1, 3, "C2f", [1024, True]],
    [-1, 1, "SPPF", [1024, 5]],
    [-1, 1, "Upsample", []], [[-1, 6], 1, "Concat", []], [-1, 3, "C2f", [512]],
    [-1, 1, "Upsample", []], [[-1, 4], 1, "Concat", []], [-1, 3, "C2f", [256]],
    [-1, 1, "Conv", [256, 3, 2]], [[-1, 12], 1, "Concat", []], [-1, 3, "C2f", [512]],
    [-1, 1, "Conv", [512, 3, 2]], [[-1, 9], 1, "Concat", []], [-1, 3, "C2f", [1024]],
    [[15, 18, 21], 1, "Head", []],
]
V5_YAML = [
    [-1, 1, "Conv", [64, 6, 2, 2]], [-1, 1, "Conv", [128, 3, 2]], [-1, 3, "C3", [128]],
    [-1, 1, "Conv", [256, 3, 2]], [-1, 6, "C3", [256]], [-1, 1, "Conv", [512, 3, 2]],
    [-1, 9, "C3", [512]], [-1, 1, "Conv", [1024, 3, 2]], [-1, 3, "C3", [1024]],
    [-1, 1, "SPPF", [1024, 5]],
    [-1, 1, "Conv", [512, 1, 1]], [-1, 1, "Upsample", []], [[-1, 6], 1, "Concat", []], [-1, 3, "C3", [512, False]],
    [-1, 1, "Conv", [256, 1, 1]], [-1, 1, "Upsample", []], [[-1, 4], 1, "Concat", []], [-1, 3, "C3", [256, False]],
    [-1, 1, "Conv", [256, 3, 2]], [[-1, 14], 1, "Concat", []], [-1, 3, "C3", [512, False]],
    [-1, 1, "Conv", [512, 3, 2]], [[-1, 10], 1, "Concat", []], [-1, 3, "C3", [1024, False]],
    [[17, 20, 23], 1, "Head", []],
]


def _model_tables(name: str):
    n = name.lower()
    pose = n.endswith("-pose")
    n = n.replace("-pose", "")
    if n.startswith("yolov8"):
        return V8_YAML, V8_SCALES[n[6]], pose
    if n.startswith("yolov5") and n.endswith("u"):
        return V5_YAML, V5_SCALES[n[6]], pose
    raise ValueError(name)


# ----------------------------------------------------------------------------- modules (A.3)
class OracleModel:
    """Functional YOLO forward over an *unfused* Ultralytics-named state dict."""

    def __init__(self, name: str, state_dict: Dict[str, np.ndarray], nc: Optional[int] = None, half: bool = False):
        # half: restates the ENGINE's half=True contract (include/mi355_yolo.h, mi355_opts.half), which is what Ultralytics'
        # half=True predictor does up to where roundings fall: weights of every conv (the stem's too) and the /255 input rounded to fp16,
        # every stored activation rounded to fp16 once (after bias + SiLU + residual, all in fp32), the head's final 1x1
        # convs, decode and NMS in fp32.  Ultralytics itself refuses half on CPU, so this mode has no CPU reference run.
        self.half = bool(half)
        self.yaml, (self.depth, self.width, self.max_ch), self.pose = _model_tables(name)
        self.nc = nc if nc is not None else (1 if self.pose else 80)
        self.kpt_shape = (17, 3) if self.pose else (0, 0)
        self.nk = self.kpt_shape[0] * self.kpt_shape[1]
        self.reg_max = 16
        self.sd = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)).clone() for k, v in state_dict.items()}
        self._fused: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.stride = [8, 16, 32]
        self.n_params = 0

    # -- utils/torch_utils.py:fuse_conv_and_bn
    def _fused_conv(self, prefix: str):
        if prefix not in self._fused:
            w = self.sd[prefix + ".conv.weight"]
            g, b = self.sd[prefix + ".bn.weight"], self.sd[prefix + ".bn.bias"]
            mu, var = self.sd[prefix + ".bn.running_mean"], self.sd[prefix + ".bn.running_var"]
            eps = 1e-3
            w_bn = torch.diag(g.div(torch.sqrt(eps + var)))
            wf = torch.mm(w_bn, w.view(w.shape[0], -1)).view(w.shape)
            b_conv = torch.zeros(w.shape[0])
            b_bn = b - g.mul(mu).div(torch.sqrt(var + eps))
            bf = torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + b_bn
            if self.half:                                # model.half(): every conv's weights, the stem's included
                wf = wf.half().float()
            self._fused[prefix] = (wf, bf)
        return self._fused[prefix]

    def _store(self, x):
        """what a stored activation holds: fp32, or fp16-rounded values in half mode"""
        return x.half().float() if self.half else x

    # -- nn/modules/conv.py:Conv.forward_fuse  (act(conv(x)), act = SiLU, pad = autopad(k) unless given)
    def Conv(self, x, prefix, k, s, p=None, res=None):
        w, b = self._fused_conv(prefix)
        assert w.shape[2] == k, (prefix, w.shape, k)
        y = F.silu(F.conv2d(x, w, b, stride=s, padding=(k // 2 if p is None else p)))
        if res is not None:
            y = res + y                      # Bottleneck's `x + self.cv2(self.cv1(x))`
        return self._store(y)

    # -- nn/modules/block.py:Bottleneck
    def Bottleneck(self, x, prefix, shortcut, k=(3, 3)):
        return self.Conv(self.Conv(x, prefix + ".cv1", k[0], 1), prefix + ".cv2", k[1], 1, res=x if shortcut else None)

    # -- nn/modules/block.py:C2f.forward
    def C2f(self, x, prefix, n, shortcut):
        y = list(self.Conv(x, prefix + ".cv1", 1, 1).chunk(2, 1))
        for i in range(n):
            y.append(self.Bottleneck(y[-1], f"{prefix}.m.{i}", shortcut))
        return self.Conv(torch.cat(y, 1), prefix + ".cv2", 1, 1)

    # -- nn/modules/block.py:C3.forward
    def C3(self, x, prefix, n, shortcut):
        a = self.Conv(x, prefix + ".cv1", 1, 1)
        for i in range(n):
            a = self.Bottleneck(a, f"{prefix}.m.{i}", shortcut, k=(1, 3))
        return self.Conv(torch.cat((a, self.Conv(x, prefix + ".cv2", 1, 1)), 1), prefix + ".cv3", 1, 1)

    # -- nn/modules/block.py:SPPF.forward
    def SPPF(self, x, prefix, k=5):
        y = [self.Conv(x, prefix + ".cv1", 1, 1)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], kernel_size=k, stride=1, padding=k // 2))
        return self.Conv(torch.cat(y, 1), prefix + ".cv2", 1, 1)

    def _seq3(self, x, prefix):
        """head branch: Conv3 -> Conv3 -> nn.Conv2d 1x1 (with bias, no BN, no act)."""
        x = self.Conv(self.Conv(x, prefix + ".0", 3, 1), prefix + ".1", 3, 1)
        w = self.sd[prefix + ".2.weight"]
        return F.conv2d(x, w.half().float() if self.half else w, self.sd[prefix + ".2.bias"])

    # -- nn/modules/head.py:Detect.forward/_inference, Pose.forward/kpts_decode; utils/tal.py
    def Head(self, feats: List[torch.Tensor], prefix: str):
        bs = feats[0].shape[0]
        no = self.nc + self.reg_max * 4
        x = [torch.cat((self._seq3(f, f"{prefix}.cv2.{i}"), self._seq3(f, f"{prefix}.cv3.{i}")), 1)
             for i, f in enumerate(feats)]
        # make_anchors(x, stride, 0.5)
        anchor_points, stride_tensor = [], []
        for xi, s in zip(x, self.stride):
            h, w = xi.shape[2:]
            sx = torch.arange(end=w, dtype=torch.float32) + 0.5
            sy = torch.arange(end=h, dtype=torch.float32) + 0.5
            sy, sx = torch.meshgrid(sy, sx, indexing="ij")
            anchor_points.append(torch.stack((sx, sy), -1).view(-1, 2))
            stride_tensor.append(torch.full((h * w, 1), float(s), dtype=torch.float32))
        anchors = torch.cat(anchor_points).transpose(0, 1)
        strides = torch.cat(stride_tensor).transpose(0, 1)
        x_cat = torch.cat([xi.view(bs, no, -1) for xi in x], 2)
        box, cls = x_cat.split((self.reg_max * 4, self.nc), 1)
        # DFL: conv(arange(16)) over softmax of the 16 bins
        b, _, a = box.shape
        proj = torch.arange(self.reg_max, dtype=torch.float32).view(1, self.reg_max, 1, 1)
        dist = F.conv2d(box.view(b, 4, self.reg_max, a).transpose(2, 1).softmax(1), proj).view(b, 4, a)
        # dist2bbox(xywh=True)
        lt, rb = dist.chunk(2, 1)
        x1y1 = anchors.unsqueeze(0) - lt
        x2y2 = anchors.unsqueeze(0) + rb
        c_xy = (x1y1 + x2y2) / 2
        wh = x2y2 - x1y1
        dbox = torch.cat((c_xy, wh), 1) * strides
        y = torch.cat((dbox, cls.sigmoid()), 1)
        if not self.pose:
            return y
        kpt = torch.cat([self._seq3(f, f"{prefix}.cv4.{i}").view(bs, self.nk, -1) for i, f in enumerate(feats)], -1)
        ndim = self.kpt_shape[1]
        yk = kpt.clone()
        if ndim == 3:
            yk[:, 2::ndim] = yk[:, 2::ndim].sigmoid()
        yk[:, 0::ndim] = (yk[:, 0::ndim] * 2.0 + (anchors[0] - 0.5)) * strides
        yk[:, 1::ndim] = (yk[:, 1::ndim] * 2.0 + (anchors[1] - 0.5)) * strides
        return torch.cat([y, yk], 1)

    # -- nn/tasks.py:parse_model + BaseModel._predict_once
    def _ch(self, c):
        return int(math.ceil(min(c, self.max_ch) * self.width / 8) * 8)

    @torch.no_grad()
    def forward(self, x: torch.Tensor, return_features: bool = False):
        ys: List[torch.Tensor] = []
        x = self._store(x)                               # half: im.half() -- the /255 input is an fp16 tensor too
        for i, (f, n, m, args) in enumerate(self.yaml):
            n = max(round(n * self.depth), 1) if n > 1 else n
            if isinstance(f, int):
                xin = x if f == -1 and i == 0 else ys[f if f >= 0 else i + f]
            else:
                xin = [ys[j if j >= 0 else i + j] for j in f]
            p = f"model.{i}"
            if m == "Conv":
                out = self.Conv(xin, p, args[1], args[2], args[3] if len(args) > 3 else None)
            elif m == "C2f":
                out = self.C2f(xin, p, n, args[1] if len(args) > 1 else False)
            elif m == "C3":
                out = self.C3(xin, p, n, args[1] if len(args) > 1 else True)
            elif m == "SPPF":
                out = self.SPPF(xin, p, args[1])
            elif m == "Upsample":
                out = F.interpolate(xin, scale_factor=2.0, mode="nearest")
            elif m == "Concat":
                out = torch.cat(xin, 1)
            elif m == "Head":
                if return_features:
                    return xin
                out = self.Head(xin, p)
            ys.append(out)
        return ys[-1]

    def count_params(self) -> int:
        """fused parameter count as Ultralytics' model.info() prints it (DFL's 16 frozen weights included)."""
        tot = 0
        for k, v in self.sd.items():
            if k.endswith("conv.weight") and ".dfl." not in k:
                tot += v.numel() + v.shape[0]         # fused conv gains a bias
            elif k.endswith(".2.weight") or k.endswith(".2.bias") or ".dfl." in k:
                tot += v.numel()
        return tot


# ----------------------------------------------------------------------------- preprocess (A.1)
def resize_linear_u8(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv2.resize(img, (dw, dh), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC, restated from
    OpenCV's fixed-point path (opencv-python==4.12.0.88, /root/reference/requirements.txt:69):
    11-bit coefficients, horizontal pass in int32, vertical pass
    ``((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2``.  Pure integer arithmetic."""
    sh, sw = img.shape[:2]
    if (sw, sh) == (dw, dh):
        return img.copy()
    SCALE = 2048

    def coeffs(dn, sn):
        scale = sn / dn
        idx = np.zeros(dn, np.int64)
        c0 = np.zeros(dn, np.int64)
        c1 = np.zeros(dn, np.int64)
        for d in range(dn):
            fx = np.float32((d + 0.5) * scale - 0.5)
            s = int(math.floor(fx))
            fx = np.float32(fx - s)
            if s < 0:
                s, fx = 0, np.float32(0)
            if s >= sn - 1:
                s, fx = sn - 1, np.float32(0)
            idx[d] = s
            # saturate_cast<short>(cvRound(f * 2048)) -- cvRound = round half to even
            c0[d] = int(np.rint(np.float32(np.float32(1.0) - fx) * np.float32(SCALE)))
            c1[d] = int(np.rint(fx * np.float32(SCALE)))
        return idx, c0, c1

    xi, xa0, xa1 = coeffs(dw, sw)
    yi, yb0, yb1 = coeffs(dh, sh)
    src = img.astype(np.int64)
    xi1 = np.minimum(xi + 1, sw - 1)
    hor = src[:, xi, :] * xa0[None, :, None] + src[:, xi1, :] * xa1[None, :, None]      # [sh, dw, C]
    yi1 = np.minimum(yi + 1, sh - 1)
    s0, s1 = hor[yi], hor[yi1]
    out = (((yb0[:, None, None] * (s0 >> 4)) >> 16) + ((yb1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_geometry(h0: int, w0: int, new_shape=(640, 640), auto=True, stride=32, scaleup=True):
    """data/augment.py:LetterBox.__call__ geometry -> (new_unpad (w,h), top, bottom, left, right)."""
    r = min(new_shape[0] / h0, new_shape[1] / w0)
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(w0 * r)), int(round(h0 * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_unpad, top, bottom, left, right


def letterbox(img: np.ndarray, new_shape=(640, 640), auto=True, stride=32) -> np.ndarray:
    h0, w0 = img.shape[:2]
    new_unpad, top, bottom, left, right = letterbox_geometry(h0, w0, new_shape, auto, stride)
    if (w0, h0) != new_unpad:
        img = resize_linear_u8(img, new_unpad[0], new_unpad[1])
    out = np.full((img.shape[0] + top + bottom, img.shape[1] + left + right, 3), 114, np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out


def preprocess(frames: Sequence[np.ndarray], imgsz: int = 640) -> torch.Tensor:
    """engine/predictor.py:BasePredictor.preprocess: letterbox each -> stack -> BGR->RGB -> NCHW -> float -> /255."""
    im = np.stack([letterbox(f, (imgsz, imgsz)) for f in frames])
    im = im[..., ::-1].transpose((0, 3, 1, 2))
    im = np.ascontiguousarray(im)
    t = torch.from_numpy(im).float()
    t /= 255
    return t


# ----------------------------------------------------------------------------- NMS (A.5)
def xywh2xyxy(x: torch.Tensor) -> torch.Tensor:
    y = torch.empty_like(x)
    xy = x[..., :2]
    wh = x[..., 2:] / 2
    y[..., :2] = xy - wh
    y[..., 2:] = xy + wh
    return y


def nms_greedy(boxes: torch.Tensor, scores: torch.Tensor, iou_thres: float) -> torch.Tensor:
    """torchvision.ops.nms (CPU kernel): stable descending sort, suppress IoU > thr,
    IoU = inter / (area_i + area_j - inter), all in fp32.  The walk over the sorted candidates runs on numpy views of the
    same fp32 values (identical IEEE operations, a fraction of the per-call overhead of 0-d torch tensors: this function
    sits inside bench.py's timed CPU baseline); one vectorised IoU row per KEPT box, suppressed candidates cost one test."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = boxes.detach().to(torch.float32).numpy()
    order = torch.sort(scores, stable=True, descending=True)[1].numpy()
    b = np.ascontiguousarray(b[order])                          # candidates in score order
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    n = b.shape[0]
    suppressed = np.zeros(n, dtype=bool)
    thr = np.float32(iou_thres)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 == n:
            break
        w = np.maximum(np.minimum(x2[i], x2[i + 1:]) - np.maximum(x1[i], x1[i + 1:]), np.float32(0))
        h = np.maximum(np.minimum(y2[i], y2[i + 1:]) - np.maximum(y1[i], y1[i + 1:]), np.float32(0))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > thr
    return torch.from_numpy(order[np.asarray(keep, dtype=np.int64)].astype(np.int64))


def non_max_suppression(prediction: torch.Tensor, conf_thres=0.25, iou_thres=0.7, classes=None, agnostic=False,
                        max_det=300, nc=0, max_nms=30000, max_wh=7680, return_idxs=False):
    """utils/nms.py:non_max_suppression (multi_label=False, no labels, not rotated, not end2end).
    prediction: [B, 4+nc+extra, A] -> list of [n, 6+extra] rows (xyxy, conf, cls, extra), conf-descending.
    With return_idxs the anchor index of every kept row is returned too (bookkeeping only)."""
    bs = prediction.shape[0]
    nc = nc or (prediction.shape[1] - 4)
    extra = prediction.shape[1] - nc - 4
    mi = 4 + nc
    xc = prediction[:, 4:mi].amax(1) > conf_thres
    xinds = torch.arange(prediction.shape[-1]).expand(bs, -1)[..., None]
    prediction = prediction.transpose(-1, -2)
    prediction = torch.cat((xywh2xyxy(prediction[..., :4]), prediction[..., 4:]), dim=-1)
    output = [torch.zeros((0, 6 + extra))] * bs
    keepi = [torch.zeros((0,), dtype=torch.int64)] * bs
    for xi, (x, xk) in enumerate(zip(prediction, xinds)):
        filt = xc[xi]
        x = x[filt]
        xk = xk[filt]
        if not x.shape[0]:
            continue
        box, cls, mask = x.split((4, nc, extra), 1)
        conf, j = cls.max(1, keepdim=True)
        filt = conf.view(-1) > conf_thres
        x = torch.cat((box, conf, j.float(), mask), 1)[filt]
        xk = xk[filt]
        if classes is not None:
            filt = (x[:, 5:6] == torch.tensor(classes, dtype=x.dtype)).any(1)
            x, xk = x[filt], xk[filt]
        n = x.shape[0]
        if not n:
            continue
        if n > max_nms:
            filt = x[:, 4].argsort(descending=True)[:max_nms]
            x, xk = x[filt], xk[filt]
        c = x[:, 5:6] * (0 if agnostic else max_wh)
        scores = x[:, 4]
        boxes = x[:, :4] + c
        i = nms_greedy(boxes, scores, iou_thres)
        i = i[:max_det]
        output[xi], keepi[xi] = x[i], xk[i].view(-1)
    return (output, keepi) if return_idxs else output


# ----------------------------------------------------------------------------- scale-back (A.6)
def scale_boxes(img1_shape, boxes: torch.Tensor, img0_shape) -> torch.Tensor:
    """utils/ops.py:scale_boxes + clip_boxes (padding=True, xywh=False)."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad_x = round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1)
    pad_y = round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1)
    boxes[..., 0] -= pad_x
    boxes[..., 1] -= pad_y
    boxes[..., 2] -= pad_x
    boxes[..., 3] -= pad_y
    boxes[..., :4] /= gain
    h, w = img0_shape[:2]
    boxes[..., 0] = boxes[..., 0].clamp(0, w)
    boxes[..., 1] = boxes[..., 1].clamp(0, h)
    boxes[..., 2] = boxes[..., 2].clamp(0, w)
    boxes[..., 3] = boxes[..., 3].clamp(0, h)
    return boxes


def scale_coords(img1_shape, coords: torch.Tensor, img0_shape) -> torch.Tensor:
    """utils/ops.py:scale_coords + clip_coords (normalize=False, padding=True)."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    coords[..., 0] -= pad[0]
    coords[..., 1] -= pad[1]
    coords[..., 0] /= gain
    coords[..., 1] /= gain
    h, w = img0_shape[:2]
    coords[..., 0] = coords[..., 0].clamp(0, w)
    coords[..., 1] = coords[..., 1].clamp(0, h)
    return coords


# ----------------------------------------------------------------------------- Results math (A.7)
def boxes_xywh(xyxy: torch.Tensor) -> torch.Tensor:
    """utils/ops.py:xyxy2xywh"""
    y = torch.empty_like(xyxy)
    y[..., 0] = (xyxy[..., 0] + xyxy[..., 2]) / 2
    y[..., 1] = (xyxy[..., 1] + xyxy[..., 3]) / 2
    y[..., 2] = xyxy[..., 2] - xyxy[..., 0]
    y[..., 3] = xyxy[..., 3] - xyxy[..., 1]
    return y


def boxes_xywhn(xyxy: torch.Tensor, orig_shape) -> torch.Tensor:
    """engine/results.py:Boxes.xywhn: xywh with x/w divided by width and y/h by height."""
    xywh = boxes_xywh(xyxy)
    xywh[..., [0, 2]] /= orig_shape[1]
    xywh[..., [1, 3]] /= orig_shape[0]
    return xywh


def keypoints_xy(data: torch.Tensor) -> torch.Tensor:
    """engine/results.py:Keypoints.__init__/.xy: points with conf < 0.5 are zeroed."""
    k = data.clone()
    if k.shape[-1] == 3:
        mask = k[..., 2] < 0.5
        k[..., :2][mask] = 0
    return k[..., :2]


# ----------------------------------------------------------------------------- end-to-end predict
@torch.no_grad()
def predict(model: OracleModel, frames: Sequence[np.ndarray], conf=0.25, iou=0.7, classes=None, max_det=300,
            imgsz=640):
    """models/yolo/{detect,pose}/predict.py postprocess: -> list of dicts
    {boxes [n,6] (orig px), kpts [n,17,3] or None, anchor_idx [n]} for BGR uint8 frames."""
    im = preprocess(frames, imgsz)
    pred = model.forward(im)
    rows, idxs = non_max_suppression(pred, conf, iou, classes=classes, max_det=max_det, nc=model.nc, return_idxs=True)
    out = []
    for r, ai, f in zip(rows, idxs, frames):
        r = r.clone()
        r[:, :4] = scale_boxes(im.shape[2:], r[:, :4], f.shape)
        k = None
        if model.pose:
            k = r[:, 6:].view(len(r), *model.kpt_shape).clone()
            k = scale_coords(im.shape[2:], k, f.shape)
        out.append({"boxes": r[:, :6], "kpts": k, "anchor_idx": ai})
    return out, pred
