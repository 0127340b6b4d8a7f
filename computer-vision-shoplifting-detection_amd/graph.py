"""Model graph -> fused-op program for the MI355X engine.

This is the host-side restatement of what Ultralytics does when it builds a
model from its embedded yaml (``nn/tasks.py:parse_model``) and walks it at
inference time (``nn/tasks.py:BaseModel._predict_once``), i.e. SURVEY.md
Appendix A.2-A.4.  The reference reaches that code through
``/root/reference/model.py:18`` (``YOLO(path)``) and ``model.py:38``
(``model.track(frame, ...)``).

Instead of a module tree we emit a flat *program*: a list of buffers (NHWC
fp32 activations, described relative to the network input size) and a list of
fused ops (stem / conv+bias+SiLU(+residual) / upsample / SPPF pools) that read
and write *channel slices* of those buffers, so that ``chunk``/``cat`` never
copy anything (C2f, SPPF and the neck concats become channel-offset writes).
The program is serialised into the ``.mi355w`` weight file (weights.py) and
executed by ``csrc/engine.hip``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

# scale -> (depth_multiple, width_multiple, max_channels)   (yolov8.yaml `scales`)
SCALES: Dict[str, Tuple[float, float, int]] = {
    "n": (0.33, 0.25, 1024),
    "s": (0.33, 0.50, 1024),
    "m": (0.67, 0.75, 768),
    "l": (1.00, 1.00, 512),
    "x": (1.00, 1.25, 512),
}

# yolov5.yaml `scales` (the "u" checkpoints are built from it)
SCALES_V5: Dict[str, Tuple[float, float, int]] = {
    "n": (0.33, 0.25, 1024),
    "s": (0.33, 0.50, 1024),
    "m": (0.67, 0.75, 1024),
    "l": (1.00, 1.00, 1024),
    "x": (1.33, 1.25, 1024),
}

# (from, repeats, module, args) -- yolov8.yaml / yolov8-pose.yaml (Appendix A.2)
_V8_BACKBONE = [
    (-1, 1, "Conv", (64, 3, 2)),
    (-1, 1, "Conv", (128, 3, 2)),
    (-1, 3, "C2f", (128, True)),
    (-1, 1, "Conv", (256, 3, 2)),
    (-1, 6, "C2f", (256, True)),
    (-1, 1, "Conv", (512, 3, 2)),
    (-1, 6, "C2f", (512, True)),
    (-1, 1, "Conv", (1024, 3, 2)),
    (-1, 3, "C2f", (1024, True)),
    (-1, 1, "SPPF", (1024, 5)),
]
_V8_HEAD = [
    (-1, 1, "Upsample", ()),
    ((-1, 6), 1, "Concat", ()),
    (-1, 3, "C2f", (512, False)),
    (-1, 1, "Upsample", ()),
    ((-1, 4), 1, "Concat", ()),
    (-1, 3, "C2f", (256, False)),
    (-1, 1, "Conv", (256, 3, 2)),
    ((-1, 12), 1, "Concat", ()),
    (-1, 3, "C2f", (512, False)),
    (-1, 1, "Conv", (512, 3, 2)),
    ((-1, 9), 1, "Concat", ()),
    (-1, 3, "C2f", (1024, False)),
    ((15, 18, 21), 1, "HEAD", ()),
]

# yolov5u (yolov5.yaml, anchor-free "u" head): C3 blocks, 6x6 stem.  The
# reference's literal checkpoint is yolov5mu.pt (/root/reference/model.py:18).
_V5_BACKBONE = [
    (-1, 1, "Conv", (64, 6, 2)),
    (-1, 1, "Conv", (128, 3, 2)),
    (-1, 3, "C3", (128, True)),
    (-1, 1, "Conv", (256, 3, 2)),
    (-1, 6, "C3", (256, True)),
    (-1, 1, "Conv", (512, 3, 2)),
    (-1, 9, "C3", (512, True)),
    (-1, 1, "Conv", (1024, 3, 2)),
    (-1, 3, "C3", (1024, True)),
    (-1, 1, "SPPF", (1024, 5)),
]
_V5_HEAD = [
    (-1, 1, "Conv", (512, 1, 1)),
    (-1, 1, "Upsample", ()),
    ((-1, 6), 1, "Concat", ()),
    (-1, 3, "C3", (512, False)),
    (-1, 1, "Conv", (256, 1, 1)),
    (-1, 1, "Upsample", ()),
    ((-1, 4), 1, "Concat", ()),
    (-1, 3, "C3", (256, False)),
    (-1, 1, "Conv", (256, 3, 2)),
    ((-1, 14), 1, "Concat", ()),
    (-1, 3, "C3", (512, False)),
    (-1, 1, "Conv", (512, 3, 2)),
    ((-1, 10), 1, "Concat", ()),
    (-1, 3, "C3", (1024, False)),
    ((17, 20, 23), 1, "HEAD", ()),
]

FAMILIES = {"v8": (_V8_BACKBONE, _V8_HEAD), "v5u": (_V5_BACKBONE, _V5_HEAD)}

OP_STEM, OP_CONV, OP_UPSAMPLE, OP_SPPF_POOL = 0, 1, 2, 3
ACT_NONE, ACT_SILU = 0, 1
TASK_DETECT, TASK_POSE = 0, 1
REG_MAX = 16


def make_divisible(x: float, divisor: int = 8) -> int:
    return int(math.ceil(x / divisor) * divisor)


@dataclass
class ConvSpec:
    """One fused Conv(+BN)+act, named as in the Ultralytics state dict."""
    name: str            # e.g. "model.2.m.0.cv1" (BN-fused Conv) or "model.22.cv2.0.2" (plain conv2d)
    cin: int
    cout: int
    k: int
    s: int
    act: int             # ACT_SILU for Conv modules, ACT_NONE for the head's final nn.Conv2d
    has_bn: bool         # True: checkpoint holds <name>.conv.weight + <name>.bn.*; False: <name>.weight/.bias
    stride_div: int      # output resolution = input / stride_div

    @property
    def pad(self) -> int:
        """autopad(k) = k // 2, except the yolov5 6x6 stem which is declared with p=2."""
        return 2 if self.k == 6 else self.k // 2


@dataclass
class View:
    buf: int
    choff: int
    c: int


@dataclass
class Op:
    type: int
    k: int = 0
    s: int = 1
    act: int = 0
    src: Optional[View] = None
    dst: Optional[View] = None
    res: Optional[View] = None
    conv: int = -1       # index into Program.convs


@dataclass
class HeadLevel:
    buf: int             # buffer holding the raw head maps of this level
    box_off: int
    cls_off: int
    kpt_off: int
    stride: int


@dataclass
class Program:
    family: str
    scale: str
    task: int
    nc: int
    nkpt: int
    kdim: int
    buffers: List[Tuple[int, int]] = field(default_factory=list)   # (channels, stride_div)
    ops: List[Op] = field(default_factory=list)
    convs: List[ConvSpec] = field(default_factory=list)
    levels: List[HeadLevel] = field(default_factory=list)

    @property
    def nk(self) -> int:
        return self.nkpt * self.kdim

    @property
    def no(self) -> int:
        """Channels of the decoded prediction tensor: 4 + nc (+ nk)."""
        return 4 + self.nc + self.nk

    def num_anchors(self, h: int, w: int) -> int:
        return sum((h // lv.stride) * (w // lv.stride) for lv in self.levels)

    def param_count(self, include_dfl: bool = True) -> int:
        """Fused parameter count (weights + biases), the number Ultralytics prints in
        ``model.info()`` after ``fuse()``; the DFL arange conv (16 frozen params) is counted there."""
        n = sum(c.cout * c.cin * c.k * c.k + c.cout for c in self.convs)
        return n + (REG_MAX if include_dfl else 0)

    def macs(self, h: int = 640, w: int = 640) -> int:
        return sum(c.cout * c.cin * c.k * c.k * (h // c.stride_div) * (w // c.stride_div) for c in self.convs)

    def act_bytes(self, h: int = 640, w: int = 640, elem: int = 4) -> int:
        """SURVEY 8(d) 'layerwise bytes': sum over convs of input read + output write."""
        tot = 0
        for c in self.convs:
            ho, wo = h // c.stride_div, w // c.stride_div
            hi, wi = ho * c.s, wo * c.s
            tot += (c.cin * hi * wi + c.cout * ho * wo) * elem
        return tot


class _Builder:
    def __init__(self, prog: Program):
        self.p = prog

    def new_buf(self, c: int, sd: int) -> int:
        self.p.buffers.append((c, sd))
        return len(self.p.buffers) - 1

    def conv(self, name: str, src: Optional[View], dst: View, cin: int, cout: int, k: int, s: int, sd_out: int,
             act: int = ACT_SILU, has_bn: bool = True, res: Optional[View] = None, stem: bool = False) -> View:
        self.p.convs.append(ConvSpec(name, cin, cout, k, s, act, has_bn, sd_out))
        self.p.ops.append(Op(OP_STEM if stem else OP_CONV, k, s, act, src, dst, res, len(self.p.convs) - 1))
        return dst


def build_program(family: str = "v8", scale: str = "n", task: str = "detect", nc: Optional[int] = None,
                  kpt_shape: Tuple[int, int] = (17, 3)) -> Program:
    """parse_model + the module forward()s, flattened (Appendix A.2-A.4)."""
    depth, width, max_ch = (SCALES_V5 if family == "v5u" else SCALES)[scale]
    tsk = TASK_POSE if task == "pose" else TASK_DETECT
    if nc is None:
        nc = 1 if tsk == TASK_POSE else 80
    nkpt, kdim = (kpt_shape if tsk == TASK_POSE else (0, 0))
    prog = Program(family, scale, tsk, nc, nkpt, kdim)
    b = _Builder(prog)
    backbone, head = FAMILIES[family]
    nodes = list(backbone) + list(head)

    def ch(c: int) -> int:
        return make_divisible(min(c, max_ch) * width, 8)

    def rep(n: int) -> int:
        return max(round(n * depth), 1) if n > 1 else n

    # ---- pass 1: channel count / resolution of every node, and concat placement -------------
    n_nodes = len(nodes)
    c_out = [0] * n_nodes
    sd_out = [1] * n_nodes
    for i, (frm, n, mod, args) in enumerate(nodes):
        f = [frm] if isinstance(frm, int) else list(frm)
        f = [(i + j if j < 0 else j) for j in f]
        cin = 3 if i == 0 else c_out[f[0]]
        sdin = 1 if i == 0 else sd_out[f[0]]
        if mod == "Conv":
            c_out[i], sd_out[i] = ch(args[0]), sdin * args[2]
        elif mod in ("C2f", "C3", "SPPF"):
            c_out[i], sd_out[i] = ch(args[0]), sdin
        elif mod == "Upsample":
            c_out[i], sd_out[i] = cin, sdin // 2
        elif mod == "Concat":
            c_out[i], sd_out[i] = sum(c_out[j] for j in f), sd_out[f[0]]
            assert all(sd_out[j] == sd_out[f[0]] for j in f)
        elif mod == "HEAD":
            pass
    placement: Dict[int, View] = {}
    node_buf: Dict[int, int] = {}
    for i, (frm, n, mod, args) in enumerate(nodes):
        if mod == "Concat":
            f = [(i + j if j < 0 else j) for j in frm]
            node_buf[i] = b.new_buf(c_out[i], sd_out[i])
            off = 0
            for j in f:
                assert j not in placement, "a tensor feeding two concats would need a copy op"
                placement[j] = View(node_buf[i], off, c_out[j])
                off += c_out[j]

    def out_view(i: int) -> View:
        if i in placement:
            return placement[i]
        if i not in node_buf:
            node_buf[i] = b.new_buf(c_out[i], sd_out[i])
        return View(node_buf[i], 0, c_out[i])

    # ---- pass 2: emit ops ---------------------------------------------------------------------
    views: List[Optional[View]] = [None] * n_nodes
    for i, (frm, n, mod, args) in enumerate(nodes):
        f = [frm] if isinstance(frm, int) else list(frm)
        f = [(i + j if j < 0 else j) for j in f]
        name = f"model.{i}"
        sd = sd_out[i]
        if mod == "Conv":
            if i == 0:
                views[i] = b.conv(name, None, out_view(i), 3, c_out[i], args[1], args[2], sd, stem=True)
            else:
                x = views[f[0]]
                views[i] = b.conv(name, x, out_view(i), x.c, c_out[i], args[1], args[2], sd)
        elif mod == "C2f":
            x = views[f[0]]
            c2, nrep, shortcut = c_out[i], rep(n), args[1]
            c = int(c2 * 0.5)
            ybuf = b.new_buf((2 + nrep) * c, sd)                    # cat(ys, 1): never materialised by copies
            b.conv(f"{name}.cv1", x, View(ybuf, 0, 2 * c), x.c, 2 * c, 1, 1, sd)
            for r in range(nrep):
                last = View(ybuf, (1 + r) * c, c)                     # ys[-1]
                tmp = View(b.new_buf(c, sd), 0, c)
                b.conv(f"{name}.m.{r}.cv1", last, tmp, c, c, 3, 1, sd)
                b.conv(f"{name}.m.{r}.cv2", tmp, View(ybuf, (2 + r) * c, c), c, c, 3, 1, sd,
                       res=last if shortcut else None)
            views[i] = b.conv(f"{name}.cv2", View(ybuf, 0, (2 + nrep) * c), out_view(i), (2 + nrep) * c, c2, 1, 1, sd)
        elif mod == "C3":
            x = views[f[0]]
            c2, nrep, shortcut = c_out[i], rep(n), args[1]
            c_ = int(c2 * 0.5)
            ybuf = b.new_buf(2 * c_, sd)                             # cat(m(cv1(x)), cv2(x))
            cur = View(b.new_buf(c_, sd), 0, c_)
            b.conv(f"{name}.cv1", x, cur, x.c, c_, 1, 1, sd)
            b.conv(f"{name}.cv2", x, View(ybuf, c_, c_), x.c, c_, 1, 1, sd)
            for r in range(nrep):
                tmp = View(b.new_buf(c_, sd), 0, c_)
                b.conv(f"{name}.m.{r}.cv1", cur, tmp, c_, c_, 1, 1, sd)
                dst = View(ybuf, 0, c_) if r == nrep - 1 else View(b.new_buf(c_, sd), 0, c_)
                b.conv(f"{name}.m.{r}.cv2", tmp, dst, c_, c_, 3, 1, sd, res=cur if shortcut else None)
                cur = dst
            views[i] = b.conv(f"{name}.cv3", View(ybuf, 0, 2 * c_), out_view(i), 2 * c_, c2, 1, 1, sd)
        elif mod == "SPPF":
            x = views[f[0]]
            c2 = c_out[i]
            c_ = x.c // 2
            ybuf = b.new_buf(4 * c_, sd)
            b.conv(f"{name}.cv1", x, View(ybuf, 0, c_), x.c, c_, 1, 1, sd)
            prog.ops.append(Op(OP_SPPF_POOL, args[1], 1, 0, View(ybuf, 0, c_), View(ybuf, c_, 3 * c_)))
            views[i] = b.conv(f"{name}.cv2", View(ybuf, 0, 4 * c_), out_view(i), 4 * c_, c2, 1, 1, sd)
        elif mod == "Upsample":
            x = views[f[0]]
            views[i] = out_view(i)
            prog.ops.append(Op(OP_UPSAMPLE, 0, 2, 0, x, views[i]))
        elif mod == "Concat":
            views[i] = View(node_buf[i], 0, c_out[i])
        elif mod == "HEAD":
            feats = [views[j] for j in f]
            ch0 = feats[0].c
            c2h = max(16, ch0 // 4, REG_MAX * 4)
            c3h = max(ch0, min(nc, 100))
            c4h = max(ch0 // 4, prog.nk) if tsk == TASK_POSE else 0
            cls_off = 4 * REG_MAX
            kpt_off = cls_off + ((nc + 3) // 4) * 4
            tot = kpt_off + prog.nk
            for li, x in enumerate(feats):
                sdl = sd_out[f[li]]
                hb = b.new_buf(tot, sdl)
                branches = [("cv2", c2h, 4 * REG_MAX, 0), ("cv3", c3h, nc, cls_off)]
                if tsk == TASK_POSE:
                    branches.append(("cv4", c4h, prog.nk, kpt_off))
                for bn_, cmid, cfin, off in branches:
                    t1 = View(b.new_buf(cmid, sdl), 0, cmid)
                    t2 = View(b.new_buf(cmid, sdl), 0, cmid)
                    b.conv(f"{name}.{bn_}.{li}.0", x, t1, x.c, cmid, 3, 1, sdl)
                    b.conv(f"{name}.{bn_}.{li}.1", t1, t2, cmid, cmid, 3, 1, sdl)
                    b.conv(f"{name}.{bn_}.{li}.2", t2, View(hb, off, cfin), cmid, cfin, 1, 1, sdl,
                           act=ACT_NONE, has_bn=False)
                prog.levels.append(HeadLevel(hb, 0, cls_off, kpt_off if tsk == TASK_POSE else 0, sdl))
        else:
            raise ValueError(mod)
    return prog


def merge_sibling_convs(prog: Program, fused: Optional[Dict[str, tuple]] = None):
    """Engine-side optimisation pass: convs that read the SAME input view with the same geometry (the first 3x3 of the
    box / class / keypoint branches of a head level: ``Detect.cv2[i][0]``, ``cv3[i][0]``, ``Pose.cv4[i][0]``) become ONE conv
    whose output channels are the concatenation -- the input tile is staged once instead of two or three times and a level
    costs one launch instead of three.  Every output channel is computed exactly as before (a conv's channels are
    independent), so results are bit-identical; the consumers simply read channel slices of the shared output buffer.

    Returns ``(program, fused)``: a new Program (unused buffers dropped, indices remapped) and, if ``fused`` (name ->
    (w, b)) was given, the matching weight dict with the merged entries ``"a+b+c"``.  ``build_program`` itself stays the
    literal module graph: the oracle, the synthetic checkpoints and the converter work on that.
    """
    import copy
    import numpy as np
    prog = copy.deepcopy(prog)
    fused = dict(fused) if fused is not None else None
    n_ops = len(prog.ops)
    head_bufs = {lv.buf for lv in prog.levels}

    def writers(buf):
        return [o for o in prog.ops if o.dst is not None and o.dst.buf == buf]

    groups: Dict[tuple, List[int]] = {}
    for i, o in enumerate(prog.ops):
        if o.type != OP_CONV or o.res is not None or o.dst.buf in head_bufs:
            continue
        c = prog.convs[o.conv]
        whole = o.dst.choff == 0 and o.dst.c == prog.buffers[o.dst.buf][0] and len(writers(o.dst.buf)) == 1
        if not whole:
            continue
        groups.setdefault((o.src.buf, o.src.choff, o.src.c, c.k, c.s, c.act, prog.buffers[o.dst.buf][1]), []).append(i)
    drop_ops, drop_convs, buf_map = set(), set(), {}
    for key, members in groups.items():
        if len(members) < 2:
            continue
        # slices must start on multiples of 8 channels (16 bytes of fp16): at most one member may have a ragged width, last
        members = sorted(members, key=lambda i: (prog.convs[prog.ops[i].conv].cout % 8 != 0, i))
        if sum(prog.convs[prog.ops[i].conv].cout % 8 != 0 for i in members) > 1:
            continue
        specs = [prog.convs[prog.ops[i].conv] for i in members]
        total = sum(c.cout for c in specs)
        name = "+".join(c.name for c in specs)
        if len(name) > 63:
            continue
        first = prog.ops[members[0]]
        nb = len(prog.buffers)
        prog.buffers.append((total, key[6]))
        off = 0
        for i, c in zip(members, specs):
            buf_map[prog.ops[i].dst.buf] = (nb, off)
            off += c.cout
        if fused is not None:
            fused[name] = (np.ascontiguousarray(np.concatenate([fused[c.name][0] for c in specs], 0)),
                           np.ascontiguousarray(np.concatenate([fused[c.name][1] for c in specs], 0)))
            for c in specs:
                del fused[c.name]
        keep = specs[0]
        prog.convs[first.conv] = ConvSpec(name, keep.cin, total, keep.k, keep.s, keep.act, keep.has_bn, keep.stride_div)
        first.dst = View(nb, 0, total)
        for i in members[1:]:
            drop_ops.add(i)
            drop_convs.add(prog.ops[i].conv)
    if not buf_map:
        return prog, fused
    # consumers of the merged-away buffers read slices of the shared one
    for o in prog.ops:
        for attr in ("src", "res"):
            v = getattr(o, attr)
            if v is not None and v.buf in buf_map:
                nb, off = buf_map[v.buf]
                setattr(o, attr, View(nb, off + v.choff, v.c))
    prog.ops = [o for i, o in enumerate(prog.ops) if i not in drop_ops]
    conv_new = {}
    convs2 = []
    for i, c in enumerate(prog.convs):
        if i not in drop_convs:
            conv_new[i] = len(convs2)
            convs2.append(c)
    prog.convs = convs2
    for o in prog.ops:
        if o.conv >= 0:
            o.conv = conv_new[o.conv]
    # drop buffers nothing refers to any more
    used = set(lv.buf for lv in prog.levels)
    for o in prog.ops:
        for v in (o.src, o.dst, o.res):
            if v is not None:
                used.add(v.buf)
    remap, bufs2 = {}, []
    for i, bdesc in enumerate(prog.buffers):
        if i in used:
            remap[i] = len(bufs2)
            bufs2.append(bdesc)
    prog.buffers = bufs2
    for o in prog.ops:                                             # fresh View objects: the builder shares them between ops
        for attr in ("src", "dst", "res"):
            v = getattr(o, attr)
            if v is not None:
                setattr(o, attr, View(remap[v.buf], v.choff, v.c))
    for lv in prog.levels:
        lv.buf = remap[lv.buf]
    assert len(prog.ops) < n_ops
    return prog, fused


def engine_program(family: str = "v8", scale: str = "n", task: str = "detect", nc: Optional[int] = None) -> Program:
    """The program the engine actually runs: ``build_program`` + the optimisation passes (for tools that match launches)."""
    return merge_sibling_convs(build_program(family, scale, task, nc=nc))[0]


def parse_model_name(name: str) -> Tuple[str, str, str]:
    """'yolov8n', 'yolov8s-pose', 'yolov5mu' -> (family, scale, task)."""
    n = name.lower().replace(".pt", "").replace(".mi355w", "").replace(".yaml", "")
    task = "pose" if n.endswith("-pose") else "detect"
    n = n.replace("-pose", "")
    if n.startswith("yolov8") and len(n) == 7 and n[6] in SCALES:
        return "v8", n[6], task
    if n.startswith("yolov5") and n.endswith("u") and len(n) == 8 and n[6] in SCALES:
        return "v5u", n[6], task
    raise ValueError(f"unsupported model name {name!r} (yolov8{{n,s,m,l,x}}[-pose] or yolov5{{n,s,m,l,x}}u)")
