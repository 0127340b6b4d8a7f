"""Checkpoint -> fused weights -> ``.mi355w`` file.

What Ultralytics does at ``YOLO(path)`` + first predict (``/root/reference/model.py:18``):
load a state dict, then ``fuse()`` folds every BatchNorm into its conv
(``utils/torch_utils.py:fuse_conv_and_bn``, BN eps = 1e-3).  Here the fold is
done once, on the host, in the same fp32 operation order, and the result is
written with the op program (graph.py) into one flat file that the C++ engine
maps read-only.

File layout (little endian), version 1::

    char  magic[8] = "MI355YW1"
    u32   version, header_bytes
    u32   family, scale, task, nc, nkpt, kdim, reg_max
    u32   n_buffers, n_ops, n_convs, n_levels
    u32   json_off, json_bytes
    u64   data_bytes
    buffers[n_buffers] : u32 channels, u32 stride_div
    ops[n_ops]         : i32[16]  type,k,s,act, src(buf,choff,c), dst(buf,choff,c), res(buf,choff), conv, pad, 0,0
    convs[n_convs]     : char name[64]; u32 cin,cout,k,s,pad,act; u64 w_off, b_off   (relative to data region)
    levels[n_levels]   : u32 buf, box_off, cls_off, kpt_off, stride
    json               : utf-8 metadata (names, description) -- for the Python side only
    <pad to 256>       : data region: fp32 OIHW fused weights + fp32 biases, each 256-B aligned
"""
from __future__ import annotations

import json
import struct
from typing import Dict, Tuple

import numpy as np

from .graph import (ACT_NONE, ConvSpec, HeadLevel, Op, Program, View, build_program, merge_sibling_convs)

MAGIC = b"MI355YW1"
VERSION = 1
BN_EPS = 1e-3  # ultralytics Conv: nn.BatchNorm2d(c2, eps=0.001, momentum=0.03) after initialize_weights()
_FAMILY_ID = {"v8": 0, "v5u": 1}
_FAMILY_NAME = {v: k for k, v in _FAMILY_ID.items()}


def fuse_conv_bn(w: np.ndarray, gamma: np.ndarray, beta: np.ndarray, mean: np.ndarray, var: np.ndarray,
                 eps: float = BN_EPS) -> Tuple[np.ndarray, np.ndarray]:
    """fuse_conv_and_bn in fp32, same operation order as the torch code:
    ``w' = diag(gamma / sqrt(eps + var)) @ w`` and ``b' = beta - gamma * mean / sqrt(var + eps)`` (conv has no bias)."""
    f32 = np.float32
    w, gamma, beta, mean, var = (np.asarray(a, dtype=f32) for a in (w, gamma, beta, mean, var))
    scale = gamma / np.sqrt(f32(eps) + var)                                  # bn.weight.div(torch.sqrt(bn.eps + bn.running_var))
    wf = (w.reshape(w.shape[0], -1) * scale[:, None]).reshape(w.shape)      # mm with a diagonal matrix: one product per element
    bf = beta - (gamma * mean) / np.sqrt(var + f32(eps))                     # bn.bias - bn.weight.mul(mean).div(sqrt(var + eps))
    return wf.astype(f32), bf.astype(f32)


def fuse_state_dict(prog: Program, sd: Dict[str, np.ndarray]) -> Dict[str, Tuple[np.ndarray, np.ndarray]]:
    """Unfused Ultralytics-named state dict -> {conv name: (w OIHW fp32, b fp32)}."""
    out = {}
    for c in prog.convs:
        if c.has_bn:
            w = np.asarray(sd[f"{c.name}.conv.weight"], dtype=np.float32)    # checkpoints are stored fp16 -> fp32
            wf, bf = fuse_conv_bn(w, sd[f"{c.name}.bn.weight"], sd[f"{c.name}.bn.bias"],
                                  sd[f"{c.name}.bn.running_mean"], sd[f"{c.name}.bn.running_var"])
        else:
            wf = np.asarray(sd[f"{c.name}.weight"], dtype=np.float32)
            bf = np.asarray(sd[f"{c.name}.bias"], dtype=np.float32)
        if wf.shape != (c.cout, c.cin, c.k, c.k) or bf.shape != (c.cout,):
            raise ValueError(f"{c.name}: checkpoint shape {wf.shape} does not match graph ({c.cout},{c.cin},{c.k},{c.k})")
        out[c.name] = (np.ascontiguousarray(wf), np.ascontiguousarray(bf))
    return out


def _v(view) -> Tuple[int, int, int]:
    return (-1, 0, 0) if view is None else (view.buf, view.choff, view.c)


def to_bytes(prog: Program, fused: Dict[str, Tuple[np.ndarray, np.ndarray]], meta: dict | None = None) -> bytes:
    meta = dict(meta or {})
    meta.setdefault("family", prog.family)
    meta.setdefault("scale", prog.scale)
    meta.setdefault("task", "pose" if prog.task else "detect")
    jb = json.dumps(meta).encode()
    # data region layout
    offs, cur = [], 0
    for c in prog.convs:
        w, b = fused[c.name]
        w_off = cur
        cur = (cur + w.nbytes + 255) // 256 * 256
        b_off = cur
        cur = (cur + b.nbytes + 255) // 256 * 256
        offs.append((w_off, b_off))
    data_bytes = cur
    fixed = 8 + 4 * 2 + 4 * 7 + 4 * 4 + 4 * 2 + 8
    tables = 8 * len(prog.buffers) + 64 * len(prog.ops) + (64 + 24 + 16) * len(prog.convs) + 20 * len(prog.levels)
    json_off = fixed + tables
    header_bytes = (json_off + len(jb) + 255) // 256 * 256
    out = bytearray(header_bytes + data_bytes)
    p = 0

    def put(fmt, *vals):
        nonlocal p
        struct.pack_into("<" + fmt, out, p, *vals)
        p += struct.calcsize("<" + fmt)

    put("8s", MAGIC)
    put("II", VERSION, header_bytes)
    put("7I", _FAMILY_ID[prog.family], ord(prog.scale), prog.task, prog.nc, prog.nkpt, prog.kdim, 16)
    put("4I", len(prog.buffers), len(prog.ops), len(prog.convs), len(prog.levels))
    put("II", json_off, len(jb))
    put("Q", data_bytes)
    assert p == fixed
    for ch, sd in prog.buffers:
        put("II", ch, sd)
    for op in prog.ops:
        pad = prog.convs[op.conv].pad if op.conv >= 0 else 0
        put("16i", op.type, op.k, op.s, op.act, *_v(op.src), *_v(op.dst), *_v(op.res)[:2], op.conv, pad, 0, 0)
    for c, (w_off, b_off) in zip(prog.convs, offs):
        put("64s", c.name.encode())
        put("6I", c.cin, c.cout, c.k, c.s, c.pad, c.act)
        put("QQ", w_off, b_off)
    for lv in prog.levels:
        put("5I", lv.buf, lv.box_off, lv.cls_off, lv.kpt_off, lv.stride)
    assert p == json_off
    out[p:p + len(jb)] = jb
    for c, (w_off, b_off) in zip(prog.convs, offs):
        w, b = fused[c.name]
        out[header_bytes + w_off: header_bytes + w_off + w.nbytes] = w.astype("<f4", copy=False).tobytes()
        out[header_bytes + b_off: header_bytes + b_off + b.nbytes] = b.astype("<f4", copy=False).tobytes()
    return bytes(out)


def write_mi355w(path: str, prog: Program, fused, meta: dict | None = None) -> None:
    with open(path, "wb") as f:
        f.write(to_bytes(prog, fused, meta))


def from_bytes(blob: bytes):
    """-> (Program, fused dict, meta).  Inverse of :func:`to_bytes` (used by the Python side and the tests;
    the engine has its own C++ reader of the same layout)."""
    if blob[:8] != MAGIC:
        raise ValueError("not a .mi355w file (bad magic)")
    p = 8

    def get(fmt):
        nonlocal p
        vals = struct.unpack_from("<" + fmt, blob, p)
        p += struct.calcsize("<" + fmt)
        return vals

    version, header_bytes = get("II")
    if version != VERSION:
        raise ValueError(f"unsupported .mi355w version {version}")
    fam, scale, task, nc, nkpt, kdim, _reg = get("7I")
    nb, nops, nconv, nlev = get("4I")
    json_off, json_bytes = get("II")
    (data_bytes,) = get("Q")
    prog = Program(_FAMILY_NAME[fam], chr(scale), task, nc, nkpt, kdim)
    for _ in range(nb):
        prog.buffers.append(get("II"))
    for _ in range(nops):
        t = get("16i")

        def mk(b, o, c):
            return None if b < 0 else View(b, o, c)
        prog.ops.append(Op(t[0], t[1], t[2], t[3], mk(*t[4:7]), mk(*t[7:10]), mk(t[10], t[11], t[9]), t[12]))
    offs = []
    for _ in range(nconv):
        (nm,) = get("64s")
        cin, cout, k, s, pad, act = get("6I")
        w_off, b_off = get("QQ")
        prog.convs.append(ConvSpec(nm.rstrip(b"\0").decode(), cin, cout, k, s, act, act != ACT_NONE, 0))
        offs.append((w_off, b_off))
    for _ in range(nlev):
        prog.levels.append(HeadLevel(*get("5I")))
    meta = json.loads(blob[json_off:json_off + json_bytes].decode())
    # stride_div / has_bn are not stored per conv; recover stride_div from the op's destination buffer
    for op in prog.ops:
        if op.conv >= 0:
            prog.convs[op.conv].stride_div = prog.buffers[op.dst.buf][1]
    fused = {}
    for c, (w_off, b_off) in zip(prog.convs, offs):
        w = np.frombuffer(blob, "<f4", c.cout * c.cin * c.k * c.k, header_bytes + w_off).reshape(c.cout, c.cin, c.k, c.k)
        b = np.frombuffer(blob, "<f4", c.cout, header_bytes + b_off)
        fused[c.name] = (w, b)
    return prog, fused, meta


def read_mi355w(path: str):
    with open(path, "rb") as f:
        return from_bytes(f.read())


def build_from_state_dict(name: str, sd: Dict[str, np.ndarray], nc: int | None = None, meta: dict | None = None) -> bytes:
    """Model name ('yolov8n-pose', 'yolov5mu', ...) + unfused state dict -> .mi355w bytes."""
    from .graph import parse_model_name
    family, scale, task = parse_model_name(name)
    prog = build_program(family, scale, task, nc=nc)
    m = {"model": name}
    m.update(meta or {})
    return to_bytes(*merge_sibling_convs(prog, fuse_state_dict(prog, sd)), m)
