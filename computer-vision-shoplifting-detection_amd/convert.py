"""Ultralytics ``.pt`` checkpoint -> ``.mi355w`` image, without importing ultralytics.

``YOLO("./models/yolov5mu.pt")`` (``/root/reference/model.py:18``) unpickles a dict whose ``"model"`` entry is an
``ultralytics.nn.tasks.DetectionModel``/``PoseModel`` instance.  The package is not installed here (and must not be
needed at inference time), so the pickle is read with stub classes standing in for everything under ``ultralytics.*``:
only the ``nn.Module`` state layout (``_modules`` / ``_parameters`` / ``_buffers``) and three attributes (``yaml``,
``names``, ``nc``) are used.  Weights stored as fp16 are widened to fp32, BatchNorm is folded by ``weights.py``.

PARITY UNPINNED: no real checkpoint exists in the build image; the reader is exercised on look-alike checkpoints
fabricated by ``tests/test_convert.py``.  Nothing is downloaded: a missing file is an error.
"""
from __future__ import annotations

import io
import pickle
import types
from typing import Dict, Tuple

import numpy as np
import torch

from .weights import build_from_state_dict


class _Stub:
    """Placeholder for any class under ``ultralytics.*`` found in the pickle."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[0], dict):
            self.__dict__.update(state[0])
            if isinstance(state[1], dict):
                self.__dict__.update(state[1])

    def __call__(self, *a, **k):
        return None


def _stub_for(module: str, name: str):
    return type(name, (_Stub,), {"__module__": module})


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] in ("ultralytics", "models", "utils", "thop", "dill"):
            return _stub_for(module, name)
        return super().find_class(module, name)


_pickle_module = types.ModuleType("cvsd_amd_stub_pickle")
_pickle_module.Unpickler = _Unpickler
_pickle_module.load = lambda f, **kw: _Unpickler(f, **kw).load()
_pickle_module.__name__ = "pickle"


def _walk(mod, prefix: str, out: Dict[str, np.ndarray]) -> None:
    d = getattr(mod, "__dict__", {})
    for kind in ("_parameters", "_buffers"):
        for k, v in (d.get(kind) or {}).items():
            if v is not None and torch.is_tensor(v):
                out[f"{prefix}{k}"] = v.detach().float().cpu().numpy()
    for k, child in (d.get("_modules") or {}).items():
        if child is not None:
            _walk(child, f"{prefix}{k}.", out)


def read_checkpoint(path: str) -> Tuple[Dict[str, np.ndarray], dict]:
    """-> (unfused state dict with Ultralytics names, info dict with yaml / names / nc)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_pickle_module)
    model = ckpt.get("ema") or ckpt.get("model") if isinstance(ckpt, dict) else ckpt
    if model is None:
        raise ValueError(f"{path}: no 'model' / 'ema' entry in the checkpoint")
    sd: Dict[str, np.ndarray] = {}
    _walk(model, "", sd)
    if not sd:
        raise ValueError(f"{path}: no tensors found in the checkpoint's module tree")
    info = {"yaml": getattr(model, "yaml", None) or {}, "names": getattr(model, "names", None),
            "nc": getattr(model, "nc", None)}
    return sd, info


def infer_model_name(sd: Dict[str, np.ndarray], info: dict) -> str:
    """Family from the block type of layer 2 (C2f has ``m.0.cv1`` with 3x3, C3 has ``cv3``), scale from the stem
    width, task from the presence of the keypoint branch."""
    c0 = sd["model.0.conv.weight"].shape[0]
    family = "v5u" if "model.2.cv3.conv.weight" in sd else "v8"
    scale = {16: "n", 32: "s", 48: "m", 64: "l", 80: "x"}.get(c0)
    if scale is None:
        raise ValueError(f"unsupported stem width {c0}")
    head = 24 if family == "v5u" else 22
    pose = f"model.{head}.cv4.0.0.conv.weight" in sd
    if any(k.startswith(f"model.{head}.") and (".proto." in k or ".cv4." in k and not pose) for k in sd):
        raise ValueError("segmentation / OBB checkpoints are not supported (detect and pose only)")
    base = f"yolov5{scale}u" if family == "v5u" else f"yolov8{scale}"
    return base + ("-pose" if pose else "")


def convert_pt(path: str) -> bytes:
    sd, info = read_checkpoint(path)
    name = infer_model_name(sd, info)
    head = 24 if name.startswith("yolov5") else 22
    nc = int(sd[f"model.{head}.cv3.0.2.weight"].shape[0])
    names = info.get("names")
    meta = {"source": path}
    if isinstance(names, dict):
        meta["names"] = {str(k): str(v) for k, v in names.items()}
    elif isinstance(names, (list, tuple)):
        meta["names"] = {str(i): str(v) for i, v in enumerate(names)}
    return build_from_state_dict(name, sd, nc=nc, meta=meta)


def main(argv=None) -> None:
    import argparse
    ap = argparse.ArgumentParser(description="Convert an Ultralytics .pt checkpoint to .mi355w")
    ap.add_argument("src")
    ap.add_argument("dst")
    a = ap.parse_args(argv)
    with open(a.dst, "wb") as f:
        f.write(convert_pt(a.src))


if __name__ == "__main__":
    main()
