"""The .pt reader on look-alike checkpoints: module tree pickled under fake ``ultralytics.*`` class paths, fp16 weights."""
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn as nn


def _fabricate(path, sd, names, half=True):
    """Pickle an nn.Module tree whose custom classes live in a (temporary) fake ``ultralytics`` package."""
    fake = {}
    for modname in ("ultralytics", "ultralytics.nn", "ultralytics.nn.tasks", "ultralytics.nn.modules",
                    "ultralytics.nn.modules.conv", "ultralytics.nn.modules.block", "ultralytics.nn.modules.head"):
        fake[modname] = types.ModuleType(modname)
    def cls(mod, name):
        c = type(name, (nn.Module,), {"__module__": mod})
        setattr(fake[mod], name, c)
        return c
    Generic = cls("ultralytics.nn.modules.block", "Block")
    Model = cls("ultralytics.nn.tasks", "DetectionModel")
    sys.modules.update(fake)
    try:
        root = Model()
        for key, arr in sd.items():
            parts = key.split(".")
            m = root
            for p in parts[:-1]:
                if p not in m._modules:
                    m.add_module(p, Generic())
                m = m._modules[p]
            t = torch.from_numpy(np.array(arr))
            if parts[-1] in ("running_mean", "running_var"):
                m.register_buffer(parts[-1], t)
            else:
                m.register_parameter(parts[-1], nn.Parameter(t.half() if half else t, requires_grad=False))
        root.yaml = {"nc": 80}
        root.names = names
        root.nc = 80
        torch.save({"model": root, "ema": None, "epoch": -1}, path)
    finally:
        for k in fake:
            sys.modules.pop(k, None)


def test_convert_fabricated_checkpoint(tmp_path, v8n):
    from cvsd_amd import convert, weights
    prog, sd = v8n
    p = str(tmp_path / "yolov8n.pt")
    _fabricate(p, sd, {0: "person", 1: "bicycle"}, half=True)
    assert "ultralytics" not in sys.modules
    got, info = convert.read_checkpoint(p)
    assert set(got) == set(sd)
    for k in sd:                                        # stored fp16 -> widened to fp32
        want = sd[k] if k.endswith(("running_mean", "running_var")) else sd[k].astype(np.float16).astype(np.float32)
        np.testing.assert_array_equal(got[k], want)
    assert convert.infer_model_name(got, info) == "yolov8n"
    blob = convert.convert_pt(p)
    prog2, fused, meta = weights.from_bytes(blob)
    # 63 module convs; the engine program merges the three (cv2[i][0], cv3[i][0]) sibling pairs of the head: 60 launches
    assert meta["names"]["0"] == "person" and prog2.nc == 80 and len(prog2.convs) == 60
    assert sum(c.cout * c.cin * c.k * c.k + c.cout for c in prog2.convs) == sum(c.cout * c.cin * c.k * c.k + c.cout for c in prog.convs)


def test_infer_model_name_variants():
    from cvsd_amd import convert
    from tools import synth
    for name in ("yolov8n-pose", "yolov5nu"):
        _, sd = synth.synthetic_checkpoint(name, seed=0)
        assert convert.infer_model_name(sd, {}) == name


def test_missing_checkpoint_is_an_error_not_a_download(tmp_path):
    from cvsd_amd import convert
    with pytest.raises(FileNotFoundError):
        convert.read_checkpoint(str(tmp_path / "nope.pt"))
