import numpy as np
import pytest
import torch

from cvsd_amd import weights
from cvsd_amd.graph import build_program


def test_fold_is_fuse_conv_and_bn(v8n):
    """fold == Ultralytics' fuse_conv_and_bn evaluated in float64, to fp32 rounding; bitwise equal to the
    canonical-order oracle's own statement of it"""
    from oracle import det
    prog, sd = v8n
    fused = weights.fuse_state_dict(prog, sd)
    dm = det.DetOracleModel("yolov8n", sd)
    for c in prog.convs:
        if not c.has_bn:
            continue
        w64 = sd[c.name + ".conv.weight"].astype(np.float64)
        s64 = sd[c.name + ".bn.weight"].astype(np.float64) / np.sqrt(1e-3 + sd[c.name + ".bn.running_var"].astype(np.float64))
        np.testing.assert_allclose(fused[c.name][0], w64 * s64[:, None, None, None], rtol=3e-7, atol=1e-12)
        wf, bf = dm._fused_conv(c.name)
        assert np.array_equal(wf.numpy(), fused[c.name][0]) and np.array_equal(bf.numpy(), fused[c.name][1])


def test_torch_oracle_fold_within_2ulp(v8n):
    from oracle import yolo_oracle as O
    prog, sd = v8n
    fused = weights.fuse_state_dict(prog, sd)
    om = O.OracleModel("yolov8n", sd)
    for c in prog.convs[:20]:
        wf, _ = om._fused_conv(c.name)
        np.testing.assert_allclose(wf.numpy(), fused[c.name][0], rtol=3e-7, atol=0)


def test_file_roundtrip_and_header(v8n_pose):
    prog, sd = v8n_pose
    fused = weights.fuse_state_dict(prog, sd)
    blob = weights.to_bytes(prog, fused, {"model": "yolov8n-pose", "names": {"0": "person"}})
    assert blob[:8] == b"MI355YW1" and len(blob) % 256 == 0
    p2, f2, meta = weights.from_bytes(blob)
    assert meta["names"] == {"0": "person"} and p2.task == 1 and p2.nkpt == 17 and p2.kdim == 3
    assert [(c.name, c.cin, c.cout, c.k, c.s) for c in p2.convs] == [(c.name, c.cin, c.cout, c.k, c.s) for c in prog.convs]
    assert len(p2.ops) == len(prog.ops) and p2.buffers == [tuple(b) for b in prog.buffers]
    for k in fused:
        assert np.array_equal(f2[k][0], fused[k][0]) and np.array_equal(f2[k][1], fused[k][1])
    with pytest.raises(ValueError):
        weights.from_bytes(b"garbage!" + blob[8:])


def test_shape_mismatch_is_reported(v8n):
    prog, sd = v8n
    bad = dict(sd)
    bad["model.1.conv.weight"] = bad["model.1.conv.weight"][:, :8]
    with pytest.raises(ValueError, match="model.1"):
        weights.fuse_state_dict(prog, bad)


def test_synthetic_checkpoint_is_deterministic(v8n):
    """the golden vectors depend on it: same seed -> same bits (also across hosts: golden file holds the sha256)"""
    import hashlib
    import os
    prog, sd = v8n
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(sd[k].tobytes())
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
    assert h.hexdigest() == bytes(g["ckpt_sha256/yolov8n"]).decode()
