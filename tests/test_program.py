"""The op program (channel-slice plumbing of graph.py) computes the same function as the module-by-module oracle."""
import numpy as np
import pytest

from cvsd_amd import weights
from tools import program_ref as PR
from tools import synth


@pytest.mark.parametrize("name", ["yolov8n", "yolov8n-pose", "yolov5nu"])
def test_program_matches_module_oracle(name):
    from oracle import yolo_oracle as O
    prog, sd = synth.synthetic_checkpoint(name, seed=0)
    fused = weights.fuse_state_dict(prog, sd)
    frames = synth.synthetic_frames(2, 96, 128, seed=3)
    x = O.preprocess(list(frames), 128)
    want = O.OracleModel(name, sd).forward(x).numpy()
    ex = PR.ProgramExecutor(prog, np.float64)
    names = [c.name for c in prog.convs]
    ex.run(x.permute(0, 2, 3, 1).numpy().astype(np.float64), lambda ci, src: fused[names[ci]])
    got = PR.decode_head(prog, ex.head_maps())
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-2          # float64 program vs fp32 torch: re-association noise only
    assert np.abs(got - want).mean() < 2e-4
