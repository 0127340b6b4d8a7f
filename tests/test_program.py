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


@pytest.mark.parametrize("name", ["yolov8n", "yolov8n-pose", "yolov5nu"])
def test_merged_sibling_convs_compute_the_same_function(name):
    """graph.merge_sibling_convs (what weights.build_from_state_dict ships to the engine): the head's first 3x3 convs of a
    level share their input, so they run as one conv with concatenated output channels; consumers read slices.  Same
    parameters, same MACs, fewer ops and buffers, and the float64 program output is unchanged (each output channel is the
    same dot product; the GPU tests hold the engine to bit-identity with the UNMERGED module-by-module oracle)."""
    from cvsd_amd.graph import OP_CONV, merge_sibling_convs
    from oracle import yolo_oracle as O
    prog, sd = synth.synthetic_checkpoint(name, seed=0)
    fused = weights.fuse_state_dict(prog, sd)
    prog2, fused2 = merge_sibling_convs(prog, fused)
    n_branches = 3 if prog.nk else 2
    assert len(prog2.ops) == len(prog.ops) - 3 * (n_branches - 1) and len(prog2.convs) == len(prog.convs) - 3 * (n_branches - 1)
    assert len(prog2.buffers) == len(prog.buffers) - 3 * (n_branches - 1)
    assert prog2.param_count() == prog.param_count() and prog2.macs() == prog.macs()
    assert set(fused2) == {c.name for c in prog2.convs} and sum(1 for c in prog2.convs if "+" in c.name) == 3
    for o in prog2.ops:                                                    # every view stays inside its buffer, 4-channel aligned
        for v in (o.src, o.dst, o.res):
            if v is not None:
                assert v.choff % 4 == 0 and v.choff + v.c <= prog2.buffers[v.buf][0]
    for buf in range(len(prog2.buffers)):                                  # one writer per channel: no WAW hazard for the scheduler
        spans = sorted((o.dst.choff, o.dst.choff + (3 * o.src.c if o.type == 3 else o.dst.c)) for o in prog2.ops if o.dst.buf == buf)
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), (buf, spans)
    frames = synth.synthetic_frames(1, 96, 128, seed=3)
    x = O.preprocess(list(frames), 128).permute(0, 2, 3, 1).numpy().astype(np.float64)
    outs = []
    for p, f in ((prog, fused), (prog2, fused2)):
        ex = PR.ProgramExecutor(p, np.float64)
        names = [c.name for c in p.convs]
        ex.run(x, lambda ci, src: f[names[ci]])
        outs.append(PR.decode_head(p, ex.head_maps()))
    np.testing.assert_allclose(outs[1], outs[0], rtol=0, atol=1e-9)
    # the file round trip keeps the merged program
    p3, f3, _ = weights.from_bytes(weights.to_bytes(prog2, fused2))
    assert [c.name for c in p3.convs] == [c.name for c in prog2.convs]
    np.testing.assert_array_equal(f3[prog2.convs[-1].name][0], fused2[prog2.convs[-1].name][0])
