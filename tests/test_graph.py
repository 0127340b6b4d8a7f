"""Structural pins of the op program: fused parameter counts and GFLOPs equal the public Ultralytics model cards."""
import pytest

from cvsd_amd.graph import OP_CONV, OP_SPPF_POOL, OP_STEM, OP_UPSAMPLE, build_program, parse_model_name

# "YOLOv8n summary (fused): ... 3,151,904 parameters, 8.7 GFLOPs" etc. (public model cards / SURVEY.md 8(d))
CARDS = {
    "yolov8n": (3151904, 8.7), "yolov8s": (11156544, 28.6), "yolov8m": (25886080, 78.9), "yolov8l": (43668288, 165.2),
    "yolov8x": (68200608, 257.8), "yolov8n-pose": (3289964, 9.2), "yolov8s-pose": (11615724, 30.2),
    "yolov8m-pose": (26447596, 81.0),
    "yolov5nu": (2649200, 7.7), "yolov5su": (9142496, 24.0), "yolov5mu": (25091536, 64.2),
}


@pytest.mark.parametrize("name", sorted(CARDS))
def test_param_count_and_gflops_match_model_card(name):
    prog = build_program(*parse_model_name(name))
    params, gflops = CARDS[name]
    assert prog.param_count() == params
    assert abs(2 * prog.macs() / 1e9 - gflops) < 0.06


def test_survey_table_bytes():
    """SURVEY.md 8(d) 'layerwise bytes' column"""
    assert round(build_program("v8", "n", "detect").act_bytes() / 1e6, 1) == 140.3
    assert round(build_program("v8", "s", "pose").act_bytes() / 1e6, 1) == 269.6
    assert build_program("v8", "n", "detect").num_anchors(640, 640) == 8400


@pytest.mark.parametrize("name", ["yolov8n", "yolov8m-pose", "yolov5mu"])
def test_program_is_well_formed(name):
    prog = build_program(*parse_model_name(name))
    written = {}
    for op in prog.ops:
        if op.type in (OP_CONV, OP_UPSAMPLE, OP_SPPF_POOL):
            for c in range(op.src.choff, op.src.choff + op.src.c):       # every read channel was produced earlier
                assert (op.src.buf, c) in written, (name, op)
        if op.res is not None:
            assert all((op.res.buf, c) in written for c in range(op.res.choff, op.res.choff + op.dst.c))
        n_out = 3 * op.src.c if op.type == OP_SPPF_POOL else op.dst.c
        assert op.dst.choff % 4 == 0 and op.dst.choff + n_out <= prog.buffers[op.dst.buf][0]
        for c in range(op.dst.choff, op.dst.choff + n_out):
            assert (op.dst.buf, c) not in written, "channel written twice"
            written[(op.dst.buf, c)] = True
        if op.type in (OP_CONV, OP_STEM):
            cv = prog.convs[op.conv]
            assert cv.cout == op.dst.c and (op.type == OP_STEM or cv.cin == op.src.c)
    assert len(prog.levels) == 3 and [lv.stride for lv in prog.levels] == [8, 16, 32]


def test_parse_model_name_rejects_unknown():
    with pytest.raises(ValueError):
        parse_model_name("resnet50")
