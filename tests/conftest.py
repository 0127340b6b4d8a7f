import os
import sys

import pytest




ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # Launch-plan choices are persisted per (model, shape); the suite must not depend on -- or leave behind -- machine state in
    # ~/.cache/mi355yolo: every session tunes into its own directory (child processes such as bench.py inherit it).
    import tempfile
    if "MI355_PLAN_CACHE" not in os.environ:
        os.environ["MI355_PLAN_CACHE"] = tempfile.mkdtemp(prefix="mi355plans_")


@pytest.fixture(scope="session")
def v8n():
    """(program, unfused synthetic state dict) of YOLOv8n detect, seed 0."""
    from tools import synth
    return synth.synthetic_checkpoint("yolov8n", seed=0)


@pytest.fixture(scope="session")
def v8n_pose():
    from tools import synth
    return synth.synthetic_checkpoint("yolov8n-pose", seed=0)
