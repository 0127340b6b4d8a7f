import os
import sys

import pytest

os.environ.setdefault("MI355_CONV_V2", "1")     # tests also cover the opt-in loader-wave conv kernel (read once per process)
os.environ.setdefault("MI355_CONV_V5", "1")     # ... and the opt-in persistent/prefetched 3x3 kernel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def v8n():
    """(program, unfused synthetic state dict) of YOLOv8n detect, seed 0."""
    from tools import synth
    return synth.synthetic_checkpoint("yolov8n", seed=0)


@pytest.fixture(scope="session")
def v8n_pose():
    from tools import synth
    return synth.synthetic_checkpoint("yolov8n-pose", seed=0)
