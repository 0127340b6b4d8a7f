"""Diagnostic: error statistics of the half=True engine against (a) the oracle's half-storage restatement and
(b) the fp32 oracle, on the pre-NMS head tensor.  Run on the GPU box; numbers go into tests/test_gpu_half.py."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvsd_amd import YOLO
from oracle import yolo_oracle as O
from tools import synth


def stats(name, d):
    d = np.abs(d).ravel()
    print(f"    {name:8s} max {d.max():.4g}  p99.9 {np.quantile(d, 0.999):.4g}  p99 {np.quantile(d, 0.99):.4g}  median {np.median(d):.4g}")


for name, n, size in [("yolov8n", 2, 640), ("yolov8n-pose", 2, 640), ("yolov8s-pose", 1, 320), ("yolov8m", 1, 1280)]:
    ckpt = synth.synthetic_checkpoint(name, seed=0)
    m = YOLO.from_state_dict(name, ckpt[1], half=True)
    frames = synth.synthetic_frames(n, size, size, seed=31)
    got = m.raw_head(frames, imgsz=size)
    x = O.preprocess(list(frames), size)
    for label, half in (("half-oracle", True), ("fp32-oracle", False)):
        om = O.OracleModel(name, ckpt[1], half=half)
        want = om.forward(x).numpy()
        nc = om.nc
        print(f"{name} {size} vs {label}")
        stats("box", got[:, :4] - want[:, :4])
        stats("score", got[:, 4:4 + nc] - want[:, 4:4 + nc])
        if om.pose:
            k = got[:, 4 + nc:] - want[:, 4 + nc:]
            stats("kpt xy", np.delete(k, np.s_[2::3], axis=1))
            stats("kpt conf", k[:, 2::3])
