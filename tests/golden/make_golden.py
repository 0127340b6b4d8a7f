"""Generate the committed golden vectors under tests/golden/ (run in the build container; the outputs travel).

    python tests/golden/make_golden.py

Everything is produced by the oracles (oracle/det.py canonical-order C kernels, oracle/yolo_oracle.py) from seeded
inputs and the seeded synthetic checkpoints -- the reference ships no fixtures for this path.
Inputs are regenerated from their seeds at test time; only expected outputs (and small inputs) are stored."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import det, yolo_oracle as O  # noqa: E402
from tools import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def ckpt_hash(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(sd[k].tobytes())
    return h.hexdigest()


def main():
    g = {}
    # checkpoints: hashes pin the generator
    for name in ("yolov8n", "yolov8n-pose"):
        prog, sd = synth.synthetic_checkpoint(name, seed=0)
        g[f"ckpt_sha256/{name}"] = np.frombuffer(ckpt_hash(sd).encode(), dtype=np.uint8)
    # conv tiles (canonical-order oracle)
    rng = np.random.default_rng(1234)
    for tag, (n, h, w, cin, cout, k, s, act, res) in {
        "conv3x3_s1_res": (1, 12, 20, 32, 48, 3, 1, 1, 1), "conv3x3_s2": (2, 16, 16, 16, 32, 3, 2, 1, 0),
        "conv1x1_51": (1, 8, 8, 51, 51, 1, 1, 0, 0)}.items():
        x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
        wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        r = rng.standard_normal((n, h // s, w // s, cout), dtype=np.float32) if res else None
        g[f"{tag}/x"], g[f"{tag}/w"], g[f"{tag}/b"] = x, wt, b
        if res:
            g[f"{tag}/res"] = r
        g[f"{tag}/meta"] = np.array([k, s, act], np.int32)
        g[f"{tag}/y"] = det.conv2d(x, wt, b, stride=s, act=bool(act), residual=r)
    # det_expf table
    xs = np.concatenate([np.linspace(-104, 89, 387), np.array([-1e-3, 0.0, 1e-3, 0.5, -0.5])]).astype(np.float32)
    g["expf/x"], g["expf/y"] = xs, det.expf(xs)
    # letterbox (integer work): a 30x40 frame -> imgsz 64 (resize + pad), and 48x48 -> 64 (resize only)
    for tag, (h, w) in {"lb_30x40": (30, 40), "lb_48x48": (48, 48), "lb_100x37": (100, 37)}.items():
        f = np.random.default_rng(h * w).integers(0, 256, (h, w, 3), dtype=np.uint8)
        g[f"{tag}/in"], g[f"{tag}/out"] = f, O.letterbox(f, (64, 64))
    # full net, small: pre-NMS head tensor for 2 frames of 64x96 (imgsz 96)
    for name in ("yolov8n", "yolov8n-pose"):
        prog, sd = synth.synthetic_checkpoint(name, seed=0)
        dm = det.DetOracleModel(name, sd)
        frames = synth.synthetic_frames(2, 64, 96, seed=11)
        want, pred = det.predict(dm, list(frames), conf=0.25, imgsz=96)
        g[f"head_64x96/{name}"] = pred.numpy()
        # post-NMS rows at 640x640 (2 frames, seed 21)
        frames = synth.synthetic_frames(2, 640, 640, seed=21)
        want, _ = det.predict(dm, list(frames), conf=0.25, imgsz=640)
        for i, r in enumerate(want):
            g[f"rows_640/{name}/{i}/boxes"] = r["boxes"].numpy()
            g[f"rows_640/{name}/{i}/anchors"] = r["anchor_idx"].numpy().astype(np.int32)
            if r["kpts"] is not None:
                g[f"rows_640/{name}/{i}/kpts"] = r["kpts"].numpy()
    np.savez_compressed(os.path.join(OUT, "golden_v1.npz"), **g)
    print("wrote", os.path.join(OUT, "golden_v1.npz"), os.path.getsize(os.path.join(OUT, "golden_v1.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
