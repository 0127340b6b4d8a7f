#!/usr/bin/env python
"""Generate tests/golden/poselift_fixture.npz -- runs in the BUILD CONTAINER ONLY (it imports the reference).

What it pins: the pickle tree written by ``cvsd_amd.poselift_bridge`` is read by the
REFERENCE'S OWN loader, ``/root/reference/shopformer/data/poselift_dataset.py:PoseLiftDataset`` (``_load_data``
``:231-254``, ``_extract_sequences`` ``:256-323``), and what that loader returns is stored as data:

    pose detections  : YOLOv8n-pose synthetic checkpoint (tools/synth.py, seed 0) evaluated by the canonical-order CPU
                       oracle (oracle/det.py -- the GPU engine is bit-identical to it, so the GPU test can demand equality)
    tracker + writer : the product's own video_to_poselift (tracker.py, PoseLiftWriter), fed by the oracle through a
                       predict() adapter
    loader           : PoseLiftDataset(data_dir, split='train' and 'test', seq_len 12, stride 6), imported from
                       /root/reference -- nothing of it is copied; only its OUTPUT tensors are stored
    second loader    : /root/reference/shopformer_2/data/poselift_dataset.py:PoseLiftDataset with num_keypoints=18 (the synthetic
                       neck keypoint of add_neck_keypoint, :57-91, appended to the 17 COCO joints) on the same pickle tree; its
                       windows [C, T, 18] are stored under the "s2_" keys -- a second reference-held pin of the bridge's format

Stored: the frames' seed/geometry, the bridge dict (flattened arrays), and per split the loader's (n_samples, every
window tensor [C,T,V], labels).  tests/test_poselift_fixture.py (CPU) re-reads the bridge dict with a restated loader
and must reproduce the reference loader's windows exactly; tests/test_gpu_pipeline.py (GPU) runs the real engine through
the same bridge and must reproduce the stored dict bit for bit.

    python tests/golden/make_poselift_fixture.py
"""
from __future__ import annotations

import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF_LOADER = "/root/reference/shopformer/data/poselift_dataset.py"
REF_LOADER_2 = "/root/reference/shopformer_2/data/poselift_dataset.py"
OUT = os.path.join(ROOT, "tests", "golden", "poselift_fixture.npz")

MODEL, SEED_W = "yolov8n-pose", 0
N_FRAMES, H, W, SEED_F, IMGSZ, CONF, BATCH = 20, 192, 256, 91, 256, 0.25, 8
SEQ_LEN, STRIDE = 12, 6


def clip_frames():
    from tools import synth
    return synth.synthetic_clip(N_FRAMES, H, W, seed=SEED_F)


class OracleAsModel:
    """predict() surface of cvsd_amd.YOLO backed by the canonical-order CPU oracle."""
    task = "pose"

    def __init__(self):
        from oracle import det
        from tools import synth
        _, sd = synth.synthetic_checkpoint(MODEL, seed=SEED_W)
        self.det, self.om = det, det.DetOracleModel(MODEL, sd)

    def predict(self, batch, conf=0.25, imgsz=640, **kw):
        from cvsd_amd.results import Results
        want, _ = self.det.predict(self.om, list(batch), conf=conf, imgsz=imgsz)
        out = []
        for w, f in zip(want, batch):
            r = Results(None, "f.jpg", {0: "person"}, boxes=w["boxes"].clone(), keypoints=w["kpts"].clone(), orig_shape=f.shape[:2])
            r.keypoints_raw = w["kpts"].numpy().copy()
            out.append(r)
        return out


def flatten(data):
    """{frame: {pid: [bbox(4), kpts(17,3)]}} -> arrays (frame, pid, bbox, kpts) in dict iteration order + frame list"""
    fr, pid, bb, kp = [], [], [], []
    for f, people in data.items():
        for p, (b, k) in people.items():
            fr.append(f); pid.append(p); bb.append(b); kp.append(k)
    return (np.asarray(list(data.keys()), np.int64), np.asarray(fr, np.int64), np.asarray(pid, np.int64),
            np.asarray(bb, np.float32).reshape(-1, 4), np.asarray(kp, np.float32).reshape(-1, 17, 3))


def main():
    from cvsd_amd.poselift_bridge import video_to_poselift
    spec = importlib.util.spec_from_file_location("ref_poselift_dataset", REF_LOADER)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)                       # the reference's own module, imported where it lies
    spec2 = importlib.util.spec_from_file_location("ref_poselift_dataset_2", REF_LOADER_2)
    ref2 = importlib.util.module_from_spec(spec2)
    spec2.loader.exec_module(ref2)                     # ... and the second loader (18 keypoints: COCO-17 + neck)

    frames = clip_frames()
    model = OracleAsModel()
    with tempfile.TemporaryDirectory() as root:
        labels = (np.arange(N_FRAMES) >= N_FRAMES // 2).astype(np.int64)        # GT: second half is 'shoplifting'
        for split in ("Train", "Test"):
            os.makedirs(os.path.join(root, "Pickle_files", split))
        os.makedirs(os.path.join(root, "Pickle_files", "GT"))
        data = video_to_poselift(model, list(frames), out_path=os.path.join(root, "Pickle_files", "Train", "cam1_0001.pkl"),
                                 conf=CONF, batch=BATCH, imgsz=IMGSZ)
        video_to_poselift(model, list(frames), out_path=os.path.join(root, "Pickle_files", "Test", "cam1_0001.pkl"),
                          conf=CONF, batch=BATCH, imgsz=IMGSZ)
        np.save(os.path.join(root, "Pickle_files", "GT", "cam1_0001.npy"), labels)
        store = {}
        for split in ("train", "test"):
            for inc in (False, True):
                ds = ref.PoseLiftDataset(root, split=split, seq_len=SEQ_LEN, stride=STRIDE, normalize=True, include_confidence=inc)
                key = f"{split}_{'xyc' if inc else 'xy'}"
                items = [ds[i] for i in range(len(ds))]
                store[key + "_n"] = np.int64(len(ds))
                store[key + "_x"] = (torch.stack([x for x, _ in items]).numpy() if items else np.zeros((0, 3 if inc else 2, SEQ_LEN, 17), np.float32))
                store[key + "_y"] = np.asarray([int(y) for _, y in items], np.int64)
                ds2 = ref2.PoseLiftDataset(root, split=split, seq_len=SEQ_LEN, stride=STRIDE, num_keypoints=18, normalize=True,
                                           include_confidence=inc)
                items2 = [ds2[i] for i in range(len(ds2))]
                store["s2_" + key + "_n"] = np.int64(len(ds2))
                store["s2_" + key + "_x"] = (torch.stack([x for x, _ in items2]).numpy() if items2
                                             else np.zeros((0, 3 if inc else 2, SEQ_LEN, 18), np.float32))
                store["s2_" + key + "_y"] = np.asarray([int(y) for _, y in items2], np.int64)
    keys, fr, pid, bb, kp = flatten(data)
    assert store["train_xy_n"] > 0 and store["s2_train_xy_n"] > 0, "no 12-frame window: the fixture would pin nothing"
    np.savez_compressed(OUT, frame_keys=keys, row_frame=fr, row_pid=pid, row_bbox=bb, row_kpts=kp, gt=labels,
                        meta=np.asarray([N_FRAMES, H, W, SEED_F, IMGSZ, BATCH, SEQ_LEN, STRIDE], np.int64), conf=np.float64(CONF), **store)
    print(f"wrote {OUT}: {len(fr)} person rows over {len(keys)} frames; windows: "
          + ", ".join(f"{k[:-2]}={int(v)}" for k, v in store.items() if k.endswith("_n")))


if __name__ == "__main__":
    main()
