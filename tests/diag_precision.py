"""Diagnostic: error of the GPU engine and of the fp32 CPU oracle against a float64 execution of the same program."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools import synth, program_ref as PR
from cvsd_amd import weights, YOLO
from oracle import yolo_oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "yolov8n"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
prog, sd = synth.synthetic_checkpoint(name, seed=0)
fused = weights.fuse_state_dict(prog, sd)
frames = synth.synthetic_frames(n, 640, 640, seed=5)
x = O.preprocess(list(frames), 640)
om = O.OracleModel(name, sd)
pred_o = om.forward(x).numpy()
ex = PR.ProgramExecutor(prog, np.float64); names = [c.name for c in prog.convs]
ex.run(x.permute(0, 2, 3, 1).numpy().astype(np.float64), lambda ci, src: fused[names[ci]])
pred_t = PR.decode_head(prog, ex.head_maps())
m = YOLO.from_state_dict(name, sd)
pred_g = m.raw_head(frames)
nc = prog.nc
def rep(tag, a, b):
    d = np.abs(a - b)
    print(f"{tag:18s} box max {d[:, :4].max():.3e} p99.9 {np.quantile(d[:, :4], 0.999):.3e} mean {d[:, :4].mean():.3e} | "
          f"score max {d[:, 4:4+nc].max():.3e}" + (f" | kpt max {d[:, 4+nc:].max():.3e}" if prog.nk else ""))
rep("oracle32 vs f64", pred_o, pred_t)
rep("gpu vs f64", pred_g, pred_t)
rep("gpu vs oracle32", pred_g, pred_o)
for lo, hi, nm in [(0, 6400, "P3"), (6400, 8000, "P4"), (8000, 8400, "P5")]:
    print(nm, "gpu-f64 box max", np.abs(pred_g - pred_t)[:, :4, lo:hi].max(), "oracle-f64", np.abs(pred_o - pred_t)[:, :4, lo:hi].max())
