"""Experiment: one engine on n frames per pass against TWO engines (two host threads, two HIP streams) on n/2 frames each.
Dependent kernels of one chain cannot overlap their tails; two independent chains can fill each other's ramps and tails.
    python tools/two_chains.py yolov8m 1280 16 1      # model size batch half
"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvsd_amd import YOLO
from cvsd_amd.weights import build_from_state_dict
from tools import synth

name, size, batch, half = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4]))
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
_, sd = synth.synthetic_checkpoint(name, seed=0)
blob = build_from_state_dict(name, sd)
g = torch.Generator().manual_seed(1)
frames = torch.randint(0, 256, (batch, size, size, 3), dtype=torch.uint8, generator=g).cuda()


def run(models, parts, steps):
    def work(m, f):
        for _ in range(steps):
            m._infer_rows(f, 0.25, 0.7, None, 300, size)
    th = [threading.Thread(target=work, args=(m, f)) for m, f in zip(models, parts)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


one = YOLO(blob, batch_chunk=batch, half=half)
run([one], [frames], 3)
dt1 = run([one], [frames], steps)
print(f"{name} {size} batch {batch} half={half}: one chain  {batch * steps / dt1:8.1f} frames/s  {dt1 / steps * 1e3:.3f} ms/step")
h = batch // 2
if h >= 1:
    two = [YOLO(blob, batch_chunk=h, half=half) for _ in range(2)]
    parts = [frames[:h].contiguous(), frames[h:2 * h].contiguous()]
    run(two, parts, 3)
    dt2 = run(two, parts, steps)
    print(f"{name} {size} batch 2 x {h} half={half}: two chains {2 * h * steps / dt2:8.1f} frames/s  {dt2 / steps * 1e3:.3f} ms/step")
    dt1b = run([one], [frames], steps)
    print(f"  (one chain again {batch * steps / dt1b:8.1f} frames/s)")
