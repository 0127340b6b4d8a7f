#!/usr/bin/env python
"""Throughput of the per-frame YOLO hot path on N MI355X (frames/s @ 640x640), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A *step* is one pass of the whole hot path (letterbox/stem -> conv graph -> head decode -> NMS -> scale-back ->
rows on the host, plus -- for N > 1 -- the per-step gather of rows to rank 0) over one batch of synthetic
640x640 BGR frames per GPU.  Frames are resident in HBM when the timed region starts.  Weak scaling: the
per-GPU batch is fixed.  Rank 0 prints ONE JSON line (see the contract in the task statement).

The headline (`value`) is BASELINE.json's metric on configs[1]'s model and size at the batch that fills the chip (512 per GPU).
At N = 1 the same line carries `configs`: every BASELINE configuration as stated (batch 1 / 32 / 8 and 64 / 1280-half 2 and
16), each with its own throughput and conv-stack roofline fraction, and `host_fed_value`: the headline workload fed from
pinned host memory (PCIe-inclusive; never `value`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3        # MI355X fp32 matrix (= vector) peak, /opt/skills/guides/MI355X_MICROARCH.md
F16_PEAK_TFLOPS = 2500.0        # dense f16 MFMA peak (same guide; the sparsity figure is not used)
N_BASE_FRAMES = 16              # frames made by tools/synth.py on the host (the CPU baseline / parity sample comes from these)

# BASELINE.json configs 2-5 as stated, per GPU (config 4 shards batch 64 over 8 GPUs: 8 per GPU, and 64 per GPU for the
# throughput form; config 5 shards batch 16 over 8 GPUs: 2 per GPU, and 16 on one GPU): (model, size, batch, half, steps, warmup)
EXTRA_CONFIGS = [
    ("yolov8n", 640, 1, False, 600, 100),
    ("yolov8n-pose", 640, 32, False, 40, 5),
    ("yolov8s-pose", 640, 8, False, 60, 8),
    ("yolov8s-pose", 640, 64, False, 12, 3),
    ("yolov8m", 1280, 2, True, 40, 5),
    ("yolov8m", 1280, 16, True, 10, 3),
]


def make_frames(n: int, size: int, seed: int):
    """-> (uint8 cuda tensor [n, size, size, 3], the first min(n, 16) frames as numpy).  The first 16 frames are
    tools/synth.py's; the rest are per-frame noisy variants made on the GPU, so that all n frames are distinct bytes
    (n x 1.23 MB does not fit the 256 MiB Infinity Cache at n = 512: the stem reads HBM, as in production)."""
    import torch
    from tools import synth
    base_np = synth.synthetic_frames(min(n, N_BASE_FRAMES), size, size, seed=seed)
    base = torch.from_numpy(base_np).cuda()
    if n <= len(base_np):
        return base[:n].contiguous(), base_np[:n]
    frames = torch.empty((n, size, size, 3), dtype=torch.uint8, device="cuda")
    frames[:len(base_np)] = base
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    b16 = base.to(torch.int16)
    for s in range(len(base_np), n, len(base_np)):
        m = min(len(base_np), n - s)
        noise = torch.randint(-24, 25, (m, size, size, 3), generator=gen, device="cuda", dtype=torch.int16)
        frames[s:s + m] = (b16[:m] + noise).clamp_(0, 255).to(torch.uint8)
    torch.cuda.synchronize()
    return frames, base_np


def conv_flops_per_frame(model_name: str, size: int) -> float:
    from cvsd_amd.graph import build_program, parse_model_name
    pg = build_program(*parse_model_name(model_name))
    return 2.0 * sum(c.cout * c.cin * c.k * c.k * (640 // c.stride_div) ** 2 for c in pg.convs if c.cin != 3) * (size / 640.0) ** 2


def profile_convs(model, frames, size: int, steps: int = 3):
    """Per-kind device time of a step from HIP events on the engine's own stream (profiling mode: one in-order stream)."""
    model.set_profiling(True)
    conv_ms, launches, kinds = 0.0, 0, {}
    for _ in range(steps):
        model._infer_rows(frames, 0.25, 0.7, None, 300, size)
        t = model.last_timing()
        conv_ms += t["conv_ms"]
        launches += t["conv_launches"]
        for k in ("stem_ms", "conv_ms", "pool_ms", "upsample_ms", "letterbox_ms", "decode_ms", "nms_ms"):
            kinds[k] = kinds.get(k, 0.0) + t[k] / steps
    model.set_profiling(False)
    return conv_ms / steps, launches // steps, kinds


def measure_config(model_name: str, size: int, batch: int, half: bool, steps: int, warmup: int) -> dict:
    """One BASELINE configuration on this GPU: frames resident in HBM, rows returned to the host every step."""
    import torch
    from cvsd_amd import YOLO
    from cvsd_amd.weights import build_from_state_dict
    from tools import synth
    _, sd = synth.synthetic_checkpoint(model_name, seed=0)
    model = YOLO(build_from_state_dict(model_name, sd), device=torch.cuda.current_device(), batch_chunk=batch, half=half)
    frames, _ = make_frames(batch, size, seed=2000 + batch)
    for _ in range(warmup):
        model._infer_rows(frames, 0.25, 0.7, None, 300, size)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model._infer_rows(frames, 0.25, 0.7, None, 300, size)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    conv_ms, launches, _ = profile_convs(model, frames, size, steps=2)
    achieved = conv_flops_per_frame(model_name, size) * batch / (conv_ms * 1e-3) / 1e12
    peak = F16_PEAK_TFLOPS if half else FP32_PEAK_TFLOPS
    del model, frames
    torch.cuda.empty_cache()
    return {"workload": f"{model_name} {size}x{size} batch {batch}" + (" half=True" if half else ""), "dtype": "f16" if half else "f32",
            "value": round(batch * steps / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "roofline": {"achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "launches_per_step": launches}}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="yolov8n")
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU per step")
    ap.add_argument("--chunk", type=int, default=0, help="engine batch_chunk: frames per pass through the net (default: the batch)")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-configuration measurements and host_fed_value (headline only)")
    ap.add_argument("--host-frames", action="store_true",
                    help="diagnostic: hand the engine HOST frames each step (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--half", action="store_true",
                    help="BASELINE config 5 mode: the half=True engine (fp16 storage, fp32 accumulate); the line then says "
                         "dtype f16 and prices the convs against the dense f16 MFMA peak.  Never the default headline.")
    args = ap.parse_args()
    if args.chunk <= 0:
        args.chunk = args.batch

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the hot path)")
    # BENCH_DRYRUN_ONE_GPU=1 (tests only): all ranks share cuda:0 and the collectives run over gloo, to rehearse the
    # N > 1 code path on a one-GPU box.  The real launch is one rank per GPU over RCCL (backend "nccl").
    dry = os.environ.get("BENCH_DRYRUN_ONE_GPU") == "1"
    if dry:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if dry else "nccl"
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from cvsd_amd import YOLO
    from cvsd_amd import dist as cdist
    from cvsd_amd.weights import build_from_state_dict
    from tools import synth

    # ---- C1: rank 0 builds the synthetic checkpoint, everyone receives the weight image over RCCL ----
    blob = None
    prog, sd = None, None
    if rank == 0:
        prog, sd = synth.synthetic_checkpoint(args.model, seed=0)
        blob = build_from_state_dict(args.model, sd)
    blob = cdist.broadcast_weights(blob)
    model = YOLO(blob, device=local_rank, batch_chunk=args.chunk, half=args.half)
    plan_warm = world > 1          # rank 0 tunes its launch plans first and persists them; the others then load the file
    layers, params, _, gflops = model.info()

    # ---- synthetic frames of this rank's shard, resident in HBM: B distinct frames ----
    B = args.batch
    frames, frames_np = make_frames(B, args.size, seed=1000 + rank)
    step_src = frames
    if args.host_frames:
        step_src = frames.cpu().pin_memory().numpy()

    ncols = 7 + model.kpt_shape[0] * model.kpt_shape[1]
    # N > 1: rows stay in HBM (infer_async) and are gathered to rank 0 over RCCL one step behind the engine, so step k's
    # C2 + C3 overlap step k+1's kernels; the timed region ends with the last step's gather (flush).
    gath = cdist.DeviceRowGather(model, B, max_det=300, ncols=ncols, conf=0.25, iou=0.7, imgsz=args.size) if world > 1 and not args.host_frames else None

    def step():
        if gath is not None:
            return gath.submit(frames)
        res = model._infer_rows(step_src, 0.25, 0.7, None, 300, args.size)
        if world > 1:
            rows, counts, _ = res
            return cdist.gather_rows(rows, counts, ncols=ncols)
        return res

    def finish():
        if gath is not None:
            gath.flush()
            model.sync()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if plan_warm:
        if rank == 0:
            model._infer_rows(frames, 0.25, 0.7, None, 300, args.size)
        dist.barrier()
    for _ in range(args.warmup):
        step()
    finish()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dry else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline of the dominant kernel (implicit-GEMM conv), HIP events on the engine's stream ----
    PROF_STEPS = 3
    conv_ms, launches, kinds = profile_convs(model, frames, args.size, PROF_STEPS)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    conv_flops_frame = conv_flops_per_frame(args.model, args.size)
    scale = (args.size / 640.0) ** 2
    achieved = conv_flops_frame * B / (conv_ms * 1e-3) / 1e12
    fps = world * B * args.steps / dt
    # HBM traffic of the conv launches from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this
    # process); only quoted when it was collected on this exact workload
    traffic = None
    for tname in ("r02_conv_traffic.json", "r01_conv_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath) and args.model == "yolov8n" and args.size == 640 and B == 512 and args.chunk == 512 and not args.half:
            with open(tpath) as f:
                traffic = json.load(f)["hbm_bytes_per_launch_avg"]
            break
    peak = F16_PEAK_TFLOPS if args.half else FP32_PEAK_TFLOPS
    line = {
        "metric": f"frames/s @{args.size}x{args.size}" + (" (half=True engine: diagnostic)" if args.half else "") + (" (HOST frames, PCIe-inclusive: diagnostic)" if args.host_frames else ""), "value": round(fps, 1), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if args.half else "f32", "data": "synthetic",
        "config": {"workload": f"{args.model} {args.size}x{args.size} synthetic BGR frames ({B} distinct frames per GPU), batch {B}/GPU/step, "
                               f"predict conf=0.25 iou=0.7 max_det=300 (letterbox+stem, conv graph, decode, NMS, rows to host)",
                   "global_batch": B * world, "params": params, "gflop_per_frame": round(gflops * scale, 3),
                   "parallelism": f"frame-sharded dp{world}", "collectives_backend": backend, "ranks": world},
        "roofline": {"bound": "mfma", "kernel": ("conv_igemm_f16 + conv1x1_pipe_f16" if args.half else "conv_igemm_f32 + conv1x1_pipe_f32 + conv1x1_stream_f32") + " (every conv launch of a step)",
                     "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "traffic": traffic, "avg_launch_us": round(conv_ms * 1e3 / max(launches, 1), 2),
                     "launches_per_step": launches,
                     "flop_per_launch_avg": conv_flops_frame * B / max(launches, 1)},
        "device_ms_per_step": {k: round(v, 3) for k, v in kinds.items()},
    }
    headline_default = (args.model == "yolov8n" and args.size == 640 and B == 512 and not args.half and not args.host_frames)
    if world == 1 and not args.no_cpu_baseline and not args.half:
        line["cpu_baseline"] = cpu_baseline(args, sd, frames_np, model)
    if world == 1 and headline_default and not args.no_configs:
        # the headline workload fed from PINNED host memory, 128-frame chunks (measured best of 32 / 64 / 128 / 256: 8,455 / 9,043 /
        # 9,266 / 8,906 frames/s): chunk k+1 crosses PCIe while chunk k computes
        del model
        torch.cuda.empty_cache()
        host = frames.cpu().pin_memory().numpy()
        m2 = YOLO(blob, device=local_rank, batch_chunk=128)
        for _ in range(2):
            m2._infer_rows(host, 0.25, 0.7, None, 300, args.size)
        t0 = time.perf_counter()
        HS = 5
        for _ in range(HS):
            m2._infer_rows(host, 0.25, 0.7, None, 300, args.size)
        torch.cuda.synchronize()
        line["host_fed_value"] = {"value": round(B * HS / (time.perf_counter() - t0), 1), "unit": "frames/s",
                                  "what": "same workload, frames in pinned host memory, engine chunk 128 (H2D of chunk k+1 overlaps chunk k)"}
        del m2, host, frames
        torch.cuda.empty_cache()
        line["configs"] = [measure_config(*c) for c in EXTRA_CONFIGS]
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(args, sd, frames_np, model):
    """The torch-CPU oracle (a restatement of the Ultralytics CPU path: kind 'port') timed on the host cores on a
    bounded sample of the same workload; the same frames go through the GPU path for a parity figure, and both are
    measured against a float64 execution of the same program (tools/precision.py)."""
    import torch
    from oracle import yolo_oracle as O
    om = O.OracleModel(args.model, sd)
    n = min(args.cpu_frames, len(frames_np))
    sample = list(frames_np[:n])
    O.predict(om, sample[:2], imgsz=args.size)                     # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        want, _ = O.predict(om, sample, imgsz=args.size)
        reps += 1
        if time.perf_counter() - t0 > 10.0 or reps >= 5:
            break
    dt = time.perf_counter() - t0
    got = model.predict(frames_np[:n], imgsz=args.size)
    same = sum(int(np.array_equal(g.anchor_idx, w["anchor_idx"].numpy())) for g, w in zip(got, want))
    err = 0.0
    for g, w in zip(got, want):
        if np.array_equal(g.anchor_idx, w["anchor_idx"].numpy()) and len(g.anchor_idx):
            err = max(err, float(np.abs(g.boxes.data.numpy()[:, :4] - w["boxes"].numpy()[:, :4]).max()))
    # the canonical-operation-order C oracle on 2 of the frames: the GPU rows must be identical bit for bit
    from oracle import det
    dwant, _ = det.predict(det.DetOracleModel(args.model, sd), sample[:2], imgsz=args.size)
    bit_exact = all(np.array_equal(g.anchor_idx, w["anchor_idx"].numpy()) and
                    np.array_equal(g.boxes.data.numpy(), w["boxes"].numpy()) for g, w in zip(got[:2], dwant))
    parity = {"frames_with_identical_indices": same, "frames": n, "max_box_abs_err_px": err,
              "rows_bit_exact_vs_canonical_order_oracle": bool(bit_exact)}
    # both fp32 implementations against float64 on 2 of the frames (pre-NMS head tensor, box channels, pixels)
    try:
        from tools import precision as P
        nf = min(2, n)
        ref = P.f64_head(args.model, sd, frames_np[:nf], args.size)
        e_gpu = P.group_errors(model.raw_head(frames_np[:nf], imgsz=args.size), ref, model.nc)["box"]
        e_cpu = P.group_errors(om.forward(O.preprocess(sample[:nf], args.size)).numpy(), ref, model.nc)["box"]
        parity["err_vs_f64_px"] = {"gpu": {k: float(f"{v:.3e}") for k, v in e_gpu.items()},
                                   "torch_cpu": {k: float(f"{v:.3e}") for k, v in e_cpu.items()}, "frames": nf,
                                   "what": "pre-NMS box channels vs a float64 execution of the same fused program"}
    except Exception as e:      # the yardstick is a report, not a gate of the benchmark
        parity["err_vs_f64_px"] = {"error": repr(e)}
    return {"value": round(n * reps / dt, 2), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} of the benchmark frames x {reps} passes through oracle/yolo_oracle.py (torch {torch.__version__} "
                      f"CPU fp32, batch {n})",
            "parity_vs_gpu": parity}


if __name__ == "__main__":
    main()
