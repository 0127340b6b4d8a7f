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
    # (model, size, batch, half, steps, warmup, cpu_budget_s): cpu_budget_s = wall-clock budget of the torch-CPU reference
    # timing at THIS config's batch (0: none -- the batch-64 form of config 4 is 8x the batch-8 entry's CPU work)
    ("yolov8n", 640, 1, False, 600, 100, 6.0),
    ("yolov8n-pose", 640, 32, False, 40, 5, 20.0),
    ("yolov8s-pose", 640, 8, False, 60, 8, 14.0),
    ("yolov8s-pose", 640, 64, False, 12, 3, 0.0),
    ("yolov8m", 1280, 2, True, 40, 5, 14.0),
    ("yolov8m", 1280, 16, True, 10, 3, 0.0),
    # the reference's literal checkpoint family and call: YOLO("./models/yolov5mu.pt") + .track() = batch 1 (/root/reference/model.py:18,38)
    ("yolov5mu", 640, 1, False, 300, 50, 8.0),
    # the headline workload with mi355_opts.fast_act = 1 (opt-in tolerance mode: SiLU on v_exp_f32 / v_rcp_f32 instead of the canonical,
    # bit-reproducible form); never the headline value -- its own entry with its own parity block
    ("yolov8n", 640, 512, False, 12, 3, 0.0, True),
]
HEADLINE_CPU_BUDGET_S = 18.0


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


def conv_bytes_per_frame(model_name: str, size: int, elem_bytes: int) -> float:
    """SURVEY 8(d)'s layerwise bytes of the same convs conv_flops_per_frame counts: every conv reads its input once and writes its
    output once (the compulsory HBM traffic when each layer round-trips; fused launches move less), activations only."""
    from cvsd_amd.graph import build_program, parse_model_name
    pg = build_program(*parse_model_name(model_name))
    px = lambda sd: (640 // sd) ** 2
    return float(sum(elem_bytes * (c.cin * px(c.stride_div) * c.s * c.s + c.cout * px(c.stride_div)) for c in pg.convs if c.cin != 3)) * (size / 640.0) ** 2


def weight_bytes(model_name: str, elem_bytes: int) -> float:
    from cvsd_amd.graph import build_program, parse_model_name
    pg = build_program(*parse_model_name(model_name))
    return float(sum(elem_bytes * c.cout * c.cin * c.k * c.k for c in pg.convs if c.cin != 3))


HBM_PEAK_TBS = 8.0              # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def roofline_fractions(model_name: str, size: int, batch: int, half: bool, conv_ms: float) -> dict:
    """BOTH fractions of the conv stack -- algorithmic FLOPs over the matrix peak of the arithmetic type, algorithmic
    (layerwise) bytes over the HBM peak -- from the same HIP-event conv time; `bound` names the larger one."""
    es = 2 if half else 4
    flops = conv_flops_per_frame(model_name, size) * batch
    nbytes = conv_bytes_per_frame(model_name, size, es) * batch + weight_bytes(model_name, es)
    t = conv_ms * 1e-3
    ff = flops / t / 1e12 / (F16_PEAK_TFLOPS if half else FP32_PEAK_TFLOPS)
    hf = nbytes / t / 1e12 / HBM_PEAK_TBS
    return {"flops_frac": round(ff, 4), "hbm_frac": round(hf, 4), "bound": "mfma" if ff >= hf else "hbm",
            "algorithmic_gflop_per_step": round(flops / 1e9, 2), "algorithmic_gb_per_step": round(nbytes / 1e9, 3),
            "algorithmic_gb_per_s": round(nbytes / t / 1e9, 1)}


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


def _delta_stats(d) -> dict:
    """|delta| samples -> the figures north_star's accuracy clause is about (1e-3 = its tolerance)."""
    d = np.asarray(d, dtype=np.float64).ravel()
    if d.size == 0:
        return {"n": 0}
    return {"n": int(d.size), "max": float(f"{d.max():.3e}"), "p999": float(f"{np.quantile(d, 0.999):.3e}"),
            "mean": float(f"{d.mean():.3e}"), "frac_within_1e-3": round(float((d <= 1e-3).mean()), 5)}


def parity_report(model, model_name: str, sd, frames_np, size: int, want=None, f64_frames: int = 2) -> dict:
    """GPU rows vs the torch-CPU oracle's rows on the same frames (post-NMS, original-image pixels): frames whose kept
    anchor lists are identical, |delta| of every box / keypoint coordinate on those frames, the fraction within
    north_star's 1e-3 -- and, side by side, BOTH fp32 implementations against a float64 execution of the same program at
    the anchors that survive NMS (what 1e-3 means for fp32 at stride 32 is only visible against that yardstick)."""
    from oracle import yolo_oracle as O
    from tools import precision as P
    om = None
    if want is None:
        om = O.OracleModel(model_name, sd)
        want, _ = O.predict(om, list(frames_np), imgsz=size)
    got = model.predict(frames_np, imgsz=size)
    same, rows, d_box, d_kpt, cls_equal = 0, 0, [], [], True
    for g, w in zip(got, want):
        wa = w["anchor_idx"].numpy()
        if not np.array_equal(g.anchor_idx, wa):
            continue
        same += 1
        rows += len(wa)
        if not len(wa):
            continue
        gb, wb = g.boxes.data.numpy(), w["boxes"].numpy()
        cls_equal &= bool(np.array_equal(gb[:, 5], wb[:, 5]))
        d_box.append(np.abs(gb[:, :4] - wb[:, :4]).ravel())
        if w["kpts"] is not None:
            d_kpt.append(np.abs(g.keypoints_raw[..., :2] - w["kpts"].numpy()[..., :2]).ravel())
    out = {"vs": "oracle/yolo_oracle.py (torch CPU fp32 restatement of the Ultralytics path; parity unpinned against Ultralytics itself)",
           "frames": len(got), "frames_with_identical_indices": same, "rows_compared": rows, "class_indices_identical": cls_equal,
           "post_nms_box_abs_err_px": _delta_stats(np.concatenate(d_box) if d_box else [])}
    if d_kpt:
        out["post_nms_kpt_abs_err_px"] = _delta_stats(np.concatenate(d_kpt))
    try:
        nf = min(f64_frames, len(frames_np))
        if nf > 0:
            om = om or O.OracleModel(model_name, sd)
            ref = P.f64_head(model_name, sd, frames_np[:nf], size)
            gh = model.raw_head(frames_np[:nf], imgsz=size)
            th = om.forward(O.preprocess(list(frames_np[:nf]), size)).numpy()
            sel = [(i, a) for i in range(nf) for a in got[i].anchor_idx.tolist()]
            if sel:
                ii, aa = np.array([s_[0] for s_ in sel]), np.array([s_[1] for s_ in sel])
                pick = lambda t: t[ii, :4, aa].astype(np.float64)
                out["kept_anchor_box_abs_err_px"] = {"gpu_vs_f64": _delta_stats(np.abs(pick(gh) - pick(ref))),
                                                     "torch_vs_f64": _delta_stats(np.abs(pick(th) - pick(ref))),
                                                     "gpu_vs_torch": _delta_stats(np.abs(pick(gh) - pick(th))),
                                                     "frames": nf, "what": "xywh of the anchors kept by NMS, pre-scale-back; float64 = the same fused program in double"}
    except Exception as e:          # the yardstick is a report, not a gate of the benchmark
        out["kept_anchor_box_abs_err_px"] = {"error": repr(e)}
    return out


def half_parity_report(model_name: str, sd, frames_np, size: int, half_model) -> dict:
    """half=True has no CPU reference run (Ultralytics refuses half on CPU): its post-NMS rows are measured against the fp32
    ENGINE's rows on the same frames, matched by source anchor."""
    import torch
    from cvsd_amd import YOLO
    from cvsd_amd.weights import build_from_state_dict
    m32 = YOLO(build_from_state_dict(model_name, sd), device=torch.cuda.current_device(), batch_chunk=len(frames_np))
    r32 = m32.predict(frames_np, imgsz=size)
    r16 = half_model.predict(frames_np, imgsz=size)
    matched = total = flips = 0
    d_box, d_score = [], []
    for a, b in zip(r32, r16):
        ia = {int(k): i for i, k in enumerate(a.anchor_idx)}
        ib = {int(k): i for i, k in enumerate(b.anchor_idx)}
        common = sorted(set(ia) & set(ib))
        total += len(set(ia) | set(ib))
        matched += len(common)
        for k in common:
            da, db = a.boxes.data.numpy()[ia[k]], b.boxes.data.numpy()[ib[k]]
            d_box.append(float(np.abs(da[:4] - db[:4]).max()))
            d_score.append(abs(float(da[4] - db[4])))
            flips += int(da[5] != db[5])
    del m32
    torch.cuda.empty_cache()
    return {"vs": "the fp32 engine on the same frames (no CPU reference exists for half=True), rows matched by source anchor",
            "frames": len(frames_np), "rows_matched": matched, "rows_total": total, "class_flips": flips,
            "row_box_abs_err_px": {"median": float(f"{np.median(d_box):.3e}") if d_box else None, "max": float(f"{max(d_box):.3e}") if d_box else None},
            "row_score_abs_err_max": float(f"{max(d_score):.3e}") if d_score else None}


def cpu_reference(model_name: str, sd, frames_np, size: int, batch: int, budget_s: float, fp32_note: str = "") -> tuple:
    """the torch-CPU oracle (restated Ultralytics CPU path, fp32, all host cores) on `batch` frames of
    this config: 3 warm-ups, then the MEDIAN of up to 20 timed iterations within a wall-clock budget; end-to-end (letterbox +
    /255 -> net -> NMS -> scale-back rows) and net-only (the forward pass alone) reported separately.
    -> (report dict, the oracle's rows of the last iteration for the parity figures)."""
    import platform
    import torch
    from oracle import yolo_oracle as O
    om = O.OracleModel(model_name, sd)
    reps = (batch + len(frames_np) - 1) // len(frames_np)
    sample = list(np.concatenate([frames_np] * reps)[:batch]) if reps > 1 else list(frames_np[:batch])

    def one():
        t0 = time.perf_counter()
        im = O.preprocess(sample, size)
        t1 = time.perf_counter()
        with torch.no_grad():
            pred = om.forward(im)
        t2 = time.perf_counter()
        rows, idxs = O.non_max_suppression(pred, 0.25, 0.7, max_det=300, nc=om.nc, return_idxs=True)
        res = []
        for r, ai, f in zip(rows, idxs, sample):
            r = r.clone()
            r[:, :4] = O.scale_boxes(im.shape[2:], r[:, :4], f.shape)
            k = O.scale_coords(im.shape[2:], r[:, 6:].view(len(r), *om.kpt_shape).clone(), f.shape) if om.pose else None
            res.append({"boxes": r[:, :6], "kpts": k, "anchor_idx": ai})
        t3 = time.perf_counter()
        return (t3 - t0, t2 - t1, t1 - t0, t3 - t2), res

    t_start = time.perf_counter()
    (first, want) = one()
    warm = 3 if first[0] * 6 < budget_s else 1          # a slow config keeps its budget for timed iterations
    for _ in range(warm - 1):
        one()
    times = []
    t_loop = time.perf_counter()
    while len(times) < 20 and (len(times) < 3 or time.perf_counter() - t_loop + (times[-1][0] if times else 0) < budget_s):
        t, want = one()
        times.append(t)
    tt = np.array(times)
    med = np.median(tt, axis=0)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    rep = {"value": round(batch / med[0], 2), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
           "end_to_end": {"frames_per_s": round(batch / med[0], 2), "median_ms": round(med[0] * 1e3, 2),
                          "min_ms": round(float(tt[:, 0].min()) * 1e3, 2), "max_ms": round(float(tt[:, 0].max()) * 1e3, 2)},
           "net_only": {"frames_per_s": round(batch / med[1], 2), "median_ms": round(med[1] * 1e3, 2)},
           "preprocess_median_ms": round(med[2] * 1e3, 2), "postprocess_median_ms": round(med[3] * 1e3, 2),
           "batch": batch, "warmups": warm, "iterations": len(times), "budget_s": budget_s,
           "nproc": os.cpu_count(), "cpu_model": cpu_model or platform.processor(), "torch": torch.__version__,
           "sample": f"{batch} frames per iteration ({len(frames_np)} distinct synthetic frames" + (", repeated" if reps > 1 else "") +
                     f") through oracle/yolo_oracle.py: torch {torch.__version__} CPU fp32, {torch.get_num_threads()} threads; "
                     f"{warm} warm-ups, median of {len(times)} iterations, {time.perf_counter() - t_start:.1f} s of CPU work" + fp32_note}
    return rep, want


def measure_config(model_name: str, size: int, batch: int, half: bool, steps: int, warmup: int, cpu_budget_s: float = 0.0, fast_act: bool = False) -> dict:
    """One BASELINE configuration on this GPU: frames resident in HBM, rows returned to the host every step."""
    import torch
    from cvsd_amd import YOLO
    from cvsd_amd.weights import build_from_state_dict
    from tools import synth
    _, sd = synth.synthetic_checkpoint(model_name, seed=0)
    model = YOLO(build_from_state_dict(model_name, sd), device=torch.cuda.current_device(), batch_chunk=batch, half=half, fast_act=fast_act)
    frames, frames_np = make_frames(batch, size, seed=2000 + batch)
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
    plan = model.plan_info()
    # HBM traffic per conv launch of BASELINE config 5 at batch 16: quoted from its PMC passes (profiles/) when -- and only when -- they ran the
    # same launch plans as this process (as the headline's roofline.traffic)
    traffic = traffic_source = None
    if half and model_name == "yolov8m" and size == 1280 and batch == 16 and not fast_act:
        tpath = os.path.join(ROOT, "profiles", "r04_cfg5_conv_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("plan_hash") == plan["plan_hash"]:
                traffic = tj["hbm_bytes_per_launch_avg"]
                traffic_source = f"profiles/r04_cfg5_conv_traffic.json (separate rocprofv3 --pmc passes of this workload, same plan_hash {plan['plan_hash']})"
            else:
                traffic_source = f"profiles/r04_cfg5_conv_traffic.json not quoted: collected on plan_hash {tj.get('plan_hash')}, this run has {plan['plan_hash']}"
    out = {"workload": f"{model_name} {size}x{size} batch {batch}" + (" half=True" if half else "") + (" fast_act=1 (tolerance mode, opt-in)" if fast_act else ""),
           "dtype": "f16" if half else "f32",
           "value": round(batch * steps / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "roofline": {"achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                        **roofline_fractions(model_name, size, batch, half, conv_ms),
                        "launches_per_step": launches, "plan_hash": plan["plan_hash"], "plan_source": plan["plan_source"],
                        **({"traffic": traffic, "traffic_source": traffic_source} if traffic_source else {})},
           "activation_bytes": plan["activation_bytes"]}
    if fast_act:
        # parity of the tolerance mode, measured here (no CPU timing belongs to it): rows against the torch-CPU oracle and both against
        # float64 on 4 of the benchmark's frames, and the head tensor against the CANONICAL engine's on the same frames
        n_par = min(4, len(frames_np))
        out["parity"] = parity_report(model, model_name, sd, frames_np[:n_par], size, f64_frames=2)
        canon = YOLO(build_from_state_dict(model_name, sd), device=torch.cuda.current_device(), batch_chunk=n_par)
        hf, hc = model.raw_head(frames_np[:n_par], imgsz=size), canon.raw_head(frames_np[:n_par], imgsz=size)
        nc = model.nc
        out["parity"]["vs_canonical_engine_head"] = {"box_abs_err_px": _delta_stats(np.abs(hf[:, :4] - hc[:, :4])),
                                                     "score_abs_err": _delta_stats(np.abs(hf[:, 4:4 + nc] - hc[:, 4:4 + nc])),
                                                     "identical_bits": bool(np.array_equal(hf, hc))}
        del canon
    del frames, model
    torch.cuda.empty_cache()
    return out


def config_cpu_and_parity(out: dict, model_name: str, size: int, batch: int, half: bool, steps: int, warmup: int, cpu_budget_s: float = 0.0,
                          fast_act: bool = False) -> None:
    """Second phase, AFTER every GPU timing of the line: the torch-CPU reference at this config's batch and the parity figures of
    the same frames (the engine is re-created: its launch plans come from the plan file the first phase wrote).  Kept apart from
    the GPU timings because the CPU reference's worker threads keep spinning between calls and slow the host side of a
    latency-bound GPU loop down by 10-20 % (measured at batch 1)."""
    import torch
    from cvsd_amd import YOLO
    from cvsd_amd.weights import build_from_state_dict
    from tools import synth
    if cpu_budget_s <= 0:
        return
    _, sd = synth.synthetic_checkpoint(model_name, seed=0)
    model = YOLO(build_from_state_dict(model_name, sd), device=torch.cuda.current_device(), batch_chunk=batch, half=half)
    _, frames_np = make_frames(min(batch, N_BASE_FRAMES), size, seed=2000 + batch)
    n_par = min(batch, len(frames_np))
    if half:
        out["cpu_baseline"], _ = cpu_reference(model_name, sd, frames_np, size, batch, cpu_budget_s,
                                               fp32_note="; fp32 on the CPU (Ultralytics refuses half=True on CPU: there is no fp16 CPU path to time)")
        out["parity"] = half_parity_report(model_name, sd, frames_np[:n_par], size, model)
    else:
        out["cpu_baseline"], want = cpu_reference(model_name, sd, frames_np, size, batch, cpu_budget_s)
        out["parity"] = parity_report(model, model_name, sd, frames_np[:n_par], size, want=want[:n_par],
                                      f64_frames=1 if conv_flops_per_frame(model_name, size) > 2e10 else 2)
    del model
    torch.cuda.empty_cache()


def launch_ranks(n: int, argv: list) -> int:
    """``python bench.py --gpus N`` without a launcher: start N FRESH child processes of this script, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relay rank 0's single JSON line and return the first non-zero child exit
    code (0 when all ranks succeeded).  Runs before anything in this process has touched the GPU (torch is not even
    imported here): a process that has initialised HIP must never be re-executed on this pool, and it is not -- the parent
    only waits.  A rank that fails takes the others down with it (exact PIDs), so a broken rank cannot hang the job in a
    collective."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this host driver (RCCL needs it)
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout carries the JSON line; the other ranks' stdout goes to stderr so that exactly one line is relayed
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, cwd=os.getcwd(),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for o in live:
                    procs[o].terminate()
        time.sleep(0.05)
    reader.join(timeout=10)
    for line in out0:
        if rc == 0 or not line.startswith("{"):               # a failed job prints no result line
            sys.stdout.write(line)
    sys.stdout.flush()
    if rc == 0 and sum(l.startswith("{") for l in out0) != 1:
        print("bench.py: rank 0 did not print exactly one JSON line", file=sys.stderr)
        rc = 1
    return rc


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
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (no torch, no HIP call before or after)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dry = os.environ.get("BENCH_DRYRUN_ONE_GPU") == "1"
    # device_count() does not initialise the GPU: a rank without a GPU of its own leaves before it opens one
    if torch.cuda.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the hot path)")
    if not dry and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible (--gpus {args.gpus})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the hot path)")
    # BENCH_DRYRUN_ONE_GPU=1 (tests only): all ranks share cuda:0 and the collectives run over gloo, to rehearse the
    # N > 1 code path on a one-GPU box.  The real launch is one rank per GPU over RCCL (backend "nccl").
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
    # the world size the collective library itself reports (RCCL's communicator when backend == "nccl")
    comm_world = dist.get_world_size() if world > 1 else 1
    if world > 1:
        backend = dist.get_backend()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    conv_flops_frame = conv_flops_per_frame(args.model, args.size)
    scale = (args.size / 640.0) ** 2
    achieved = conv_flops_frame * B / (conv_ms * 1e-3) / 1e12
    fps = world * B * args.steps / dt
    # HBM traffic per conv launch cannot be measured inside this process (rocprofv3 PMC passes are separate runs: the guide's
    # HBM section); the figure collected on this exact workload is kept under profiles/ and quoted ONLY with its source
    # named and only while the launch sequence it was collected on has the same length as this run's -- else null.
    traffic, traffic_source = None, None
    plan = model.plan_info()
    tag = "cfg5" if (args.half and args.model == "yolov8m" and args.size == 1280 and B == 16) else "v1" if (args.model == "yolov8n" and args.size == 640 and B == 512 and not args.half) else None
    tpath = os.path.join(ROOT, "profiles", f"r04_{tag}_conv_traffic.json") if tag and args.chunk == B else None
    if tpath and os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("plan_hash") == plan["plan_hash"]:
            traffic = tj["hbm_bytes_per_launch_avg"]
            traffic_source = (f"profiles/r04_{tag}_conv_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this command, FETCH doubled per "
                              f"the guide's gfx950 correction; same launch plans: plan_hash {plan['plan_hash']}; not measured in this process)")
        else:
            traffic_source = f"profiles/r04_{tag}_conv_traffic.json not quoted: collected on plan_hash {tj.get('plan_hash')}, this run has {plan['plan_hash']}"
    peak = F16_PEAK_TFLOPS if args.half else FP32_PEAK_TFLOPS
    line = {
        "metric": f"frames/s @{args.size}x{args.size}" + (" (half=True engine: diagnostic)" if args.half else "") + (" (HOST frames, PCIe-inclusive: diagnostic)" if args.host_frames else ""), "value": round(fps, 1), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if args.half else "f32", "data": "synthetic",
        "config": {"workload": f"{args.model} {args.size}x{args.size} synthetic BGR frames ({B} distinct frames per GPU), batch {B}/GPU/step, "
                               f"predict conf=0.25 iou=0.7 max_det=300 (letterbox+stem, conv graph, decode, NMS, rows to host)",
                   "global_batch": B * world, "params": params, "gflop_per_frame": round(gflops * scale, 3),
                   "parallelism": f"frame-sharded dp{world}", "collectives_backend": backend, "ranks": world,
                   "collective_world_size": comm_world,
                   "collectives_per_step": 0 if world == 1 else 2},
        "roofline": {"bound": "mfma", "kernel": ("conv_igemm_f16 + conv3x3_lw_f16 + conv1x1_lwx_f16 + conv1x1_pipe_f16" if args.half else "conv_igemm_f32 + conv1x1_pipe_f32 + conv1x1_stream_f32") + " (every conv launch of a step)",
                     "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "traffic": traffic, "traffic_source": traffic_source, "avg_launch_us": round(conv_ms * 1e3 / max(launches, 1), 2),
                     **{k: v for k, v in roofline_fractions(args.model, args.size, B, args.half, conv_ms).items() if k != "bound"},
                     "launches_per_step": launches,
                     "flop_per_launch_avg": conv_flops_frame * B / max(launches, 1)},
        "device_ms_per_step": {k: round(v, 3) for k, v in kinds.items()},
    }
    headline_default = (args.model == "yolov8n" and args.size == 640 and B == 512 and not args.half and not args.host_frames)
    line["roofline"]["plan_hash"], line["roofline"]["plan_source"] = plan["plan_hash"], plan["plan_source"]
    line["config"]["activation_bytes_per_gpu"] = plan["activation_bytes"]
    want_cpu = world == 1 and not args.no_cpu_baseline and not args.half
    if want_cpu and not (world == 1 and headline_default and not args.no_configs):
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
        HS, rates = 3, []
        for _ in range(3):                                   # median of three timed groups of three steps (the PCIe-fed rate varies +-5 % run to run)
            t0 = time.perf_counter()
            for _ in range(HS):
                m2._infer_rows(host, 0.25, 0.7, None, 300, args.size)
            torch.cuda.synchronize()
            rates.append(B * HS / (time.perf_counter() - t0))
        line["host_fed_value"] = {"value": round(sorted(rates)[1], 1), "unit": "frames/s", "min": round(min(rates), 1), "max": round(max(rates), 1),
                                  "what": "same workload, frames in pinned host memory, engine chunk 128 (H2D of chunk k+1 overlaps chunk k)"}
        del m2, host, frames
        torch.cuda.empty_cache()
        line["configs"] = [measure_config(*c) for c in EXTRA_CONFIGS]                      # every GPU timing first ...
        line["track_pipeline"] = track_pipeline()
        if want_cpu:                                                                        # ... then the CPU reference runs and the parity figures
            m3 = YOLO(blob, device=local_rank, batch_chunk=args.chunk)
            line["cpu_baseline"] = cpu_baseline(args, sd, frames_np, m3)
            del m3
            torch.cuda.empty_cache()
            for out, c in zip(line["configs"], EXTRA_CONFIGS):
                config_cpu_and_parity(out, *c)
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def track_pipeline(n_frames: int = 160) -> dict:
    """The reference's frame loop end to end (`model.track(frame, persist=True, ...)` per frame, /root/reference/model.py:38):
    detector at batch 1 + BoT-SORT with its sparse-optical-flow motion compensation, on a synthetic 320x240 clip under a panning
    camera (UCF-Crime's frame size).  Frames start on the host, as cv2.VideoCapture hands them over.  For YOLOv8n three tracker
    set-ups: the compensation's frame preparation and Lucas-Kanade step on the GPU (what model.track does), on the host
    (gmc_device=None), and switched off; for the reference's literal checkpoint family (YOLOv5mu, model.py:18) and for BASELINE config
    4's model (YOLOv8s-pose) the product set-up.  `sweep_batch64` = the production form of the same loop (cvsd_amd/sweep.py):
    detection batched per clip, the tracker frame by frame, frame j + 1's motion-compensation step enqueued while frame j is
    associated."""
    import numpy as np
    from cvsd_amd import YOLO
    from cvsd_amd.sweep import process_clip
    from cvsd_amd.tracker import BYTETracker
    from cvsd_amd.weights import build_from_state_dict
    from tools import synth
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, size=(256, 320 + 3 * n_frames + 16, 3), dtype=np.uint8)
    base = ((base.astype(np.uint16) + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) // 4).astype(np.uint8)
    frames = [np.ascontiguousarray(base[8:248, 3 * k:3 * k + 320]) for k in range(n_frames)]

    class Clip:
        def __init__(self, fr):
            self.fr, self.pos = fr, 0

        def read(self):
            if self.pos >= len(self.fr):
                return False, None
            self.pos += 1
            return True, self.fr[self.pos - 1]

        def get(self, prop):
            return float(self.pos)

        def release(self):
            pass

    out = {"workload": f"model.track loop, {n_frames - 8} frames of 320x240 (host frames), conf 0.1, BoT-SORT defaults (sparseOptFlow GMC, <= 1000 corners); "
                       f"tracker core in host C++ (csrc/tracker_host.cpp)", "unit": "frames/s"}
    for name in ("yolov8n", "yolov5mu", "yolov8s-pose"):
        _, sd = synth.synthetic_checkpoint(name, seed=0)
        blob = build_from_state_dict(name, sd)
        model = YOLO(blob, batch_chunk=1)
        setups = [("gmc_on_gpu", lambda: BYTETracker(gmc_device=model.device))]
        if name == "yolov8n":
            setups += [("gmc_on_host", lambda: BYTETracker(gmc_device=None)), ("gmc_off", lambda: BYTETracker(gmc_method=None))]
        res = {}
        for key, make in setups:
            model._tracker = make()
            for f in frames[:8]:
                model.track(f, persist=True, conf=0.1)
            t0 = time.perf_counter()
            for f in frames[8:]:
                model.track(f, persist=True, conf=0.1)
            res[key] = round((n_frames - 8) / (time.perf_counter() - t0), 1)
        del model
        big = YOLO(blob, batch_chunk=64)
        process_clip(big, Clip(frames[:64]), batch=64)
        long_clip = frames * 4                                     # 640 frames: several 64-frame batches in flight
        t0 = time.perf_counter()
        process_clip(big, Clip(long_clip), batch=64)
        res["sweep_batch64"] = round(len(long_clip) / (time.perf_counter() - t0), 1)
        del big
        if name == "yolov8n":
            out.update(res)
        else:
            out[name] = res
    return out


def cpu_baseline(args, sd, frames_np, model):
    """The headline's CPU baseline: the torch-CPU oracle (kind 'port': a restatement of the Ultralytics CPU path, which is
    not installable here) on a bounded sample of the benchmark's frames, timed as BASELINE.md section 4 prescribes; the same
    frames go through the GPU path for the parity figures (indices, post-NMS deltas, fraction within 1e-3, both fp32
    implementations against float64) and two of them through the canonical-order C oracle (bit-exact gate)."""
    n = min(args.cpu_frames, len(frames_np))
    rep, want = cpu_reference(args.model, sd, frames_np[:n], args.size, n, HEADLINE_CPU_BUDGET_S)
    parity = parity_report(model, args.model, sd, frames_np[:n], args.size, want=want)
    # the canonical-operation-order C oracle on 2 of the frames: the GPU rows must be identical bit for bit
    from oracle import det
    got = model.predict(frames_np[:2], imgsz=args.size)
    dwant, _ = det.predict(det.DetOracleModel(args.model, sd), list(frames_np[:2]), imgsz=args.size)
    parity["rows_bit_exact_vs_canonical_order_oracle"] = bool(all(
        np.array_equal(g.anchor_idx, w["anchor_idx"].numpy()) and np.array_equal(g.boxes.data.numpy(), w["boxes"].numpy())
        for g, w in zip(got, dwant)))
    rep["parity_vs_gpu"] = parity
    return rep


if __name__ == "__main__":
    main()
