"""bench.py's contract on the GPU box: the JSON line at N = 1 and the N > 1 code path (two ranks rehearsed on one GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]                      # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_line_single_gpu():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "8", "--chunk", "8", "--steps", "2", "--warmup", "1",
                        "--cpu-frames", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f32" and d["vs_baseline"] is None and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    # warm-ups, median of several iterations, end-to-end and net-only, host description
    assert c["warmups"] >= 1 and c["iterations"] >= 3 and c["net_only"]["frames_per_s"] >= c["end_to_end"]["frames_per_s"] > 0
    assert c["nproc"] >= 1 and c["torch"] and c["batch"] == 2
    par = c["parity_vs_gpu"]
    assert par["rows_bit_exact_vs_canonical_order_oracle"] is True
    assert par["frames"] == 2 and 0 <= par["frames_with_identical_indices"] <= 2 and par["class_indices_identical"] is True
    if par["rows_compared"]:
        assert 0.0 <= par["post_nms_box_abs_err_px"]["frac_within_1e-3"] <= 1.0 and par["post_nms_box_abs_err_px"]["max"] < 5e-2
    e = par["kept_anchor_box_abs_err_px"]                       # both fp32 implementations against the float64 yardstick
    assert e["gpu_vs_f64"]["mean"] > 0 and e["torch_vs_f64"]["mean"] > 0 and e["gpu_vs_f64"]["mean"] <= 1.25 * e["torch_vs_f64"]["mean"]
    r = d["roofline"]
    assert len(r["plan_hash"]) == 16 and r["plan_source"] in ("tuned", "file", "cache", "memory") and "traffic_source" in r
    assert r["traffic"] is None                                 # a PMC figure is only quoted for the exact workload it was collected on
    assert "configs" not in d and "host_fed_value" not in d   # those ride on the default headline workload only


def test_bench_two_ranks_rehearsed_on_one_gpu():
    env = dict(os.environ, BENCH_DRYRUN_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--chunk", "8",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" not in d
    assert d["config"]["global_batch"] == 16


def test_bench_gpus_2_as_one_plain_command_self_launches_its_ranks():
    """`python bench.py --gpus 2` (the shape of the driver's N = 1 command): the script starts its own per-rank children before
    touching the GPU, relays rank 0's ONE JSON line and exits 0; rehearsed on one GPU (all ranks on cuda:0, gloo)."""
    env = dict(os.environ, BENCH_DRYRUN_ONE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--chunk", "8", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" not in d
    c = d["config"]
    assert c["global_batch"] == 16 and c["ranks"] == 2 and c["collective_world_size"] == 2 and c["collectives_backend"] == "gloo"


def test_bench_self_launch_fails_loudly_when_a_rank_fails():
    """more ranks than GPUs (and no dry-run switch): the ranks without a GPU exit non-zero, the launcher stops the others, prints
    no result line and returns non-zero"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "BENCH_DRYRUN_ONE_GPU")}
    # 3 ranks on a 1-GPU box: ranks 1 and 2 find no GPU of their own (kept small: a box admits few processes on its GPU at once)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--batch", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_async_device_rows_equal_the_blocking_call(v8n):
    """mi355_yolo_infer_device_async: packed rows / counts / total left in HBM == what the blocking entry point returns"""
    import numpy as np
    import torch
    from cvsd_amd import YOLO, _lib
    from tools import synth
    m = YOLO.from_state_dict("yolov8n", v8n[1], batch_chunk=4)
    frames = torch.from_numpy(synth.synthetic_frames(6, 320, 320, seed=9)).cuda()
    rows, counts, _ = m._infer_rows(frames, 0.25, 0.7, None, 300, 320)
    out = m.new_device_rows(6)
    m.infer_async(frames, out, conf=0.25, iou=0.7, imgsz=320)
    m.infer_async(frames, out, conf=0.25, iou=0.7, imgsz=320)          # a second call waits for the first before reusing scratch
    torch.cuda.current_stream().wait_stream(m.stream)
    torch.cuda.synchronize()
    d_rows, d_counts, d_total = (t.cpu().numpy() for t in out)
    np.testing.assert_array_equal(d_counts, counts)
    assert int(d_total[0]) == int(counts.sum()) > 0
    want = np.concatenate([rows[i, :counts[i]] for i in range(6)])
    np.testing.assert_array_equal(d_rows[:len(want)].view(np.uint32), want.view(np.uint32))


def test_rccl_world_of_one_with_the_engine_stream():
    """RCCL (backend 'nccl') initialises on this box and its collectives consume buffers produced on the engine's own
    non-blocking stream (event-ordered): the plumbing the 8-GPU run relies on, exercised with the one rank a 1-GPU box allows."""
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from cvsd_amd import YOLO
from tools import synth
_, sd = synth.synthetic_checkpoint("yolov8n", seed=0)
m = YOLO.from_state_dict("yolov8n", sd, batch_chunk=4)
frames = torch.from_numpy(synth.synthetic_frames(4, 320, 320, seed=9)).cuda()
out = m.new_device_rows(4)
m.infer_async(frames, out, imgsz=320)
ev = torch.cuda.Event(); ev.record(m.stream)
torch.cuda.current_stream().wait_event(ev)
rows, counts, total = out
got = [torch.empty_like(counts)]
dist.all_gather(got, counts)
blk = rows[:max(int(counts.sum().item()), 1)].contiguous()
recv = [torch.empty_like(blk)]
dist.gather(blk, recv, dst=0)
w = torch.full((1 << 20,), 7, dtype=torch.uint8, device="cuda"); dist.broadcast(w, 0)
torch.cuda.synchronize()
ref_rows, ref_counts, _ = m._infer_rows(frames, 0.25, 0.7, None, 300, 320)
assert np.array_equal(got[0].cpu().numpy(), ref_counts) and int(total.item()) == int(ref_counts.sum())
assert np.array_equal(recv[0].cpu().numpy().view(np.uint32)[:int(total.item())], np.concatenate([ref_rows[i, :ref_counts[i]] for i in range(4)]).view(np.uint32))
dist.destroy_process_group()
print("RCCL_OK", dist.is_nccl_available())
'''
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
