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
    assert c["parity_vs_gpu"]["rows_bit_exact_vs_canonical_order_oracle"] is True


def test_bench_two_ranks_rehearsed_on_one_gpu():
    env = dict(os.environ, BENCH_DRYRUN_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--chunk", "8",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" not in d
    assert d["config"]["global_batch"] == 16
