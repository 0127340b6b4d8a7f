import os
import re
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """the C-ABI library loads (no GPU needed for that) and exports each function include/mi355_yolo.h declares"""
    import ctypes
    from cvsd_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mi355_yolo.h")).read()
    declared = set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    handle = _lib.lib()
    for name in declared:
        assert getattr(handle, name) is not None
    assert ctypes.sizeof(_lib.Det) == 4 * 58


def test_no_gpu_is_a_loud_error_not_a_fallback(v8n):
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from cvsd_amd import YOLO, ops
    from cvsd_amd._lib import Mi355Error
    with pytest.raises((Mi355Error, ValueError)):
        YOLO.from_state_dict("yolov8n", v8n[1])
    with pytest.raises((Mi355Error, ValueError)):
        ops.conv2d(np.zeros((1, 4, 4, 16), np.float32), np.zeros((16, 16, 1, 1), np.float32), np.zeros(16, np.float32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "computer-vision-shoplifting-detection_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tools)\b", src, re.M), f


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from cvsd_amd import dist as cd
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        blob = bytes(range(256)) * 37 if rank == 0 else None
        got = cd.broadcast_weights(blob)                                   # C1
        assert got == bytes(range(256)) * 37
        lo, hi = cd.shard_range(10, rank, world)
        rows = np.zeros((hi - lo, 4, 58), np.float32)
        counts = np.array([(lo + i) % 4 for i in range(hi - lo)], np.int32)
        for i in range(hi - lo):
            rows[i, :counts[i], 0] = 100 * (lo + i) + np.arange(counts[i])
        blocks, all_counts = cd.gather_rows(rows, counts)                  # C2 + C3
        if rank == 0:
            flat = np.concatenate(blocks)[:, 0]
            want = np.concatenate([100 * f + np.arange(f % 4) for f in range(10)])
            assert np.array_equal(flat, want) and np.concatenate(all_counts).tolist() == [f % 4 for f in range(10)]
            out.put("ok")
        else:
            assert blocks is None
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_gather():
    """N>1 path on CPU: weight broadcast and ordered row gather with world_size 2 (gloo)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) == "ok"


def test_shard_range_covers_everything():
    from cvsd_amd.dist import shard_range
    for total in (0, 1, 7, 64, 145):
        for world in (1, 2, 4, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total and all(a[1] == b[0] for a, b in zip(r, r[1:]))


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """include/mi355_yolo.h must be consumable from C (the drop-in boundary is a C ABI, not a C++ one): a C99 client that
    references every entry point compiles with gcc and links against libmi355yolo.so."""
    import re
    import shutil
    import subprocess
    from cvsd_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "mi355_yolo.h")
    names = sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", open(hdr).read())))
    assert "mi355_yolo_infer" in names and "mi355_yolo_create" in names
    src = tmp_path / "client.c"
    body = "\n".join(f"    p[{i}] = (void*)&{n};" for i, n in enumerate(names))
    src.write_text('#include "mi355_yolo.h"\n#include <stdio.h>\nint main(void) {\n    mi355_opts o = {0}; mi355_det d; void* p[%d];\n'
                   '    o.struct_size = (int)sizeof o; (void)d;\n%s\n    printf("%%d %%d\\n", (int)sizeof(mi355_det), o.struct_size);\n    return p[0] == 0;\n}\n'
                   % (len(names), body))
    so = _lib.LIB_PATH
    exe = tmp_path / "client"
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), so,
                        "-Wl,-rpath," + os.path.dirname(so)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_bench_self_launch_without_gpus_returns_nonzero_and_no_result_line():
    """`python bench.py --gpus 2` starts its own ranks (no torch.distributed.run needed); on a box without GPUs every rank
    refuses to run (there is no CPU fallback), and the launcher reports that as a non-zero exit without a JSON line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "BENCH_DRYRUN_ONE_GPU")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the failure path is covered by tests/test_gpu_bench.py")
    assert p.returncode != 0
    assert "needs an MI355X" in p.stderr or "GPU" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
