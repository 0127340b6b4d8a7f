"""Build libmi355yolo.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m cvsd_amd.build          # or: from cvsd_amd.build import build; build()
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmi355yolo.so")
SOURCES = ["conv_f32_k3s1.hip", "conv_f32_k3s2.hip", "conv_f32_k1.hip", "conv_f32_pipe.hip", "conv_f32_splitk.hip", "conv_f32_fused_s1.hip", "conv_f32_fused_s2.hip", "conv_f32_group.hip", "conv_igemm_f16.hip", "conv_f16_fused.hip", "conv_f16_small.hip", "conv_f16_lw.hip", "conv_plan.hip",
           "misc_kernels.hip", "post_kernels.hip", "engine_load.hip", "engine_memory.hip", "engine_plans.hip", "engine_run.hip", "engine_abi.hip", "engine_ops.hip", "gmc_kernels.hip", "gmc_host.cpp", "tracker_host.cpp"]
HEADERS = ["common.h", "detmath.h", "conv_f32.h", "conv_f32_inst.h", "conv_f16.h", "engine_internal.h", os.path.join("..", "..", "include", "mi355_yolo.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit and link the shared library. Returns its path."""
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            flags = [f for f in FLAGS if not f.startswith("--offload-arch")] if s.endswith(".cpp") else FLAGS    # host-only C++
            jobs.append([_hipcc(), *flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
