"""Compile the deterministic C oracle (oracle/det_oracle.c) with gcc -> oracle/_build/libdetoracle.so.
TEST INFRASTRUCTURE ONLY.  -ffp-contract=off: every fused multiply-add in the source is an explicit fmaf()."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "det_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libdetoracle.so")


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        cmd = ["gcc", "-O3", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-mavx2", "-mfma", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", SRC, "-o", LIB, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"gcc failed: {' '.join(cmd)}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force=True))
