"""ORACLE (test infrastructure only): compile oracle/rspmm_oracle.c and oracle/torch_math_oracle.c with gcc.

Output: oracle/_build/librspmm_oracle.so, oracle/_build/libtorch_math_oracle.so (git-ignored, travel to the GPU box).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rspmm_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "librspmm_oracle.so")


MATH_SRC = os.path.join(HERE, "torch_math_oracle.c")
MATH_OUT = os.path.join(OUT_DIR, "libtorch_math_oracle.so")


def _compile(src, out, force, extra=()):
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = ["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-std=c11", *extra,
           "-shared", "-fPIC", src, "-o", out, "-lm"]
    subprocess.check_call(cmd)
    return out


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    # (-mfma: fmaf() must be the hardware fused multiply-add, not a soft-float call; the x86-64 hosts here have it)
    _compile(MATH_SRC, MATH_OUT, force, extra=("-mfma",))
    return _compile(SRC, OUT, force)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
