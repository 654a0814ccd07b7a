"""ORACLE (test infrastructure only): compile the REFERENCE's own CPU translation unit.

Recipe: g++ directly on /root/reference/ultra/rspmm/source/rspmm.cpp where it lies (no copy
of reference sources enters this repo), with the reference's own flags
(/root/reference/ultra/rspmm/rspmm.py:184-189: -Ofast -fopenmp -DAT_PARALLEL_OPENMP) and
WITHOUT -DCUDA_OP (the reference .cu does not build against torch 2.10: THC/THCAtomics.cuh is gone).

Output: oracle/_ref/rspmm_ref_cpu.so -- a pybind11 module exporting the 12
rspmm_<sum>_<mul>_{forward,backward}_cpu functions (rspmm.cpp:256-269).  The directory is
git-ignored but NOT gpurun-ignored, so the prebuilt .so travels to the GPU box, where
/root/reference does not exist.  build() is a no-op when the reference is absent.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/ultra/rspmm/source/rspmm.cpp"
OUT_DIR = os.path.join(HERE, "_ref")
NAME = "rspmm_ref_cpu"
OUT = os.path.join(OUT_DIR, NAME + ".so")


def available():
    return os.path.exists(OUT)


def build(force=False):
    if not os.path.exists(REF_SRC):
        return OUT if os.path.exists(OUT) else None
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(REF_SRC):
        return OUT
    import torch
    from torch.utils import cpp_extension as ce

    inc = []
    for p in ce.include_paths():
        inc += ["-isystem", p]
    inc += ["-isystem", sysconfig.get_paths()["include"]]
    libdir = ce.library_paths()[0]
    cmd = ["g++", "-std=c++17", "-shared", "-fPIC",
           "-Ofast", "-fopenmp", "-DAT_PARALLEL_OPENMP",
           "-DTORCH_EXTENSION_NAME=" + NAME, "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-w"] + inc + [REF_SRC, "-o", OUT,
           "-L" + libdir, "-Wl,-rpath," + libdir,
           "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python"]
    subprocess.check_call(cmd)
    return OUT


def load():
    """Import the prebuilt reference module (requires torch to be imported first)."""
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded before the extension)

    if not os.path.exists(OUT):
        raise FileNotFoundError(OUT + " not built (run oracle/build_ref.py where /root/reference exists)")
    spec = importlib.util.spec_from_file_location(NAME, OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
