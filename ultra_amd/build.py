"""Build libultra_amd.so for gfx950 with hipcc (no torch, no hipify, no cmake).

    python -m ultra_amd.build [--force]

Every translation unit under ultra_amd/csrc is compiled in parallel to ultra_amd/lib/obj/*.o and
linked into ultra_amd/lib/libultra_amd.so.  The .so is git-ignored but travels to the GPU box with
the gpurun snapshot.  hipcc cross-compiles without a GPU.
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB = os.path.join(LIB_DIR, "libultra_amd.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
          "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function", "-I" + INCLUDE]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _headers_mtime():
    hs = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, obj):
    cmd = [HIPCC] + CFLAGS + ["-c", src, "-o", obj]
    if src.endswith(".cpp"):
        cmd = [HIPCC] + CFLAGS + ["-x", "hip", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return r.stderr


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hm = _headers_mtime()
    jobs = []
    objs = []
    for src in _sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hm)
        if stale:
            jobs.append((src, obj))
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            futs = {ex.submit(_compile, s, o): s for s, o in jobs}
            for f in cf.as_completed(futs):
                warn = f.result()
                if verbose and warn.strip():
                    print(warn, file=sys.stderr)
    need_link = bool(jobs) or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)
    if need_link:
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


BINDING_SRC = os.path.join(CSRC, "torch_binding", "rspmm.cpp")
BINDING = os.path.join(LIB_DIR, "rspmm.so")


def build_torch_binding(force=False):
    """The pybind11 module `rspmm` (the reference extension's names and Tensor signatures, rspmm.cpp:256-283) as a shim
    over libultra_amd.so: g++ against torch's headers, no device code.  Output: ultra_amd/lib/rspmm.so."""
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce
    hdr = os.path.join(INCLUDE, "ultra_rspmm.h")
    if not force and os.path.exists(BINDING) and os.path.getmtime(BINDING) >= max(os.path.getmtime(BINDING_SRC), os.path.getmtime(hdr)):
        return BINDING
    os.makedirs(LIB_DIR, exist_ok=True)
    inc = []
    for p in ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include"]:
        inc += ["-isystem", p]
    libdir = ce.library_paths()[0]
    cmd = ["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=rspmm", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)] + inc + [
           BINDING_SRC, "-o", BINDING, "-L" + libdir, "-Wl,-rpath," + libdir,
           "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_python", "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("torch binding build failed:\n%s\n%s" % (r.stdout, r.stderr))
    return BINDING


def load_torch_binding():
    """import the built `rspmm` module (torch must be imported first)."""
    import importlib.util

    import torch  # noqa: F401
    spec = importlib.util.spec_from_file_location("rspmm", build_torch_binding())
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_torch_binding(force="--force" in sys.argv))
