"""The C-ABI shared library loads without a GPU and exports every symbol the headers declare.
No compute calls here (those are the -m gpu tests); only host-side entry points are exercised."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")


def declared_symbols():
    names = set()
    for fn in sorted(os.listdir(INCLUDE)):
        text = open(os.path.join(INCLUDE, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(ultra_[a-z0-9_]+)\s*\(", text))
        for s, m in re.findall(r"ULTRA_DECLARE_REFERENCE_ENTRY\((\w+),\s*(\w+)\)", text):
            if s != "SUM":
                names.add("ultra_rspmm_%s_%s_forward_cuda" % (s, m))
                names.add("ultra_rspmm_%s_%s_backward_cuda" % (s, m))
    names.discard("ultra_rspmm_")
    return sorted(n for n in names if not n.endswith("_"))


def test_library_exports_every_declared_symbol():
    from ultra_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 28, names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing
    for s in ("add", "min", "max"):
        for m in ("mul", "add"):
            for d in ("forward", "backward"):
                assert "ultra_rspmm_%s_%s_%s_cuda" % (s, m, d) in names      # rspmm.h:63-105, one per reference export


def test_host_entry_points_without_gpu():
    from ultra_amd import _lib
    lib = _lib.lib
    assert lib.ultra_abi_version() == 7
    assert lib.ultra_device_count() >= 0
    t = _lib.Tuning()
    assert lib.ultra_get_tuning(ctypes.byref(t)) == 0
    bad = _lib.Tuning(100, 0, -1, -1, 0, (ctypes.c_int32 * 3)(0, 0, 0))      # threads not a multiple of 64
    assert lib.ultra_set_tuning(ctypes.byref(bad)) == _lib.ULTRA_ERR_INVALID
    assert b"multiple of 64" in lib.ultra_last_error()
    assert lib.ultra_set_tuning(None) == 0
    # plan creation is pure host code; error paths report through ultra_last_error()
    h = ctypes.c_void_p()
    assert lib.ultra_plan_create(ctypes.byref(h), None, None, -1, 1, 1, 1, None) == _lib.ULTRA_ERR_INVALID
    assert lib.ultra_plan_create(ctypes.byref(h), None, None, 0, 3, 3, 1, None) == 0
    info = _lib.PlanInfo()
    assert lib.ultra_plan_get_info(h, ctypes.byref(info)) == 0
    assert info.num_node == 3 and info.n_item == 3 and info.on_device == 0
    assert lib.ultra_plan_destroy(h) == 0


def test_dense_entry_points_validate_shapes():
    from ultra_amd import _lib
    lib = _lib.lib
    rc = lib.ultra_conv_update(None, None, None, None, None, None, None, 10, 32, 64, 1e-5, 0, None)
    assert rc == _lib.ULTRA_ERR_UNSUPPORTED
    rc = lib.ultra_readout(None, None, None, None, None, None, None, None, 0, None, 1, 10, 10, 32, 64, None)
    assert rc == _lib.ULTRA_ERR_UNSUPPORTED
    rc = lib.ultra_relation_projection(None, None, None, None, None, None, 10, 6, 32, None)
    assert rc == _lib.ULTRA_ERR_UNSUPPORTED
    rc = lib.ultra_query_boundary(None, None, None, None, None, 2, 10, 4, 64, None, None, None, None)
    assert rc == _lib.ULTRA_ERR_INVALID


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """No silent fallback: without the .so the binding module refuses to import."""
    from ultra_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libultra_amd.so"))
    with pytest.raises(ImportError, match="no CPU or PyTorch fallback"):
        _lib._load()


def test_torch_binding_has_the_reference_extension_surface():
    """The pybind11 module `rspmm` (csrc/torch_binding/rspmm.cpp, a shim over the C ABI) exports the names of
    /root/reference/ultra/rspmm/source/rspmm.cpp:256-283 with the same Tensor signatures -- checked against the
    reference's own compiled module (oracle/_ref) where that is present -- so the unchanged wrapper
    ultra/rspmm/rspmm.py can load it.  No GPU needed: the engine library is opened lazily, CPU calls raise."""
    import torch
    from ultra_amd import build
    mod = build.load_torch_binding()
    sig = {}
    for s in ("add", "min", "max"):
        for m in ("mul", "add"):
            for d, n_arg, ret in (("forward", 5, "torch.Tensor"),
                                  ("backward", 7, "tuple[torch.Tensor, torch.Tensor, torch.Tensor]")):
                for dev in ("cpu", "cuda"):
                    name = "rspmm_%s_%s_%s_%s" % (s, m, d, dev)
                    fn = getattr(mod, name)
                    doc = fn.__doc__.splitlines()[0]
                    assert doc.count("torch.Tensor") >= n_arg, doc
                    assert doc.split("->")[1].strip().replace("Tuple", "tuple") == ret, doc
                    sig[name] = doc.split("(", 1)[1]
    from oracle import build_ref
    if build_ref.available():
        ref = build_ref.load()
        ref_names = sorted(n for n in dir(ref) if n.startswith("rspmm_"))
        assert ref_names == sorted(n for n in sig if n.endswith("_cpu"))          # built without CUDA_OP: the 12 cpu exports
        for n in ref_names:
            assert getattr(ref, n).__doc__.splitlines()[0].split("(", 1)[1].replace("Tuple", "tuple") == sig[n].replace("Tuple", "tuple")
            assert sig[n] == sig[n.replace("_cpu", "_cuda")]
    x = torch.zeros(3, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        mod.rspmm_add_mul_forward_cpu(torch.zeros(2, 0, dtype=torch.long), torch.zeros(0, dtype=torch.long), torch.zeros(0), x, x)
    with pytest.raises(RuntimeError, match="same GPU"):      # reference: checkAllSameGPU (rspmm.cu:225)
        mod.rspmm_add_mul_forward_cuda(torch.zeros(2, 0, dtype=torch.long), torch.zeros(0, dtype=torch.long), torch.zeros(0), x, x)


def test_bench_names_the_timed_kernel_with_all_its_template_arguments():
    """bench.py matches rocprofv3 dispatches against ORDER_KERNEL by name: a template parameter added to the kernel without
    the constant following it silently drops the counter passes (the roofline then falls back to the byte model)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "ultra_amd", "csrc", "rspmm_order_kernels.hpp")).read()
    m = re.search(r"template <([^>]*)>\s*__global__ void __launch_bounds__\(ORDER_THREADS\)(?: ULTRA_ORDER_VGPR_CAP)? rspmm_order_kernel", src)
    assert m, "kernel declaration not found"
    n_params = len(m.group(1).split(","))
    bench = open(os.path.join(root, "bench.py")).read()
    name = re.search(r'^ORDER_KERNEL = "([^"]*)"', bench, re.M).group(1)
    assert name.startswith("rspmm_order_kernel<") and name.endswith(">")
    assert len(name[len("rspmm_order_kernel<"):-1].split(",")) == n_params
