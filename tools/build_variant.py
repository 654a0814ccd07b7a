"""Build a variant of libultra_amd.so with extra -D flags for kernel A/B measurements on one box:

    python tools/build_variant.py NAME -DULTRA_ASM_WALK=0 [...]      ->  ultra_amd/lib/variants/libultra_amd_NAME.so
    ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_NAME.so python tools/...

Only the translation units that see such flags (the reference-order kernels, the dense epilogues) are recompiled; the other objects of the
default build are linked as they are."""
import glob
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import build as B  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    B.build()
    vdir = os.path.join(B.LIB_DIR, "variants")
    odir = os.path.join(vdir, "obj_" + name)
    os.makedirs(odir, exist_ok=True)
    objs = []
    procs = []
    for src in B._sources():
        base = os.path.basename(src)
        if base.startswith("rspmm_order_") or base in ("rspmm_api.hip", "plan.cpp", "dense_kernels.hip", "dense_order_layer.hip"):
            obj = os.path.join(odir, base + ".o")
            cmd = [B.HIPCC] + B.CFLAGS + flags + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            procs.append((subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), src))
        else:
            obj = os.path.join(B.OBJ_DIR, base + ".o")
        objs.append(obj)
    for pr, src in procs:
        out, _ = pr.communicate()
        if pr.returncode:
            raise SystemExit("hipcc failed for %s:\n%s" % (src, out))
    lib = os.path.join(vdir, "libultra_amd_%s.so" % name)
    subprocess.run([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC"] + objs + ["-o", lib], check=True)
    print(lib)


if __name__ == "__main__":
    main()
