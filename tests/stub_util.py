"""Builds the test doubles under tests/stubs/ (in-tree, so the built files travel to the GPU box)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
STUBS = os.path.join(HERE, "stubs")
OUT = os.path.join(STUBS, "_build")


def rccl_stub_path(force=False):
    """librccl test double (tests/stubs/rccl_stub.cpp): the path to hand to OCRS_RCCL_LIB."""
    src = os.path.join(STUBS, "rccl_stub.cpp")
    lib = os.path.join(OUT, "librccl_stub.so")
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        os.makedirs(OUT, exist_ok=True)
        hipcc = os.environ.get("HIPCC") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc")
        cmd = [hipcc, "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", src, "-o", lib]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building the RCCL test double failed: %s\n%s" % (" ".join(cmd), r.stderr[-3000:]))
    return lib
