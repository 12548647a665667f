"""Builds libocrs_amd.so (HIP kernels + host engine + C ABI) for gfx950, in-tree.

    python -m ocrs_amd.build            # incremental
    python -m ocrs_amd.build --force

hipcc cross-compiles without a GPU.  -ffp-contract=off is load-bearing: the
bit-exact stages spell every fused multiply-add as fmaf().
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libocrs_amd.so")

SOURCES = ["common.cpp", "layout.cpp", "model.cpp", "engine.cpp", "ctc_beam.cpp", "text_items.cpp", "abi.cpp", "abi_util.cpp", "group.cpp", "jpeg_host.cpp", "kernels_jpeg.hip", "kernels_image.hip", "kernels_ccl.hip",
           "kernels_nn.hip", "kernels_det.hip", "kernels_det_stream.hip", "kernels_det_rows.hip", "kernels_gru.hip", "kernels_gru_split.hip", "kernels_lines.hip", "kernels_beam.hip", "kernels_rec.hip", "kernels_peaks.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newest_header():
    t = 0.0
    for f in os.listdir(CSRC):
        if f.endswith(".hpp"):
            t = max(t, os.path.getmtime(os.path.join(CSRC, f)))
    t = max(t, os.path.getmtime(os.path.join(HERE, "..", "include", "ocrs_amd.h")))
    return t


def build_variant(tag, defines, sources, verbose=False):
    """An ablation build: `sources` recompiled with -D<defines>, linked with the stock objects of everything else into
    libocrs_amd.<tag>.so (loaded instead of the product when OCRS_AMD_LIB names it; tools/ab_*.sh)."""
    build()
    vdir = os.path.join(OBJ, tag)
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        op = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        if src in sources:
            op = os.path.join(vdir, src.rsplit(".", 1)[0] + ".o")
            cmd = [_hipcc()] + FLAGS + ["-D" + d for d in defines] + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(op)
    out = os.path.join(HERE, "libocrs_amd.%s.so" % tag)
    subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-lpthread"], check=True)
    return out


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = _newest_header()
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_t):
            cmd = [_hipcc()] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", op]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        if verbose and r.stderr.strip():
            print(r.stderr[-2000:])

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB):
        # librccl is NOT linked: the engine group's final result gather binds it at run time (group.cpp, dlopen), so
        # the library loads on hosts without RCCL and never mixes two RCCL builds in a process that imported torch
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"])
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python -m ocrs_amd.build --variant <tag> <file.hip>[,<file>] <DEFINE>[,<DEFINE>]
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 3].split(","), sys.argv[i + 2].split(","), verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
