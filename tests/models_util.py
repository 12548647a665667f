"""Seeded synthetic model files shared by tests, smoke() and bench.py.

Real ocrs weights are not obtainable offline; the files made
here are the [UNVERIFIED-RECALL] architectures of SURVEY.md §2.4 with seeded
weights.  The recognition head is calibrated with the ORACLE's exact executor so
the file is identical on every machine.  Cached under $OCRS_AMD_CACHE (default
/tmp/ocrs_amd_cache)."""
import hashlib
import os

import numpy as np

from ocrs_amd import modelfile as mf
from ocrs_amd import synth

CACHE = os.environ.get("OCRS_AMD_CACHE", "/tmp/ocrs_amd_cache")


def _cached(name, make):
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, name)
    if os.path.exists(path):
        with open(path, "rb") as f:
            return f.read()
    buf = make()
    tmp = path + ".%d.tmp" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(buf)
    os.replace(tmp, path)
    return buf


def detection_model_bytes(in_hw=(800, 600), depths=(8, 16, 32, 32, 64, 128, 256), seed=1, ink=None):
    """ink = (ink_level, ink_gain, ink_sign) of modelfile.build_detection; None = the default file."""
    key = "det_%dx%d_%s_s%d_v1.ocrsm" % (in_hw[0], in_hw[1], "-".join(map(str, depths)), seed)
    if ink is None:
        return _cached(key, lambda: mf.build_detection(in_hw=in_hw, depths=depths, seed=seed).to_bytes())
    lvl, gain, sign = float(ink[0]), float(ink[1]), int(ink[2])
    key = key.replace("_v1.ocrsm", "_ink%g_%g_%d_v1.ocrsm" % (lvl, gain, sign))
    return _cached(key, lambda: mf.build_detection(in_hw=in_hw, depths=depths, seed=seed, ink_level=lvl, ink_gain=gain,
                                                   ink_sign=sign).to_bytes())


def recognition_model_bytes(seed=2, n_classes=97):
    def make():
        from oracle.nn import OracleGraph
        g = mf.build_recognition(n_classes=n_classes, seed=seed)
        cal = synth.synthetic_line_crops(7, n=8)
        xp = np.full((8, 1, 64, 300), -0.5, np.float32)
        xp[:, 0, :, :256] = cal
        g = mf.calibrate_recognition_head(g, lambda buf, x: OracleGraph(buf).run_exact(x), xp)
        return g.to_bytes()

    return _cached("rec_c%d_s%d_v1.ocrsm" % (n_classes, seed), make)


def digest(buf):
    return hashlib.sha256(buf).hexdigest()[:16]
