"""Seeded synthetic model files for bench.py (no network => no real ocrs
weights).  The recognition head is calibrated with the HIP
executor itself, so nothing here touches the oracle."""
import os

import numpy as np

from . import modelfile as mf
from . import synth

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


def synthetic_detection_bytes(in_hw=(800, 600), seed=1):
    return _cached("det_%dx%d_8-16-32-32-64-128-256_s%d_v1.ocrsm" % (in_hw[0], in_hw[1], seed),
                   lambda: mf.build_detection(in_hw=in_hw, seed=seed).to_bytes())


def synthetic_recognition_bytes(seed=2, n_classes=97):
    def make():
        from . import Model
        g = mf.build_recognition(n_classes=n_classes, seed=seed)
        cal = synth.synthetic_line_crops(7, n=8)
        xp = np.full((8, 1, 64, 300), -0.5, np.float32)
        xp[:, 0, :, :256] = cal
        g = mf.calibrate_recognition_head(g, lambda buf, x: Model.load_bytes(buf).run(x), xp)
        return g.to_bytes()

    return _cached("rec_c%d_s%d_v1.ocrsm" % (n_classes, seed), make)
