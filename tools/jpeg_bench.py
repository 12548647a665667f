#!/usr/bin/env python
"""JPEG hand-off timing: N decodes of one bench page (4:2:0 q90) through ocrs_jpeg_decode_rgb /
prepare_input_jpeg; run under `rocprofv3 --kernel-trace --stats` for the kernel durations.  python tools/jpeg_bench.py [N]"""
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

from ocrs_amd import Model, OcrEngine, _lib, models, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
eng = OcrEngine(detection_model=Model.load_bytes(models.synthetic_detection_bytes()), recognition_model=None)
px = synth.synthetic_page(0, 1024, 1024, lines=80)
out = {}
for name, kw in (("420_q90", dict(quality=90, subsampling=2)), ("444_q90", dict(quality=90, subsampling=0)),
                 ("420_q90_progressive", dict(quality=90, subsampling=2, progressive=True))):
    b = io.BytesIO()
    Image.fromarray(px).save(b, "JPEG", **kw)
    data = b.getvalue()
    eng.prepare_input_jpeg(data)
    t0 = time.perf_counter()
    for _ in range(n):
        inp, cb = eng.prepare_input_jpeg(data)
    e2e = 1e3 * (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        _lib.jpeg_info(data)
    host = 1e3 * (time.perf_counter() - t0) / n
    print("%s: file %d B, to GPU %d B, host entropy decode %.2f ms, prepare_input_jpeg %.2f ms" % (name, len(data), cb, host, e2e))
