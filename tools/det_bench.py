#!/usr/bin/env python
"""Detection-only loop (BASELINE.json configs[1]: 8 synthetic 1024x1024 pages, CNN + threshold + components -> rects)
for profiling:  rocprofv3 --kernel-trace --stats -- python tools/det_bench.py [reps]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ocrs_amd import DimOrder, Model, OcrEngine, _lib, models, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = _lib.lib()
eng = OcrEngine(detection_model=Model.load_bytes(models.synthetic_detection_bytes()))
dptrs = []
for s in range(8):
    pg = synth.synthetic_page(s, 1024, 1024, lines=80)
    p = C.c_void_p()
    _lib.check(L.ocrs_device_malloc(C.c_size_t(pg.nbytes), C.byref(p)))
    _lib.check(L.ocrs_device_upload(p, pg.ctypes.data_as(C.c_void_p), C.c_size_t(pg.nbytes)))
    dptrs.append(p)
inputs = [eng.prepare_input_device(p.value, np.uint8, DimOrder.Hwc, 1024, 1024, 3) for p in dptrs]
for _ in range(3):
    eng.detect_words_batch(inputs)
_lib.check(L.ocrs_device_synchronize())
t0 = time.perf_counter()
for _ in range(reps):
    words = eng.detect_words_batch(inputs)
_lib.check(L.ocrs_device_synchronize())
dt = time.perf_counter() - t0
print("det_fuse=%s: %.1f pages/s (%.3f ms per 8 pages), %d words on page 0" % (
    os.environ.get("OCRS_DET_FUSE", "1"), reps * 8 / dt, 1e3 * dt / reps, len(words[0])))
if os.environ.get("DET_BENCH_INFLIGHT"):
    from concurrent.futures import ThreadPoolExecutor
    k = int(os.environ["DET_BENCH_INFLIGHT"])
    with ThreadPoolExecutor(k) as pool:
        list(pool.map(lambda _: eng.detect_words_batch(inputs), range(k)))
        _lib.check(L.ocrs_device_synchronize())
        t0 = time.perf_counter()
        list(pool.map(lambda _: eng.detect_words_batch(inputs), range(3 * reps)))
        _lib.check(L.ocrs_device_synchronize())
        dt = time.perf_counter() - t0
    print("%d requests in flight: %.1f pages/s" % (k, 3 * reps * 8 / dt))
