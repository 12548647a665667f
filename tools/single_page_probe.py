#!/usr/bin/env python
"""Where a one-page request spends its time with N requests in flight (the reference's call pattern):
wall time of each of the four calls, per thread, averaged.  python tools/single_page_probe.py [threads] [requests]"""
import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402

from ocrs_amd import DimOrder, Model, OcrEngine, _lib, models, synth  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n_req = int(sys.argv[2]) if len(sys.argv) > 2 else 360
L = _lib.lib()
# optional: coalesce=N coalesce_pages=N coalesce_window_us=N (ocrs_engine_params fields) as further arguments
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[3:] if "=" in a}
eng = OcrEngine(detection_model=Model.load_bytes(models.synthetic_detection_bytes()),
                recognition_model=Model.load_bytes(models.synthetic_recognition_bytes()), **kw)
dptrs = []
for s in range(16):
    pg = synth.synthetic_page(s, 1024, 1024, lines=80)
    p = C.c_void_p()
    _lib.check(L.ocrs_device_malloc(C.c_size_t(pg.nbytes), C.byref(p)))
    _lib.check(L.ocrs_device_upload(p, pg.ctypes.data_as(C.c_void_p), C.c_size_t(pg.nbytes)))
    dptrs.append(p)


def one(i):
    t0 = time.perf_counter()
    inp = eng.prepare_input_device(dptrs[i % 16].value, np.uint8, DimOrder.Hwc, 1024, 1024, 3)
    t1 = time.perf_counter()
    w = eng.detect_words_batch([inp])
    t2 = time.perf_counter()
    r, lo, po = eng.find_text_lines_batch_raw(w)
    t3 = time.perf_counter()
    ch, _ = eng.recognize_text_batch_raw([inp], r, lo, po)
    t4 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2, t4 - t3


with ThreadPoolExecutor(threads) as ex:
    list(ex.map(one, range(3 * threads)))
    eng.enable_timing(1)
    eng.stage_times(reset=True)
    t0 = time.perf_counter()
    out = list(ex.map(one, range(n_req)))
    dt = time.perf_counter() - t0
a = np.array(out) * 1e3
st = eng.stage_times(reset=True)
print("threads %d: %.1f pages/s; per request ms: prepare %.2f detect %.2f layout %.2f recognize %.2f (sum %.2f); p99 recognize %.1f detect %.1f" % (
    threads, n_req / dt, *a.mean(0), a.sum(1).mean(), np.percentile(a[:, 3], 99), np.percentile(a[:, 1], 99)))
print("   coalesce", eng.coalesce_stats(), " GPU stage ms per request:", {k: round(v[0] / n_req, 2) for k, v in st.items() if v[0] > 0})
if os.environ.get("PROBE_KERNELS"):
    with ThreadPoolExecutor(threads) as ex:
        eng.enable_timing(2)
        eng.set_kernel_timing_classes(["gemm_conv3x3_mfma", "gemm_gru_input_mfma", "gemm_gru_hidden_mfma"])
        eng.kernel_stats(reset=True)
        list(ex.map(one, range(n_req)))
        ks = eng.kernel_stats(reset=True)
    print("   live kernels:", {k: "%.2f ms x %d, %.1f TFLOP/s" % (v["ms"] / max(v["launches"], 1), v["launches"], v["flops"] / max(v["ms"], 1e-9) / 1e9)
                               for k, v in ks.items() if v["launches"]})
