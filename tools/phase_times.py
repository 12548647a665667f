"""Wall-clock time of each pipeline phase of one bench step (host + GPU, synchronous API)."""
import ctypes as C, os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ocrs_amd import DimOrder, Model, OcrEngine, _lib, models, synth
L = _lib.lib()
det = Model.load_bytes(models.synthetic_detection_bytes()); rec = Model.load_bytes(models.synthetic_recognition_bytes())
eng = OcrEngine(detection_model=det, recognition_model=rec)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pages = [synth.synthetic_page(i) for i in range(B)]
dptrs = []
for pg in pages:
    p = C.c_void_p(); _lib.check(L.ocrs_device_malloc(C.c_size_t(pg.nbytes), C.byref(p))); _lib.check(L.ocrs_device_upload(p, pg.ctypes.data_as(C.c_void_p), C.c_size_t(pg.nbytes))); dptrs.append(p)
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0) + (t - t0); return t
for it in range(6):
    if it == 2: acc.clear()
    t = time.perf_counter()
    inputs = [eng.prepare_input_device(p.value, np.uint8, DimOrder.Hwc, 1024, 1024, 3) for p in dptrs]; t = tick("prepare", t)
    words = eng.detect_words_batch(inputs); t = tick("detect", t)
    rects, lo, po = eng.find_text_lines_batch_raw(words); t = tick("layout", t)
    chars, co = eng.recognize_text_batch_raw(inputs, rects, lo, po); t = tick("recognize", t)
    del inputs; t = tick("free", t)
print({k: round(1000 * v / 4, 2) for k, v in acc.items()}, "ms/step; total", round(1000 * sum(acc.values()) / 4, 2))
