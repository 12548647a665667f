#!/usr/bin/env python
"""Mixed-load soak: one-page pipelines (coalesced inside the engine), 16-page batch pipelines and engine-group calls
([0, 0]: two members on the one GPU) all at once for a while; every result is compared with the sequential reference
computed up front; any error or difference fails.  python tools/soak.py [seconds]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402

from ocrs_amd import DimOrder, EngineGroup, ImageSource, Model, OcrEngine, _lib, models, synth  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dbuf, rbuf = models.synthetic_detection_bytes(), models.synthetic_recognition_bytes()
eng = OcrEngine(detection_model=Model.load_bytes(dbuf), recognition_model=Model.load_bytes(rbuf))
group = EngineGroup([0, 0], dbuf, rbuf)
pages = [synth.synthetic_page(s, 1024, 1024, lines=80) for s in range(16)]


def pipeline(e, pgs, prepare):
    inputs = prepare(pgs)
    words = e.detect_words_batch(inputs)
    rects, lo, po = e.find_text_lines_batch_raw(words)
    chars, co = e.recognize_text_batch_raw(inputs, rects, lo, po)
    return [w.tobytes() for w in words], chars.tobytes(), co.tobytes()


prep_e = lambda pgs: [eng.prepare_input(ImageSource.from_tensor(p, DimOrder.Hwc)) for p in pgs]
prep_g = lambda pgs: group.prepare_input_batch(pgs)
ref1 = [pipeline(eng, [p], prep_e) for p in pages[:4]]
ref16 = pipeline(eng, pages, prep_e)
assert pipeline(group, pages, prep_g) == ref16
stop = time.time() + seconds
counts = {"one": 0, "batch": 0, "group": 0}
errors = []
lock = threading.Lock()


def worker(kind, k):
    try:
        while time.time() < stop:
            if kind == "one":
                i = (k + counts["one"]) % 4
                assert pipeline(eng, [pages[i]], prep_e) == ref1[i], "one-page result differs"
            elif kind == "batch":
                assert pipeline(eng, pages, prep_e) == ref16, "16-page result differs"
            else:
                assert pipeline(group, pages, prep_g) == ref16, "group result differs"
            with lock:
                counts[kind] += 1
    except Exception as e:  # noqa: BLE001
        errors.append("%s[%d]: %r" % (kind, k, e))


threads = [threading.Thread(target=worker, args=("one", k)) for k in range(6)]
threads += [threading.Thread(target=worker, args=("batch", k)) for k in range(3)]
threads += [threading.Thread(target=worker, args=("group", k)) for k in range(2)]
t0 = time.time()
for t in threads:
    t.start()
for t in threads:
    t.join()
dt = time.time() - t0
total = counts["one"] + 16 * (counts["batch"] + counts["group"])
print("soak %.0f s: %d one-page, %d 16-page, %d group requests = %.1f pages/s; coalesce %s; errors: %s" % (
    dt, counts["one"], counts["batch"], counts["group"], total / dt, eng.coalesce_stats(), errors or "none"))
sys.exit(1 if errors else 0)
