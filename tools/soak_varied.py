#!/usr/bin/env python
"""Memory behaviour under requests of varied sizes (round 5, verdict item 6):  python tools/soak_varied.py [seconds] [threads] [--numerics relaxed|reduced]

Pages of random sizes (200-3000 pixels a side, 1-200 lines, one or two columns) through the one-page pipeline from several
threads for a while.  Asserts: every result equals the sequential run's (bytes of the word rects, tokens); the bytes in use
(ocrs_device_pool_stats) return to the idle level; the cached bytes never exceed their caps.  Prints the peaks."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402

from ocrs_amd import DimOrder, ImageSource, Model, OcrEngine, _lib, models, synth  # noqa: E402

numerics = "exact"
if "--numerics" in sys.argv and sys.argv[sys.argv.index("--numerics") + 1] == "exact":
    i = sys.argv.index("--numerics")
    del sys.argv[i:i + 2]
if "--numerics" in sys.argv:   # the same check for the relaxed modes' kernels: a mode's results do not depend on what shares its launches
    i = sys.argv.index("--numerics")
    numerics = sys.argv[i + 1]
    del sys.argv[i:i + 2]
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
eng = OcrEngine(detection_model=Model.load_bytes(models.synthetic_detection_bytes()),
                recognition_model=Model.load_bytes(models.synthetic_recognition_bytes()), numerics=numerics)
rng = np.random.default_rng(2025)
cases = []
for s in range(48):
    h, w = int(rng.integers(200, 3001)), int(rng.integers(200, 3001))
    lines = int(rng.integers(1, max(2, min(200, h // 14))))
    cases.append(synth.synthetic_page(500 + s, h, w, lines=lines, columns=1 + (w > 1200)))


def run(px):
    inp = eng.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    words = eng.detect_words(inp)
    toks = eng.recognize_tokens(inp, eng.find_text_lines(inp, words))
    return words.tobytes(), toks


ref = [run(px) for px in cases]
_lib.pool_trim()
idle = _lib.pool_stats()      # weights only: nothing cached, no arena
cap_dev, cap_pin = 4 << 30, 64 << 20      # small caps so that the trimmer really works during the run
_lib.pool_configure(device_cached_cap=cap_dev, pinned_cached_cap=cap_pin)
stop = time.time() + seconds
errors, done = [], [0]
peaks = {"device_live": 0, "device_cached": 0, "pinned_live": 0, "pinned_cached": 0}
lock = threading.Lock()


def worker(k):
    i = k
    try:
        while time.time() < stop:
            j = (i * 7 + k) % len(cases)
            assert run(cases[j]) == ref[j], "page %d differs from the sequential run" % j
            st = _lib.pool_stats()
            with lock:
                done[0] += 1
                for key in peaks:
                    peaks[key] = max(peaks[key], st[key])
                assert st["device_cached"] <= cap_dev and st["pinned_cached"] <= cap_pin, st
            i += 1
    except Exception as e:  # noqa: BLE001
        errors.append("%d: %r" % (k, e))


threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
t0 = time.time()
for t in threads:
    t.start()
for t in threads:
    t.join()
dt = time.time() - t0
busy_end = _lib.pool_stats()
_lib.pool_trim()
end = _lib.pool_stats()
if end["device_live"] > idle["device_live"] + (1 << 20) or end["pinned_live"] > idle["pinned_live"] + (1 << 16):
    errors.append("bytes in use did not return to the idle level: %s -> %s" % (idle, end))
print("soak_varied %.0f s, %d threads: %d pages of %d distinct sizes (%.1f pages/s); peaks %s MB; in use when the load stops %d MB (weights + the shared "
      "activation arena at its high-water mark), after ocrs_device_pool_trim %d MB (idle level %d MB); driver allocs / frees: device %d / %d, "
      "pinned %d / %d; caps %d / %d MB; errors: %s" % (
          dt, n_threads, done[0], len(cases), done[0] / dt, {k: v >> 20 for k, v in peaks.items()},
          busy_end["device_live"] >> 20, end["device_live"] >> 20, idle["device_live"] >> 20,
          end["device_driver_allocs"] - idle["device_driver_allocs"], end["device_driver_frees"] - idle["device_driver_frees"],
          end["pinned_driver_allocs"] - idle["pinned_driver_allocs"], end["pinned_driver_frees"] - idle["pinned_driver_frees"],
          cap_dev >> 20, cap_pin >> 20, errors or "none"))
sys.exit(1 if errors else 0)
