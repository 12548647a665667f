#!/usr/bin/env python
"""Standing canary for silent corruption between concurrently running kernels (DESIGN.md §4.4 "Concurrency").

Round 5 found another request's line crops changing — 64-byte pieces, lanes 48..63 of one wave — while the bf16-split conv /
GEMM kernels of the relaxed numerics ran beside them.  This tool is the detector as a stand-alone program AND the body of
tests/test_gpu_r6.py's canary tests: every stage of the pipeline is called through the C ABI from several threads at once and
EVERY ELEMENT of every stage's output is compared with the value the same engine computed for the same input while it had
the device to itself ("twin" of a known-good run):

    class         entry point (kernels)                                              compared
    prepare       prepare_input (prepare_image)                                      the grey page, every float
    text_map      detect_text_pixels (resize, the detection U-Net, resize back)      the probability map, every float
    words         detect_words (+ threshold, component labelling, contours, rects)   every rect, bytes
    crop          prepare_recognition_input per line (crop_lines)                    every float of every crop
    logits        recognize_logits (crops, conv stack, projections, recurrence, head)  every log-probability
    tokens        recognize_tokens (+ greedy CTC)                                    every (label, position)

    python tools/hazard_canary.py --numerics exact --seconds 20              # the claim "exact mode is immune", as a measurement
    python tools/hazard_canary.py --numerics relaxed --isolation none        # reproduces the hazard (crop mismatches within seconds)
    python tools/hazard_canary.py --numerics relaxed --isolation auto        # what the library does by default: one stream

Prints one JSON line: per class the number of checks, mismatching checks and mismatching elements; for crops the histogram of
the mismatching columns mod 64 (round 5: always 48..63).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

CLASSES = ("prepare", "text_map", "words", "crop", "logits", "tokens")


def make_pages(np, synth, n_pages, big):
    rng = np.random.default_rng(11)
    pages = []
    for s in range(n_pages):
        if big and s % 3 == 0:
            h, w, lines = 1024, 1024, 80
        else:
            h, w = int(rng.integers(260, 1201)), int(rng.integers(400, 1601))
            lines = int(rng.integers(4, max(5, min(60, h // 16))))
        pages.append(synth.synthetic_page(900 + s, h, w, lines=lines, columns=1 + (w > 1200)))
    return pages


def run(numerics="exact", isolation="auto", seconds=20.0, threads=6, n_pages=9, big=True, classes=CLASSES,
        engine_options=None, log=None):
    """Returns the report dict.  The engine is created here and destroyed before returning; the device's isolation policy is
    put back to "auto"."""
    import numpy as np

    import models_util as M
    from ocrs_amd import DimOrder, ImageSource, Model, OcrEngine, _lib, synth

    _lib.require_gpu()
    _lib.set_isolation(isolation)
    det, rec = Model.load_bytes(M.detection_model_bytes()), Model.load_bytes(M.recognition_model_bytes())
    eng = OcrEngine(detection_model=det, recognition_model=rec, numerics=numerics, options=engine_options or {})
    iso = _lib.isolation()
    pixels = make_pages(np, synth, n_pages, big)
    # ---- the reference: one thread, the device to itself
    ref = []
    for px in pixels:
        src = ImageSource.from_tensor(px, DimOrder.Hwc)
        inp = eng.prepare_input(src)
        words = eng.detect_words(inp)
        lines = eng.find_text_lines(inp, words)
        r = {"src": src, "inp": inp, "lines": lines, "prepare": inp.image(), "text_map": eng.detect_text_pixels(inp), "words": words,
             "crop": [eng.prepare_recognition_input(inp, ln) for ln in lines], "logits": eng.recognize_logits(inp, lines),
             "tokens": eng.recognize_tokens(inp, lines)}
        ref.append(r)
    # determinism of the reference itself (a second quiet pass must agree before anything is blamed on concurrency)
    for r in ref:
        assert np.array_equal(eng.detect_text_pixels(r["inp"]), r["text_map"])
        assert all(np.array_equal(a, b) for a, b in zip(eng.recognize_logits(r["inp"], r["lines"]), r["logits"]))
    stats = {c: {"checks": 0, "bad_checks": 0, "bad_elements": 0, "elements": 0} for c in classes}
    crop_cols = [0] * 64
    examples = []
    mu = threading.Lock()
    stop = time.perf_counter() + seconds

    def note(cls, n_el, bad_el, detail=None):
        with mu:
            s = stats[cls]
            s["checks"] += 1
            s["elements"] += int(n_el)
            if bad_el:
                s["bad_checks"] += 1
                s["bad_elements"] += int(bad_el)
                if detail is not None and len(examples) < 12:
                    examples.append(detail)

    def diff(a, b):
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape:
            return max(a.size, b.size), None
        ne = a.view(np.uint32) != b.view(np.uint32) if a.dtype == np.float32 else a != b
        return int(ne.sum()), ne

    def check(cls, j):
        r = ref[j]
        if cls == "prepare":
            got = eng.prepare_input(r["src"]).image()
            n, _ = diff(got, r["prepare"])
            note(cls, got.size, n, {"class": cls, "page": j, "bad": n})
        elif cls == "text_map":
            got = eng.detect_text_pixels(r["inp"])
            n, _ = diff(got, r["text_map"])
            note(cls, got.size, n, {"class": cls, "page": j, "bad": n})
        elif cls == "words":
            got = eng.detect_words(r["inp"])
            n, _ = diff(got, r["words"])
            note(cls, got.size, n, {"class": cls, "page": j, "bad": n})
        elif cls == "crop":
            for li, ln in enumerate(r["lines"]):
                got = eng.prepare_recognition_input(r["inp"], ln)
                n, ne = diff(got, r["crop"][li])
                if n and ne is not None:
                    cols = np.nonzero(ne)[1]
                    with mu:
                        for c in cols:
                            crop_cols[int(c) % 64] += 1
                note(cls, got.size, n, {"class": cls, "page": j, "line": li, "bad": n,
                                        "cols": sorted(set(int(c) for c in np.nonzero(ne)[1]))[:40] if n and ne is not None else None})
                if time.perf_counter() > stop:
                    break
        elif cls == "logits":
            got = eng.recognize_logits(r["inp"], r["lines"])
            per_line = [diff(a, b)[0] for a, b in zip(got, r["logits"])]
            n = sum(per_line) + abs(len(got) - len(r["logits"]))
            detail = {"class": cls, "page": j, "bad": n}
            if n:   # which lines, how much of each, how far off: a different arithmetic path, or garbage
                detail["lines"] = [(li, c, int(got[li].size), float(np.nanmax(np.abs(got[li] - r["logits"][li]))))
                                   for li, c in enumerate(per_line) if c][:8]
                detail["n_lines"] = len(per_line)
            note(cls, sum(a.size for a in got), n, detail)
        elif cls == "tokens":
            got = eng.recognize_tokens(r["inp"], r["lines"])
            n = sum(1 for a, b in zip(got, r["tokens"]) if a != b)
            note(cls, len(got), n, {"class": cls, "page": j, "bad": n})

    errors = []

    def worker(k):
        # every thread walks the same cycle of classes, shifted: at any moment some threads run conv stacks and recurrences
        # (recognize_logits / tokens) while others run the small kernels next to them
        it = 0
        try:
            while time.perf_counter() < stop:
                cls = classes[(it + k) % len(classes)]
                check(cls, (it * 5 + k * 2) % len(ref))
                it += 1
        except Exception as e:   # noqa: BLE001
            errors.append("%s: %s" % (type(e).__name__, e))

    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    wall = time.perf_counter() - t0
    # the quiet pass again: nothing may have been corrupted persistently
    after = sum(int(not np.array_equal(eng.detect_words(r["inp"]), r["words"])) for r in ref)
    report = {"numerics": numerics, "isolation_policy": isolation, "isolation": iso, "threads": threads, "seconds": round(wall, 2),
              "pages": [list(p.shape[:2]) for p in pixels], "lines": [len(r["lines"]) for r in ref], "classes": stats,
              "mismatching_checks": sum(s["bad_checks"] for s in stats.values()),
              "crop_mismatch_columns_mod_64": crop_cols if any(crop_cols) else None, "examples": examples, "errors": errors,
              "sequential_pass_after_differs_on_pages": after}
    del ref, eng
    _lib.set_isolation("auto")
    if log:
        log(report)
    return report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--numerics", choices=("exact", "relaxed", "reduced"), default="exact")
    ap.add_argument("--isolation", choices=("auto", "none"), default="auto")
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--pages", type=int, default=9)
    ap.add_argument("--classes", type=str, default=",".join(CLASSES))
    ap.add_argument("--option", action="append", default=[], help="engine option name=value (e.g. gru_mode=1)")
    a = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.option}
    rep = run(a.numerics, a.isolation, a.seconds, a.threads, a.pages, classes=tuple(a.classes.split(",")), engine_options=opts)
    print(json.dumps(rep))
    sys.exit(1 if rep["errors"] else 0)
