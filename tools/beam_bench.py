#!/usr/bin/env python
"""DecodeMethod::BeamSearch(100) on one bench page (1024x1024, ~77 lines, T up to 600): GPU kernel vs host threads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ocrs_amd import DecodeMethod, DimOrder, ImageSource, Model, OcrEngine, _lib, models, synth

rng = np.random.default_rng(0)
lp = rng.normal(0, 3, (600, 97)); lp = (lp - np.log(np.exp(lp).sum(1, keepdims=True))).astype(np.float32)
for impl in (0, 2):
    t0 = time.perf_counter(); r = _lib.ctc_beam_search(lp, 100, impl); dt = time.perf_counter() - t0
    print("one 600x97 matrix, width 100, impl %d: %.1f ms (%d steps)" % (impl, dt * 1e3, len(r)))
det = Model.load_bytes(models.synthetic_detection_bytes()); rec = Model.load_bytes(models.synthetic_recognition_bytes())
px = synth.synthetic_page(0, 1024, 1024, lines=80)
res = {}
for name, dm in (("greedy", DecodeMethod.Greedy), ("beam100", DecodeMethod.BeamSearch(100))):
    eng = OcrEngine(detection_model=det, recognition_model=rec, decode_method=dm)
    inp = eng.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    lines = eng.find_text_lines(inp, eng.detect_words(inp))
    for mode in ((1, 0) if name == "beam100" else (1,)):
        _lib.set_option("beam_gpu", mode)
        eng.recognize_text(inp, lines[:4])
        t0 = time.perf_counter(); out = eng.recognize_text(inp, lines); dt = time.perf_counter() - t0
        res[(name, mode)] = [str(t) if t else None for t in out]
        print("%s beam_gpu=%d: recognize_text of %d lines in %.1f ms" % (name, mode, len(lines), dt * 1e3))
_lib.set_option("beam_gpu", 1)
print("gpu == host:", res[("beam100", 1)] == res[("beam100", 0)], "| beam == greedy text on", sum(a == b for a, b in zip(res[("beam100", 1)], res[("greedy", 1)])), "of", len(res[("greedy", 1)]), "lines")
