#!/usr/bin/env python
"""Round 5, verdict item 2: what the relaxed numerics (ocrs_engine_params.numerics = relaxed) change and what they buy.

    python tools/relaxed_report.py [out.json]

Outputs: flips between an exact and a relaxed engine over the 16 bench pages, the 2 048 crops of BASELINE configs[2] and the
reference's three images (tests/golden/reference/*.npz pixels, the synthetic models those fixtures use), the largest
log-prob / probability-map differences, and the kernel-class times of one 16-page request alone in both modes.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "16")

import numpy as np  # noqa: E402

import models_util as M  # noqa: E402
from ocrs_amd import DimOrder, ImageSource, Model, OcrEngine, numerics_report as NR, synth  # noqa: E402


MODES = ("relaxed", "reduced")


def engines(dbuf, rbuf):
    det, rec = Model.load_bytes(dbuf), Model.load_bytes(rbuf)
    return {m: OcrEngine(detection_model=det, recognition_model=rec, numerics=m) for m in ("exact",) + MODES}


def class_times(engine, pages):
    """kernel-class ms of ONE 16-page request on an otherwise idle GPU (every launch timed with HIP events)"""
    inputs = [engine.prepare_input(ImageSource.from_tensor(p, DimOrder.Hwc)) for p in pages]
    def run():
        words = engine.detect_words_batch(inputs)
        rects, lo, po = engine.find_text_lines_batch_raw(words)
        return engine.recognize_text_batch_raw(inputs, rects, lo, po)
    run()
    engine.enable_timing(2)
    engine.set_kernel_timing_classes(None)
    engine.kernel_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    wall = (time.perf_counter() - t0) / 3
    ks = {k: round(v["ms"] / 3, 3) for k, v in engine.kernel_stats(reset=True).items() if v["launches"] > 0 and v["ms"] / 3 > 0.05}
    engine.enable_timing(0)
    return {"request_ms_alone": round(1e3 * wall, 2), "kernel_class_ms": dict(sorted(ks.items(), key=lambda kv: -kv[1]))}


def main():
    out = {}
    eng = engines(M.detection_model_bytes(), M.recognition_model_bytes())
    exact = eng["exact"]
    pages = [synth.synthetic_page(s, 1024, 1024, lines=80) for s in range(16)]
    if "--once" in sys.argv:    # under rocprofv3: a few 16-page requests in one mode, nothing else
        print(json.dumps(class_times(eng[sys.argv[sys.argv.index("--once") + 1]], pages)))
        return
    if "--quick" in sys.argv:   # kernel iteration: two pages of flips, the class times of every mode
        for m in MODES:
            out[m] = NR.compare_pixels(exact, eng[m], pages[:2])
            out[m].pop("flipped", None)
        out["alone"] = {m: class_times(e, pages) for m, e in eng.items()}
        print(json.dumps(out, indent=1))
        return
    inp, lines = NR.crops_request(exact, synth)
    refs = {}
    for name in ("why-rust", "polar-bears", "rust-book"):
        g = np.load(os.path.join(ROOT, "tests", "golden", "reference", name + ".npz"))
        refs[name] = (engines(M.detection_model_bytes(ink=tuple(g["ink"])), M.recognition_model_bytes()), g["pixels"])
    for m in MODES:
        o = out[m] = {}
        o["bench_pages_16"] = NR.compare_pixels(exact, eng[m], pages)
        o["crops_2048"] = NR.compare_page(exact, eng[m], inp, lines=lines)
        each = {name: NR.compare_pixels(e2["exact"], e2[m], [px]) for name, (e2, px) in refs.items()}
        o["reference_images"] = NR.merge(each.values())
        o["reference_images_each"] = each
        o["total"] = NR.merge([o["bench_pages_16"], o["crops_2048"], o["reference_images"]])
    out["alone"] = {m: class_times(e, pages) for m, e in eng.items()}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
