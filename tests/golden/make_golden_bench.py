#!/usr/bin/env python
"""Bench-scale golden fixtures, made with the CPU ORACLE in `exact` mode (slow: minutes of CPU).

    python tests/golden/make_golden_bench.py pages [seed ...]   # default seeds 0 1
    python tests/golden/make_golden_bench.py crops
    python tests/golden/make_golden_bench.py odd_large | odd_small   # pages of other sizes (ODD_PAGES below)

`pages`: BASELINE.json configs[3] — the bench's own 1024x1024 synthetic pages (ocrs_amd.synth.synthetic_page(seed,
1024, 1024, lines=80)) through the whole oracle pipeline with the bench's two models: word rects, line grouping,
greedy-CTC steps (label, pos) of every line, char boxes, text.  One file per seed:
tests/golden/bench_page_seed<seed>.npz.

`crops`: configs[2] — 2048 synthetic 64x256 crops (seed 1000) stacked into one tall page exactly as bench.py's
recognition-only leg does, each crop one line -> width group 300, T = 75: CTC steps and char boxes of all 2048 lines
(tests/golden/bench_crops_2048.npz).

The fixtures are what tests/test_gpu_bench_scale.py compares the HIP path with on the GPU box (where the oracle
would need minutes per page).  Re-run only on a deliberate change of the numeric spec.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import models_util as M  # noqa: E402
from ocrs_amd import synth  # noqa: E402
from oracle import clib  # noqa: E402
from oracle import pipeline as OP  # noqa: E402
from oracle.geometry import RotatedRect  # noqa: E402
from oracle.nn import OracleGraph, OracleModel  # noqa: E402


def recognize_with_steps(ora, inp, lines):
    """OP.TextRecognizer.recognize_text_lines (recognition.rs:404-545) that also returns the CTC steps."""
    rec = ora.recognizer
    h = rec.input_height()
    groups, order = {}, []
    for idx, word_rects in enumerate(lines):
        poly, rw = rec._line_geometry(word_rects)
        gw = -(-rw // 50) * 50
        if gw not in groups:
            groups[gw] = []
            order.append(gw)
        groups[gw].append((idx, poly, rw))
    res = {}
    for gw in order:
        members = groups[gw]
        for c0 in range(0, len(members), 20):
            chunk = members[c0:c0 + 20]
            batch = np.full((len(chunk), 1, h, gw), OP.BLACK_VALUE, np.float32)
            for bi, (idx, poly, rw) in enumerate(chunk):
                clib.prepare_text_line_into(inp[0], [(p[1], p[0]) for p in poly], rw, h, batch[bi, 0])
            out = rec.run(batch)
            for bi, (idx, poly, rw) in enumerate(chunk):
                steps = clib.ctc_greedy(out[bi])
                res[idx] = (steps, OP.text_line_from_result(poly, rw, gw, out.shape[1], steps, ora.alphabet))
    return [res[i] for i in range(len(lines))]


def pack(results):
    toks, toff, chars, coff = [], [0], [], [0]
    for steps, tl in results:
        toks += [(int(a), int(b)) for a, b in steps]
        toff.append(len(toks))
        if tl is not None:
            for c in tl.chars:
                t, l, b, r = c.rect.tlbr()
                chars.append((ord(c.char), t, l, b, r))
        coff.append(len(chars))
    return (np.array(toks, np.int32).reshape(-1, 2), np.array(toff, np.int64),
            np.array(chars, np.int32).reshape(-1, 5), np.array(coff, np.int64))


def engine():
    dbuf, rbuf = M.detection_model_bytes(), M.recognition_model_bytes()
    ora = OP.OcrEngine(detection_model=OracleModel(OracleGraph(dbuf), "exact"),
                       recognition_model=OracleModel(OracleGraph(rbuf), "exact"))
    return ora, np.array([M.digest(dbuf), M.digest(rbuf)])


ODD_PAGES = {   # name -> (seed, height, width, lines, columns): sizes the bench does not use
    "odd_large": (101, 2200, 3000, 60, 2),    # larger than the detection input (800x600) in both dimensions, landscape
    "odd_small": (102, 97, 211, 3, 1),        # smaller than the detection input, odd dimensions
}


def pages(seeds, odd=None):
    ora, digests = engine()
    for seed in seeds:
        t0 = time.time()
        if odd:
            seed, hh, ww, nl, ncol = ODD_PAGES[odd]
            px = synth.synthetic_page(seed, hh, ww, lines=nl, columns=ncol)
        else:
            px = synth.synthetic_page(seed, 1024, 1024, lines=80)
        inp = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
        words = ora.detect_words(inp)
        lines = ora.find_text_lines(inp, words)
        results = recognize_with_steps(ora, inp, lines)
        toks, toff, chars, coff = pack(results)
        np.savez_compressed(
            os.path.join(HERE, ("page_%s.npz" % odd) if odd else ("bench_page_seed%d.npz" % seed)), model_digests=digests,
            word_rects=np.array([w.to_array() for w in words], np.float32).reshape(-1, 6),
            line_rects=np.array([w.to_array() for l in lines for w in l], np.float32).reshape(-1, 6),
            line_offsets=np.cumsum([0] + [len(l) for l in lines]).astype(np.int64),
            tokens=toks, token_offsets=toff, chars=chars, char_offsets=coff)
        print("seed %d: %d words, %d lines, %d tokens, %d chars in %.0f s" % (
            seed, len(words), len(lines), len(toks), len(chars), time.time() - t0), flush=True)


def crops():
    ora, digests = engine()
    n = 2048
    t0 = time.time()
    cr = synth.synthetic_line_crops(1000, n=n)
    page = (cr.reshape(1, n * 64, 256) + 0.5).astype(np.float32)
    inp = ora.prepare_input(OP.ImageSource.from_tensor(page, "chw"))
    lines = []
    for i in range(n):
        lines.append([RotatedRect.from_array(np.array([128.0, i * 64.0 + 32.0, 0.0, 1.0, 256.0, 64.0], np.float32))])
    results = recognize_with_steps(ora, inp, lines)
    toks, toff, chars, coff = pack(results)
    np.savez_compressed(os.path.join(HERE, "bench_crops_2048.npz"), model_digests=digests[1:],
                        tokens=toks, token_offsets=toff, chars=chars, char_offsets=coff)
    print("crops: %d lines, %d tokens, %d chars in %.0f s" % (n, len(toks), len(chars), time.time() - t0), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "pages"
    if what == "pages":
        pages([int(s) for s in sys.argv[2:]] or [0, 1])
    elif what == "crops":
        crops()
    elif what in ODD_PAGES:
        pages([0], odd=what)
    else:
        raise SystemExit(__doc__)
