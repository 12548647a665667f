#!/usr/bin/env python
"""Generates tests/golden/*.npz with the CPU ORACLE (the reference cannot be run
here: no Rust toolchain, no weights — DESIGN.md §3).  The fixtures freeze the
oracle's answers on small seeded inputs so that (a) oracle drift is caught by
the CPU suite and (b) the GPU suite has a committed target besides the live
oracle.  Re-run only on a deliberate change of the numeric spec:

    python tests/golden/make_golden.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import models_util as M  # noqa: E402
from ocrs_amd import synth  # noqa: E402
from oracle import pipeline as OP  # noqa: E402
from oracle.nn import OracleGraph, OracleModel  # noqa: E402

DET_HW, DET_DEPTHS = (160, 128), (8, 16, 32, 32)
PAGE = dict(seed=11, height=200, width=288, lines=8, columns=1)


def main():
    dbuf = M.detection_model_bytes(DET_HW, DET_DEPTHS)
    rbuf = M.recognition_model_bytes()
    ora = OP.OcrEngine(detection_model=OracleModel(OracleGraph(dbuf), "exact"),
                       recognition_model=OracleModel(OracleGraph(rbuf), "exact"))
    px = synth.synthetic_page(**PAGE)
    inp = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
    prob = ora.detect_text_pixels(inp)
    words = ora.detect_words(inp)
    lines = ora.find_text_lines(inp, words)
    rec = ora.recognizer
    tokens = []
    for l in lines:
        poly, rw = rec._line_geometry(l)
        gw = -(-rw // 50) * 50
        batch = np.full((1, 1, 64, gw), -0.5, np.float32)
        from oracle import clib
        clib.prepare_text_line_into(inp[0], [(p[1], p[0]) for p in poly], rw, 64, batch[0, 0])
        tokens.append(clib.ctc_greedy(rec.run(batch)[0]))
    text = ora.get_text(inp)
    tok_flat = np.array([t for ts in tokens for t in ts], np.int32).reshape(-1, 2)
    tok_off = np.cumsum([0] + [len(ts) for ts in tokens]).astype(np.int32)
    line_off = np.cumsum([0] + [len(l) for l in lines]).astype(np.int32)
    np.savez_compressed(
        os.path.join(HERE, "pipeline_small.npz"),
        page=px[:, :, 0], model_digests=np.array([M.digest(dbuf), M.digest(rbuf)]),
        grey_crc=np.array([np.frombuffer(inp.tobytes(), np.uint32).sum(dtype=np.uint64)]),
        prob_bits_sum=np.array([np.frombuffer(prob.tobytes(), np.uint32).sum(dtype=np.uint64)]),
        mask=np.packbits(prob > np.float32(0.2)),
        word_rects=np.array([w.to_array() for w in words], np.float32).reshape(-1, 6),
        line_rects=np.array([w.to_array() for l in lines for w in l], np.float32).reshape(-1, 6),
        line_offsets=line_off, tokens=tok_flat, token_offsets=tok_off, text=np.array([text]))
    # a recognition-only fixture: 3 crops -> log-prob arg-max path
    crops = synth.synthetic_line_crops(21, n=3, width=96).astype(np.float16).astype(np.float32)  # stored as f16
    x = np.full((3, 1, 64, 100), -0.5, np.float32)
    x[:, 0, :, :96] = crops
    lp = OracleGraph(rbuf).run_exact(x)
    np.savez_compressed(os.path.join(HERE, "recognition_small.npz"), crops=crops.astype(np.float16),
                        argmax=lp.argmax(-1).astype(np.uint8),
                        logp_bits_sum=np.array([np.frombuffer(lp.tobytes(), np.uint32).sum(dtype=np.uint64)]))
    print("wrote golden fixtures:", len(words), "words,", len(lines), "lines,", len(text), "chars")


def bench_page_words():
    """tests/golden/bench_page_words_seed{0,1}.npy: the ~700 word boxes the detection stage finds on the bench's
    1024x1024 synthetic pages (seeds 0 and 1, 80 lines) with the bench's detection model (ocrs_amd.modelfile
    build_detection(seed=1), evaluated by the oracle's torch backend).  Input of the bench-scale layout test."""
    from ocrs_amd import modelfile as mf
    dbuf = mf.build_detection(in_hw=(800, 600), seed=1).to_bytes()
    ora = OP.OcrEngine(detection_model=OracleModel(OracleGraph(dbuf), "torch"))
    for seed in (0, 1):
        px = synth.synthetic_page(seed, 1024, 1024, lines=80)
        oin = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
        words = ora.detect_words(oin)
        a = np.array([w.to_array() for w in words], np.float32).reshape(-1, 6)
        np.save(os.path.join(HERE, "bench_page_words_seed%d.npy" % seed), a)
        print("bench_page_words_seed%d.npy: %d words" % (seed, len(a)))


if __name__ == "__main__":
    if "--bench-page-words" in sys.argv:
        bench_page_words()
        sys.exit(0)
    main()
