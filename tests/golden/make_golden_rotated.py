#!/usr/bin/env python
"""Golden fixtures for ROTATED / SKEWED pages, made with the CPU oracle in `exact` mode (round 5, verdict item 4).

    python tests/golden/make_golden_rotated.py [name:angle ...]        # default: all 15

Every full-pipeline parity input before round 5 was upright.  Here the reference's three images
(tests/golden/reference/<name>.npz holds their RGB8 pixels) are rotated by +3, -3, +10, -10 degrees (PIL
`Image.rotate(angle, resample=BICUBIC, expand=True, fillcolor=<the page's background>)`) and by 90 degrees
(`Image.transpose(ROTATE_90)`: exact), so that word rects carry real `up` vectors, lines are grouped along a slope
(layout_analysis.rs:19-155 on rotated rects), the line polygons are slanted quadrilaterals (recognition.rs:29-55, scanline fill
91-126), crops are resampled from rotated boxes and char boxes are cut from slanted polygons (recognition.rs:162-193,
text_items.rs:25-31, geom_util.rs:6-26).  At 90 degrees the text runs bottom-to-top: lines are vertical columns of word rects.

The rotated pixels are NOT stored (15 photos would be 20 MB): the test rebuilds them with the same PIL call from the stored
base pixels and checks their CRC against the fixture (skipping if this PIL build resamples differently).

Per case, tests/golden/rotated/<name>_<angle>.npz: pixel CRC + shape, mask bits, probability-map checksum, word rects, line
grouping, per-line crop checksums (prepare_recognition_input of the first 6 lines), greedy-CTC steps, char boxes, text.
Same synthetic weights and per-image "ink" operating points as make_golden_reference_images.py.
"""
import os
import sys
import time
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import models_util as M  # noqa: E402
from make_golden_bench import pack, recognize_with_steps  # noqa: E402
from oracle import pipeline as OP  # noqa: E402
from oracle.nn import OracleGraph, OracleModel  # noqa: E402

NAMES = ("why-rust", "polar-bears", "rust-book")
ANGLES = (3, -3, 10, -10, 90)
# background a rotation exposes: the page's own (why-rust is light text on a dark page)
FILL = {"why-rust": (50, 112, 98), "polar-bears": (255, 255, 255), "rust-book": (165, 160, 146)}
N_CROPS = 6


def rotated_pixels(px, name, angle):
    from PIL import Image
    im = Image.fromarray(px, "RGB")
    if angle == 90:
        return np.ascontiguousarray(np.asarray(im.transpose(Image.ROTATE_90)))
    return np.ascontiguousarray(np.asarray(im.rotate(angle, resample=Image.BICUBIC, expand=True, fillcolor=FILL[name])))


def bits_sum(a):
    return int(np.frombuffer(np.ascontiguousarray(a).tobytes(), np.uint32).sum(dtype=np.uint64))


def main(cases):
    out_dir = os.path.join(HERE, "rotated")
    os.makedirs(out_dir, exist_ok=True)
    engines = {}
    for name, angle in cases:
        t0 = time.time()
        g = np.load(os.path.join(HERE, "reference", name + ".npz"))
        if name not in engines:
            dbuf, rbuf = M.detection_model_bytes(ink=tuple(g["ink"])), M.recognition_model_bytes()
            engines[name] = (OP.OcrEngine(detection_model=OracleModel(OracleGraph(dbuf), "exact"),
                                          recognition_model=OracleModel(OracleGraph(rbuf), "exact")),
                             np.array([M.digest(dbuf), M.digest(rbuf)]))
        ora, digests = engines[name]
        px = rotated_pixels(g["pixels"], name, angle)
        inp = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
        prob = ora.detect_text_pixels(inp)
        mask = prob > np.float32(ora.detection_threshold())
        words = ora.detect_words(inp)
        lines = ora.find_text_lines(inp, words)
        crops = [ora.prepare_recognition_input(inp, l) for l in lines[:N_CROPS]]
        results = recognize_with_steps(ora, inp, lines)
        toks, toff, chars, coff = pack(results)
        text = "\n".join(str(tl) for _, tl in results if tl is not None)
        np.savez_compressed(
            os.path.join(out_dir, "%s_%+d.npz" % (name, angle)), model_digests=digests, ink=g["ink"], angle=np.array([angle]),
            pixel_crc=np.array([zlib.crc32(px.tobytes())], np.uint64), pixel_shape=np.array(px.shape, np.int64),
            grey_bits_sum=np.array([bits_sum(inp)], np.uint64), prob_bits_sum=np.array([bits_sum(prob)], np.uint64),
            mask=np.packbits(mask), mask_shape=np.array(mask.shape, np.int64),
            word_rects=np.array([w.to_array() for w in words], np.float32).reshape(-1, 6),
            line_rects=np.array([w.to_array() for l in lines for w in l], np.float32).reshape(-1, 6),
            line_offsets=np.cumsum([0] + [len(l) for l in lines]).astype(np.int64),
            crop_shapes=np.array([c.shape for c in crops], np.int64).reshape(-1, 2),
            crop_bits_sums=np.array([bits_sum(c) for c in crops], np.uint64),
            tokens=toks, token_offsets=toff, chars=chars, char_offsets=coff, text=np.array([text]))
        ups = np.array([w.to_array()[2:4] for w in words], np.float32).reshape(-1, 2)
        print("%s %+d %s: %d words (mean up %.3f, %.3f), %d lines, %d tokens, %d chars in %.0f s" % (
            name, angle, px.shape, len(words), float(ups[:, 0].mean()) if len(ups) else 0.0, float(ups[:, 1].mean()) if len(ups) else 0.0,
            len(lines), len(toks), len(chars), time.time() - t0), flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    cases = [(a.split(":")[0], int(a.split(":")[1])) for a in args] if args else [(n, a) for n in NAMES for a in ANGLES]
    main(cases)
