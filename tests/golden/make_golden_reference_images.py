#!/usr/bin/env python
"""Golden fixtures from the REFERENCE'S OWN images, made with the CPU oracle in `exact` mode (minutes of CPU).

    python tests/golden/make_golden_reference_images.py [why-rust polar-bears rust-book]

Inputs (read from /root/reference, which exists only in the build container — hence the pixels are stored in the
fixture): `ocrs-cli/test-data/why-rust.png` (2320x776 RGBA), `ocrs-cli/test-data/polar-bears.png` (1896x242 RGBA: shorter
than the 800-row detection input, i.e. the `pad` branch of detection.rs:159-160 at scale) and
`ocrs/examples/rust-book.jpg` (1200x1600 photo), decoded with PIL and reduced to RGB8 the way the CLI does
(`image.into_rgb8()`, ocrs-cli/src/main.rs:311-315: alpha dropped, not composited) — SURVEY.md §8c lists them as the
oracle's natural-image material.  Every parity input before round 4 was a synthetic page of rounded blobs; masks of
real glyphs (holes, nested borders, one-pixel bridges, touching letters) are where contour tracing / RDP / min-area-rect
restatements disagree first.

Per image, tests/golden/reference/<name>.npz holds: the RGB8 pixels, the thresholded text mask (packed bits) and a
checksum of the f32 probability map, word rects, line grouping, greedy-CTC steps of every line, char boxes, text.
Weights are the seeded synthetic ones of tests/models_util.py (the real ones are unobtainable offline).  The default
synthetic detection file only responds to the solid dark blobs of the synthetic pages (on these images it finds 2, 0 and
102 words), so each image gets a detection file whose hand-set "ink" feature sits at that image's operating point
(INK below: level, gain, sign — why-rust is light text on a dark page): the masks then follow the glyphs — 797 / 313 /
187 outer contours, borders of up to 3 778 points, letters with holes, touching letters, specks below min_area.
The decoded text is meaningless but deterministic.

Compared with the HIP path by tests/test_gpu_r4.py (one-page API and batch API); tests/test_golden.py re-derives the
cheap stages on CPU (drift guard).
"""
import os
import sys
import time

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

REF = "/root/reference"
IMAGES = {
    "why-rust": "ocrs-cli/test-data/why-rust.png",
    "polar-bears": "ocrs-cli/test-data/polar-bears.png",
    "rust-book": "ocrs/examples/rust-book.jpg",
}
INK = {"why-rust": (0.1, 1.0, -1), "polar-bears": (0.3, 1.0, 1), "rust-book": (-0.15, 1.0, 1)}


def engine(name):
    dbuf, rbuf = M.detection_model_bytes(ink=INK[name]), M.recognition_model_bytes()
    ora = OP.OcrEngine(detection_model=OracleModel(OracleGraph(dbuf), "exact"),
                       recognition_model=OracleModel(OracleGraph(rbuf), "exact"))
    return ora, np.array([M.digest(dbuf), M.digest(rbuf)])


def load_rgb8(path):
    """image::open(path).into_rgb8() (main.rs:311-322)."""
    from PIL import Image
    im = Image.open(path)
    if im.mode == "RGBA":
        return np.ascontiguousarray(np.asarray(im)[:, :, :3])      # alpha dropped
    return np.ascontiguousarray(np.asarray(im.convert("RGB")))


def bits_sum(a):
    return int(np.frombuffer(np.ascontiguousarray(a).tobytes(), np.uint32).sum(dtype=np.uint64))


def main(names):
    out_dir = os.path.join(HERE, "reference")
    os.makedirs(out_dir, exist_ok=True)
    for name in names:
        t0 = time.time()
        ora, digests = engine(name)
        px = load_rgb8(os.path.join(REF, IMAGES[name]))
        inp = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
        prob = ora.detect_text_pixels(inp)
        mask = prob > np.float32(ora.detection_threshold())
        words = ora.detect_words(inp)
        lines = ora.find_text_lines(inp, words)
        results = recognize_with_steps(ora, inp, lines)
        toks, toff, chars, coff = pack(results)
        text = "\n".join(str(tl) for _, tl in results if tl is not None)
        # the JPEG's own bytes travel too (the GPU box has no /root/reference): the JPEG hand-off of row f4 decodes THEM
        file_bytes = np.frombuffer(open(os.path.join(REF, IMAGES[name]), "rb").read(), np.uint8) if IMAGES[name].endswith(".jpg") \
            else np.zeros(0, np.uint8)
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"), model_digests=digests, ink=np.array(INK[name], np.float64), source=np.array([IMAGES[name]]), pixels=px,
            file_bytes=file_bytes,
            grey_bits_sum=np.array([bits_sum(inp)], np.uint64), prob_bits_sum=np.array([bits_sum(prob)], np.uint64),
            mask=np.packbits(mask), mask_shape=np.array(mask.shape, np.int64),
            word_rects=np.array([w.to_array() for w in words], np.float32).reshape(-1, 6),
            line_rects=np.array([w.to_array() for l in lines for w in l], np.float32).reshape(-1, 6),
            line_offsets=np.cumsum([0] + [len(l) for l in lines]).astype(np.int64),
            tokens=toks, token_offsets=toff, chars=chars, char_offsets=coff, text=np.array([text]))
        print("%s %s: mask %d px set, %d words, %d lines, %d tokens, %d chars in %.0f s" % (
            name, px.shape, int(mask.sum()), len(words), len(lines), len(toks), len(chars), time.time() - t0), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(IMAGES))
