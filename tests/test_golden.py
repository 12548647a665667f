"""Committed golden fixtures (tests/golden/, made by tests/golden/make_golden.py
with the CPU oracle).  CPU: the oracle still reproduces them (drift guard).
GPU: the HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

import models_util as M
from oracle import pipeline as OP
from oracle.nn import OracleGraph, OracleModel

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DET_HW, DET_DEPTHS = (160, 128), (8, 16, 32, 32)


def _bits_sum(a):
    return int(np.frombuffer(np.ascontiguousarray(a).tobytes(), np.uint32).sum(dtype=np.uint64))


def _load():
    g = np.load(os.path.join(G, "pipeline_small.npz"))
    dbuf, rbuf = M.detection_model_bytes(DET_HW, DET_DEPTHS), M.recognition_model_bytes()
    assert [M.digest(dbuf), M.digest(rbuf)] == list(g["model_digests"]), "synthetic model files changed: regenerate goldens"
    px = np.repeat(g["page"][:, :, None], 3, axis=2)
    return g, dbuf, rbuf, np.ascontiguousarray(px)


def test_oracle_reproduces_pipeline_golden():
    g, dbuf, rbuf, px = _load()
    ora = OP.OcrEngine(detection_model=OracleModel(OracleGraph(dbuf), "exact"),
                       recognition_model=OracleModel(OracleGraph(rbuf), "exact"))
    inp = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
    assert _bits_sum(inp) == int(g["grey_crc"][0])
    prob = ora.detect_text_pixels(inp)
    assert _bits_sum(prob) == int(g["prob_bits_sum"][0])
    assert np.array_equal(np.packbits(prob > np.float32(0.2)), g["mask"])
    words = ora.detect_words(inp)
    assert np.array_equal(np.array([w.to_array() for w in words], np.float32).reshape(-1, 6), g["word_rects"])
    lines = ora.find_text_lines(inp, words)
    assert np.array_equal(np.cumsum([0] + [len(l) for l in lines]), g["line_offsets"])
    assert np.array_equal(np.array([w.to_array() for l in lines for w in l], np.float32).reshape(-1, 6), g["line_rects"])
    assert ora.get_text(inp) == str(g["text"][0])


def test_oracle_reproduces_recognition_golden():
    g = np.load(os.path.join(G, "recognition_small.npz"))
    x = np.full((3, 1, 64, 100), -0.5, np.float32)
    x[:, 0, :, :96] = g["crops"].astype(np.float32)
    lp = OracleGraph(M.recognition_model_bytes()).run_exact(x)
    assert np.array_equal(lp.argmax(-1).astype(np.uint8), g["argmax"])
    assert _bits_sum(lp) == int(g["logp_bits_sum"][0])


@pytest.mark.gpu
def test_gpu_reproduces_pipeline_golden():
    from ocrs_amd import DimOrder, ImageSource, Model, OcrEngine
    g, dbuf, rbuf, px = _load()
    eng = OcrEngine(detection_model=Model.load_bytes(dbuf), recognition_model=Model.load_bytes(rbuf))
    inp = eng.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    assert _bits_sum(inp.image()) == int(g["grey_crc"][0])
    prob = eng.detect_text_pixels(inp)
    assert _bits_sum(prob) == int(g["prob_bits_sum"][0])
    words = eng.detect_words(inp)
    assert np.array_equal(words, g["word_rects"])
    lines = eng.find_text_lines(inp, words)
    assert np.array_equal(np.cumsum([0] + [len(l) for l in lines]), g["line_offsets"])
    assert np.array_equal(np.concatenate(lines), g["line_rects"])
    toks = eng.recognize_tokens(inp, lines)
    flat = np.array([t for ts in toks for t in ts], np.int32).reshape(-1, 2)
    assert np.array_equal(flat, g["tokens"])
    assert np.array_equal(np.cumsum([0] + [len(t) for t in toks]), g["token_offsets"])
    assert eng.get_text(inp) == str(g["text"][0])


@pytest.mark.gpu
def test_gpu_reproduces_recognition_golden():
    from ocrs_amd import Model
    g = np.load(os.path.join(G, "recognition_small.npz"))
    x = np.full((3, 1, 64, 100), -0.5, np.float32)
    x[:, 0, :, :96] = g["crops"].astype(np.float32)
    lp = Model.load_bytes(M.recognition_model_bytes()).run(x)
    assert np.array_equal(lp.argmax(-1).astype(np.uint8), g["argmax"])
    assert _bits_sum(lp) == int(g["logp_bits_sum"][0])


# ---- bench-scale fixtures (tests/golden/make_golden_bench.py): CPU drift guards --------------------------------
def test_oracle_and_host_layout_reproduce_bench_page_golden():
    """Seed-0 bench page (1024x1024, 80 lines): the oracle's exact detection still gives the golden word rects, and
    the PRODUCT's host layout analysis (layout.cpp through the C ABI, no GPU work) groups them into the golden lines
    (layout_analysis.rs:19-233 at the bench's ~700 words/page)."""
    from ocrs_amd import OcrEngine, synth
    g = np.load(os.path.join(G, "bench_page_seed0.npz"))
    dbuf, rbuf = M.detection_model_bytes(), M.recognition_model_bytes()
    assert [M.digest(dbuf), M.digest(rbuf)] == list(g["model_digests"]), "synthetic model files changed: regenerate goldens"
    ora = OP.OcrEngine(detection_model=OracleModel(OracleGraph(dbuf), "exact"))
    px = synth.synthetic_page(0, 1024, 1024, lines=80)
    words = ora.detect_words(ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc")))
    assert np.array_equal(np.array([w.to_array() for w in words], np.float32).reshape(-1, 6), g["word_rects"])
    lines = OcrEngine().find_text_lines(None, g["word_rects"])
    assert np.array_equal(np.cumsum([0] + [len(l) for l in lines]), g["line_offsets"])
    assert np.array_equal(np.concatenate(lines), g["line_rects"])


def test_oracle_reproduces_bench_crops_golden_sample():
    """configs[2] fixture: the first 40 of the 2048 crops through the oracle (crop -> width group 300 -> CRNN -> CTC)."""
    from ocrs_amd import synth
    from oracle import clib
    from oracle.geometry import RotatedRect
    g = np.load(os.path.join(G, "bench_crops_2048.npz"))
    rbuf = M.recognition_model_bytes()
    assert [M.digest(rbuf)] == list(g["model_digests"])
    n = 40
    crops = synth.synthetic_line_crops(1000, n=2048)[:n]
    page = (crops.reshape(1, n * 64, 256) + 0.5).astype(np.float32)
    ora = OP.OcrEngine(recognition_model=OracleModel(OracleGraph(rbuf), "exact"))
    inp = ora.prepare_input(OP.ImageSource.from_tensor(page, "chw"))
    rec = ora.recognizer
    to = g["token_offsets"]
    for c0 in range(0, n, 20):
        batch = np.full((20, 1, 64, 300), -0.5, np.float32)
        for i in range(20):
            line = [RotatedRect.from_array(np.array([128.0, (c0 + i) * 64.0 + 32.0, 0.0, 1.0, 256.0, 64.0], np.float32))]
            poly, rw = rec._line_geometry(line)
            assert rw == 256
            clib.prepare_text_line_into(inp[0], [(p[1], p[0]) for p in poly], rw, 64, batch[i, 0])
        out = rec.run(batch)
        for i in range(20):
            steps = np.array(clib.ctc_greedy(out[i]), np.int32).reshape(-1, 2)
            assert np.array_equal(steps, g["tokens"][to[c0 + i]:to[c0 + i + 1]])


def test_oracle_reproduces_small_odd_page_golden_end_to_end():
    """page_odd_small.npz (97x211 page, make_golden_bench.py odd_small): the whole oracle pipeline again — word rects,
    lines (oracle layout AND the product's host layout), text — so that a drift of the oracle or of the synthetic
    models/pages shows up on the CPU, not only as a GPU-test failure."""
    from ocrs_amd import OcrEngine, synth
    g = np.load(os.path.join(G, "page_odd_small.npz"))
    dbuf, rbuf = M.detection_model_bytes(), M.recognition_model_bytes()
    assert [M.digest(dbuf), M.digest(rbuf)] == list(g["model_digests"]), "synthetic model files changed: regenerate goldens"
    ora = OP.OcrEngine(detection_model=OracleModel(OracleGraph(dbuf), "exact"),
                       recognition_model=OracleModel(OracleGraph(rbuf), "exact"))
    px = synth.synthetic_page(102, 97, 211, lines=3, columns=1)
    inp = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
    words = ora.detect_words(inp)
    assert np.array_equal(np.array([w.to_array() for w in words], np.float32).reshape(-1, 6), g["word_rects"])
    lines = ora.find_text_lines(inp, words)
    flat = np.array([w.to_array() for l in lines for w in l], np.float32).reshape(-1, 6)
    assert np.array_equal(flat, g["line_rects"])
    assert np.array_equal(np.cumsum([0] + [len(l) for l in lines]), g["line_offsets"])
    host_lines = OcrEngine().find_text_lines(None, g["word_rects"])
    assert np.array_equal(np.concatenate(host_lines), g["line_rects"])
    text = ora.recognize_text(inp, lines)
    co = g["char_offsets"]
    for i, t in enumerate(text):
        exp = "".join(chr(c) for c in g["chars"][co[i]:co[i + 1], 0])
        assert (str(t) if t is not None else "") == exp


# ---------------------------------------------------------------- the reference's own images (round 4)
REF_IMAGES = ("why-rust", "polar-bears", "rust-book")


def _ref_fixture(name):
    g = np.load(os.path.join(G, "reference", name + ".npz"))
    dbuf, rbuf = M.detection_model_bytes(ink=tuple(g["ink"])), M.recognition_model_bytes()
    assert [M.digest(dbuf), M.digest(rbuf)] == list(g["model_digests"]), \
        "synthetic model files changed: re-run tests/golden/make_golden_reference_images.py"
    return g, dbuf, rbuf


@pytest.mark.parametrize("name", REF_IMAGES)
def test_oracle_reproduces_the_cheap_stages_of_the_reference_image_goldens(name):
    """Drift guard without the networks: grey conversion of the stored pixels, mask -> contours -> rects (the stage the
    natural glyph masks are there for) and the line grouping, re-derived with the oracle."""
    from oracle import clib
    from oracle.geometry import RotatedRect
    from oracle import layout as OL
    g, _, _ = _ref_fixture(name)
    px = g["pixels"]
    grey = OP.prepare_image(OP.ImageSource.from_tensor(px, "hwc"))
    assert _bits_sum(grey) == int(g["grey_bits_sum"][0])
    h, w = [int(v) for v in g["mask_shape"]]
    assert (h, w) == px.shape[:2]
    mask = np.unpackbits(g["mask"])[: h * w].reshape(h, w).astype(np.uint8)
    rects = clib.component_rects(mask, 3.0, 100.0)
    assert np.array_equal(np.asarray(rects, np.float32).reshape(-1, 6), g["word_rects"])
    n_contours = len(clib.find_contours_external(mask))
    assert n_contours > 2 * len(rects) or name == "polar-bears"      # many specks below min_area: natural masks
    words = [RotatedRect.from_array(r) for r in g["word_rects"]]
    lines = OL.find_text_lines(words)
    assert np.array_equal(np.cumsum([0] + [len(l) for l in lines]), g["line_offsets"])
    assert np.array_equal(np.array([x.to_array() for l in lines for x in l], np.float32).reshape(-1, 6), g["line_rects"])


@pytest.mark.parametrize("name,angle", [("why-rust", 3), ("polar-bears", -10), ("rust-book", 90), ("polar-bears", 90), ("why-rust", -10)])
def test_oracle_reproduces_the_cheap_stages_of_the_rotated_page_goldens(name, angle):
    """tests/golden/rotated/ (make_golden_rotated.py): the rotated pixels are rebuilt with the same PIL call and their CRC
    checked; grey conversion, mask -> contours -> rotated rects and the line grouping re-derived with the oracle."""
    import sys
    import zlib
    from oracle import clib
    from oracle.geometry import RotatedRect
    from oracle import layout as OL
    sys.path.insert(0, G)
    from make_golden_rotated import rotated_pixels
    g = np.load(os.path.join(G, "rotated", "%s_%+d.npz" % (name, angle)))
    base = np.load(os.path.join(G, "reference", name + ".npz"))
    px = rotated_pixels(base["pixels"], name, angle)
    if zlib.crc32(px.tobytes()) != int(g["pixel_crc"][0]):
        pytest.skip("this PIL build resamples differently from the one that made the fixture")
    grey = OP.prepare_image(OP.ImageSource.from_tensor(px, "hwc"))
    assert _bits_sum(grey) == int(g["grey_bits_sum"][0])
    h, w = [int(v) for v in g["mask_shape"]]
    mask = np.unpackbits(g["mask"])[: h * w].reshape(h, w).astype(np.uint8)
    rects = clib.component_rects(mask, 3.0, 100.0)
    assert np.array_equal(np.asarray(rects, np.float32).reshape(-1, 6), g["word_rects"])
    words = [RotatedRect.from_array(r) for r in g["word_rects"]]
    lines = OL.find_text_lines(words)
    assert np.array_equal(np.cumsum([0] + [len(l) for l in lines]), g["line_offsets"])
    assert np.array_equal(np.array([x.to_array() for l in lines for x in l], np.float32).reshape(-1, 6), g["line_rects"])
