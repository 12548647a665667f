"""Pins the CPU oracle against every known-answer test the reference holds for
the hot path.  CPU only."""
import numpy as np
import pytest

from oracle import clib
from oracle.geometry import Rect, RotatedRect, f32
from oracle.layout import filter_overlapping, find_block_separators, find_text_lines, max_empty_rects
from oracle.pipeline import (ImageSource, ImageSourceError, OcrEngine, TextChar, TextLine, line_polygon,
                             prepare_image)

import kat_util as K


class FakeDet:
    def input_shape(self):
        return K.FAKE_DETECTION_SHAPE

    def run(self, x):
        return K.fake_detection_run(x)


class FakeRec:
    def input_shape(self):
        return K.FAKE_RECOGNITION_SHAPE

    def run(self, x):
        return K.fake_recognition_run(x)


# ---------------------------------------------------------------- preprocess.rs:274-360
def test_image_source_from_bytes():
    cases = [(100, 10, 10, None), (50, 10, 10, "length"), (8 * 8 * 2, 8, 8, "channel"), (0, 0, 10, "channel")]
    for ln, w, h, err in cases:
        data = bytes(range(ln % 256)) if ln < 256 else bytes(ln)
        data = bytes((i % 256 for i in range(ln)))
        if err is None:
            ImageSource.from_bytes(data, (w, h))
        else:
            with pytest.raises(ImageSourceError, match=err):
                ImageSource.from_bytes(data, (w, h))


def test_image_source_from_data():
    a = np.arange(25, dtype=np.uint8).reshape(1, 5, 5)
    ImageSource.from_tensor(a, "chw")
    with pytest.raises(ImageSourceError):
        ImageSource.from_tensor(a, "hwc")
    with pytest.raises(ImageSourceError):
        ImageSource.from_tensor(np.zeros((0, 5, 5), np.uint8), "chw")


ITU = [0.299, 0.587, 0.114]


def grey(r, g, b):
    return -0.5 + r * ITU[0] + g * ITU[1] + b * ITU[2]


# ---------------------------------------------------------------- preprocess.rs:379-594
@pytest.mark.parametrize("shape,order", [((2, 2, 1), "hwc"), ((1, 2, 2), "chw")])
def test_prepare_image_greyscale(shape, order):
    u8 = np.array([0, 128, 255, 64], np.uint8).reshape(shape)
    r = prepare_image(ImageSource.from_tensor(u8, order))
    assert r.shape == (1, 2, 2)
    np.testing.assert_allclose(r.ravel(), [-0.5, -0.5 + 128 / 255, 0.5, -0.5 + 64 / 255], atol=1e-5)
    fl = np.array([0.0, 0.5, 1.0, 0.25], np.float32).reshape(shape)
    r = prepare_image(ImageSource.from_tensor(fl, order))
    np.testing.assert_allclose(r.ravel(), [-0.5, 0.0, 0.5, -0.25], atol=1e-5)


@pytest.mark.parametrize("data,shape,order,rgb", [
    ([100, 150, 200], (1, 1, 3), "hwc", (100, 150, 200)),
    ([100, 150, 200], (3, 1, 1), "chw", (100, 150, 200)),
    ([50, 100, 150, 255], (1, 1, 4), "hwc", (50, 100, 150)),
    ([50, 100, 150, 255], (4, 1, 1), "chw", (50, 100, 150)),
])
def test_prepare_image_rgb_rgba_u8(data, shape, order, rgb):
    r = prepare_image(ImageSource.from_tensor(np.array(data, np.uint8).reshape(shape), order))
    assert r.shape == (1, 1, 1)
    assert abs(r[0, 0, 0] - grey(*[v / 255 for v in rgb])) < 1e-5


@pytest.mark.parametrize("shape,order", [((1, 1, 3), "hwc"), ((3, 1, 1), "chw")])
def test_prepare_image_rgb_f32(shape, order):
    r = prepare_image(ImageSource.from_tensor(np.array([0.4, 0.6, 0.8], np.float32).reshape(shape), order))
    assert abs(r[0, 0, 0] - grey(0.4, 0.6, 0.8)) < 1e-5


def test_prepare_image_multi_pixel_rgb():
    hwc = np.array([255, 0, 0, 0, 255, 0, 0, 0, 255, 128, 128, 128], np.uint8).reshape(2, 2, 3)
    chw = np.ascontiguousarray(hwc.transpose(2, 0, 1))
    exp = [grey(1, 0, 0), grey(0, 1, 0), grey(0, 0, 1), grey(128 / 255, 128 / 255, 128 / 255)]
    for arr, order in ((hwc, "hwc"), (chw, "chw")):
        r = prepare_image(ImageSource.from_tensor(arr, order))
        np.testing.assert_allclose(r.ravel(), exp, atol=1e-5)


def test_prepare_image_op_order_bit_exact():
    """preprocess.rs:229-233: start at -0.5, add channel products in order, no FMA."""
    rng = np.random.default_rng(0)
    px = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    w = (np.array(ITU, np.float32) / np.float32(255.0)).astype(np.float32)
    exp = np.full((7, 5), -0.5, np.float32)
    for c in range(3):
        exp = (exp + (px[..., c].astype(np.float32) * w[c]).astype(np.float32)).astype(np.float32)
    r = prepare_image(ImageSource.from_tensor(px, "hwc"))
    assert np.array_equal(r[0], exp)


# ---------------------------------------------------------------- detection.rs:213-246
def test_find_connected_component_rects():
    mask = np.zeros((400, 400), np.uint8)
    rects = K.gen_rect_grid((10, 10), (5, 5), (10, 50), (10, 5))
    for t, l, b, r in rects:
        mask[t:b + 1, l:r + 1] = 1  # adjust_tlbr(0,0,1,1) then fill_rect (exclusive)
    comps = clib.component_rects(mask, 0.0, 100.0)
    assert len(comps) == 25
    for c in comps:
        shape = sorted([int(np.floor(abs(c[5]) + 0.5)), int(np.floor(abs(c[4]) + 0.5))])
        assert shape == [10, 50]


# ---------------------------------------------------------------- lib.rs:447-488
def test_ocr_engine_prepare_input():
    img = K.gen_test_image(3)
    eng = OcrEngine(detection_model=FakeDet())
    inp = eng.prepare_input(ImageSource.from_tensor(img, "chw"))
    assert inp.shape == (1, 100, 200)


def test_ocr_engine_detect_words():
    img = K.gen_test_image(3)
    eng = OcrEngine(detection_model=FakeDet())
    inp = eng.prepare_input(ImageSource.from_tensor(img, "chw"))
    words = eng.detect_words(inp)
    assert len(words) == 3
    boxes = sorted((w.bounding_rect().tlhw() for w in words), key=lambda b: (int(b[0]), int(b[1])))
    assert boxes == K.EXPECTED_WORD_BOXES_TLHW  # exact f32 equality, as assert_eq! in the reference


def test_engine_errors_without_models():
    eng = OcrEngine()
    inp = np.zeros((1, 4, 4), np.float32)
    with pytest.raises(RuntimeError, match="Detection model not loaded"):
        eng.detect_words(inp)
    with pytest.raises(RuntimeError, match="Recognition model not loaded"):
        eng.recognize_text(inp, [])
    assert eng.detection_threshold() == np.float32(0.2)


# ---------------------------------------------------------------- lib.rs:501-577
def _recognize(engine, image):
    inp = engine.prepare_input(ImageSource.from_tensor(image, "chw"))
    line = [RotatedRect.from_rect(Rect.from_tlhw(0, 0, image.shape[1], image.shape[2]))]
    lines = engine.recognize_text(inp, [line])
    assert len(lines) == 1 and lines[0] is not None
    return str(lines[0])


def test_ocr_engine_recognize_lines():
    image = np.zeros((1, 64, 32), np.float32)
    image[:, 2, :] = 1.0
    eng = OcrEngine(recognition_model=FakeRec(), alphabet=K.make_alphabet())
    assert _recognize(eng, image) == "0"


def test_ocr_engine_filter_chars():
    image = np.zeros((1, 64, 32), np.float32)
    image[:, 2, :] = 0.7
    image[:, 3, :] = 0.3
    eng = OcrEngine(recognition_model=FakeRec(), alphabet=K.make_alphabet())
    assert _recognize(eng, image) == "0"
    eng = OcrEngine(recognition_model=FakeRec(), alphabet=K.make_alphabet(), allowed_chars="123456789")
    assert _recognize(eng, image) == "1"


# ---------------------------------------------------------------- recognition.rs:571-595
def test_line_polygon():
    words = []
    for i in range(5):
        up = (f32(0.0), f32(-1.0 if i % 2 == 0 else 1.0))
        words.append(RotatedRect.new((f32(i * 20.0), f32(10.0)), up, f32(10.0), f32(5.0)))
    poly = line_polygon(words)
    assert len(poly) == 20
    # contains each word centre (even-odd fill over the polygon bounding box)
    ys = [p[1] for p in poly]
    xs = [p[0] for p in poly]
    top, left = min(ys), min(xs)
    m = clib.polygon_fill_mask([(p[1], p[0]) for p in poly], top, left, max(ys) - top, max(xs) - left)
    for w in words:
        c = w.bounding_rect().center()
        y, x = int(np.floor(float(c[1]) + 0.5)), int(np.floor(float(c[0]) + 0.5))
        if y - top < m.shape[0] and x - left < m.shape[1]:
            assert m[y - top, x - left] == 1


# ---------------------------------------------------------------- text_items.rs:131-186
def _gen_text_chars(text, width):
    return [TextChar(ch, Rect.from_tlhw(0, i * width, 25, width)) for i, ch in enumerate(text)]


def test_item_display_and_words():
    assert str(TextLine(_gen_text_chars("foo bar baz", 10))) == "foo bar baz"
    words = TextLine(_gen_text_chars("foo bar  baz ", 10)).words()
    assert ["".join(c.char for c in w) for w in words] == ["foo", "bar", "baz"]
    assert words[1][0].rect == Rect.from_tlhw(0, 40, 25, 10)


def test_rotated_rect_corners_order():
    """text_items.rs:156-166."""
    r = RotatedRect.new((f32(15.0), f32(12.5)), (f32(0.0), f32(-1.0)), f32(30.0), f32(25.0))
    got = [(float(c[1]), float(c[0])) for c in r.corners()]
    assert got == [(25.0, 30.0), (25.0, 0.0), (0.0, 0.0), (0.0, 30.0)]


# ---------------------------------------------------------------- empty_rects.rs:239-294
def _two_columns():
    left = K.gen_rect_grid((0, 0), (10, 5), (5, 5), (3, 2))
    lb = K.union_rects(left)
    right = K.gen_rect_grid((0, lb[3] + 20), (10, 5), (5, 5), (3, 2))
    rb = K.union_rects(right)
    return left, lb, right, rb


def test_max_empty_rects():
    page = Rect.from_tlbr(0, 0, 80, 90)
    left, lb, right, rb = _two_columns()
    obstacles = [Rect(*r) for r in left + right]
    first = next(max_empty_rects(obstacles, page, lambda r: np.float32(r.area()), 0, 0))
    assert first == Rect.from_tlbr(page.top, lb[3], page.bottom, rb[1])


def test_max_empty_rects_if_none():
    b = Rect.from_tlbr(0, 0, 5, 5)
    assert next(max_empty_rects([b], b, lambda r: np.float32(r.area()), 0, 0), None) is None
    assert next(max_empty_rects([], Rect.from_hw(0, 0), lambda r: np.float32(r.area()), 0, 0), None) is None


# ---------------------------------------------------------------- layout_analysis.rs:243-350
def test_find_block_separators():
    words = [RotatedRect.from_rect(Rect(*r)) for r in K.gen_rect_grid((0, 0), (2, 2), (10, 20), (50, -5))]
    assert len(find_block_separators(words)) == 2


def test_find_text_lines():
    left, lb, right, rb = _two_columns()
    words = [RotatedRect.from_rect(Rect(*r)) for r in left + right]
    words = K.xorshift_shuffle(words, 1234)
    lines = find_text_lines(words)
    assert len(lines) == 20
    for line in lines:
        assert len(line) == 5
        br = None
        for r in line:
            br = r.bounding_rect() if br is None else br.union(r.bounding_rect())
        assert abs(float(br.height()) - 5) <= 1.0
        assert abs(float(br.width()) - (5 * (5 + 2) - 2)) <= 1.0


# ---------------------------------------------------------------- BASELINE.json configs[0]: CPU plumbing baseline
def test_config0_single_800x600_image_cpu_path():
    """One 600w x 800h page through all four OcrEngine calls on the CPU path
    (oracle with torch-CPU networks, synthetic weights): plumbing check, no GPU."""
    import models_util as M
    from ocrs_amd import synth
    from oracle.nn import OracleGraph, OracleModel
    px = synth.synthetic_page(0, 800, 600, lines=30, columns=1)
    eng = OcrEngine(detection_model=OracleModel(OracleGraph(M.detection_model_bytes()), "torch"),
                    recognition_model=OracleModel(OracleGraph(M.recognition_model_bytes()), "torch"))
    inp = eng.prepare_input(ImageSource.from_tensor(px, "hwc"))
    assert inp.shape == (1, 800, 600)
    words = eng.detect_words(inp)
    lines = eng.find_text_lines(inp, words)
    texts = eng.recognize_text(inp, lines[:4])
    assert len(words) > 100 and len(lines) >= 25
    assert sum(1 for t in texts if t is not None) >= 3
