"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
the same seeded inputs.  Integer / byte / index results must match bit for bit;
fp32 model outputs must match bit for bit too (both sides evaluate the
canonical fmaf-chain order of DESIGN.md §4) and additionally stay within 1e-4
(probabilities) / 1e-3 (log-probs) of the PyTorch-CPU fp32 evaluation.

Run with:  python -m pytest tests -m gpu
"""
import ctypes as C

import numpy as np
import pytest

import kat_util as K
import models_util as M
import ocrs_amd
from ocrs_amd import DimOrder, ImageSource, Model, OcrEngine, _lib, synth
from oracle import clib
from oracle import pipeline as OP
from oracle.geometry import Rect, RotatedRect
from oracle.nn import OracleGraph, OracleModel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from ocrs_amd import _lib
    _lib.require_gpu()  # fail loudly: there is no CPU fallback to "pass" on


def rects_of(words):
    return np.array([w.to_array() for w in words], np.float32).reshape(-1, 6)


# ------------------------------------------------------------------ stage 0
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("order", ["hwc", "chw"])
@pytest.mark.parametrize("chans", [1, 3, 4])
def test_prepare_input_bit_exact(dtype, order, chans):
    rng = np.random.default_rng(chans * 10 + (order == "hwc"))
    h, w = 37, 53
    shape = (h, w, chans) if order == "hwc" else (chans, h, w)
    px = rng.integers(0, 256, shape).astype(np.uint8) if dtype == np.uint8 else rng.random(shape, dtype=np.float32)
    eng = OcrEngine()
    got = eng.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc if order == "hwc" else DimOrder.Chw)).image()
    exp = OP.prepare_image(OP.ImageSource.from_tensor(px, order))
    assert got.shape == exp.shape and np.array_equal(got, exp)


def test_prepare_input_rgb8_fast_path_full_page():
    px = synth.synthetic_page(3)
    got = OcrEngine().prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc)).image()
    assert np.array_equal(got, OP.prepare_image(OP.ImageSource.from_tensor(px, "hwc")))


# ------------------------------------------------------------------ reference KATs through the engine
def test_prepare_input_batch_from_pinned_host_memory_equals_single_calls():
    """ocrs_engine_prepare_input_batch (several host images, one wait) from page-locked buffers (ocrs_host_malloc) and
    from ordinary arrays gives the same grey pages as ocrs_engine_prepare_input image by image (= the oracle's, tested
    above), for u8 HWC and f32 CHW."""
    L = _lib.lib()
    eng = OcrEngine()
    rng = np.random.default_rng(11)
    for dtype, order, shape in ((np.uint8, DimOrder.Hwc, (37, 53, 3)), (np.float32, DimOrder.Chw, (3, 37, 53))):
        imgs = [(rng.integers(0, 256, shape).astype(np.uint8) if dtype == np.uint8 else rng.random(shape, np.float32))
                for _ in range(5)]
        ref = [eng.prepare_input(ImageSource.from_tensor(im, order)).image() for im in imgs]
        pinned = []
        for im in imgs:
            hp = C.c_void_p()
            _lib.check(L.ocrs_host_malloc(C.c_size_t(im.nbytes), C.byref(hp)))
            C.memmove(hp, im.ctypes.data_as(C.c_void_p), im.nbytes)
            pinned.append(hp)
        h, w = (shape[0], shape[1]) if order == DimOrder.Hwc else (shape[1], shape[2])
        for ptrs in ([p.value for p in pinned], [im.ctypes.data for im in imgs]):
            got = eng.prepare_input_batch_raw(ptrs, dtype, order, h, w, 3)
            assert len(got) == 5
            for g, r in zip(got, ref):
                assert np.array_equal(g.image(), r)
        for hp in pinned:
            _lib.check(L.ocrs_host_free(hp))
    assert eng.prepare_input_batch_raw([], np.uint8, DimOrder.Hwc, 4, 4, 3) == []


def test_kat_detect_words_fake_model():  # lib.rs:466-488
    det = Model.from_callable(K.FAKE_DETECTION_SHAPE, K.fake_detection_run)
    eng = OcrEngine(detection_model=det)
    inp = eng.prepare_input(ImageSource.from_tensor(K.gen_test_image(3), DimOrder.Chw))
    assert inp.shape == (1, 100, 200)
    words = eng.detect_words(inp)
    assert len(words) == 3
    boxes = sorted((RotatedRect.from_array(w).bounding_rect().tlhw() for w in words), key=lambda b: (int(b[0]), int(b[1])))
    assert boxes == K.EXPECTED_WORD_BOXES_TLHW


def _recognize(engine, image):
    inp = engine.prepare_input(ImageSource.from_tensor(image, DimOrder.Chw))
    line = rects_of([RotatedRect.from_rect(Rect.from_tlhw(0, 0, image.shape[1], image.shape[2]))])
    lines = engine.recognize_text(inp, [line])
    assert len(lines) == 1 and lines[0] is not None
    return str(lines[0])


def test_kat_recognize_lines_fake_model():  # lib.rs:527-577
    rec = Model.from_callable(K.FAKE_RECOGNITION_SHAPE, K.fake_recognition_run)
    image = np.zeros((1, 64, 32), np.float32)
    image[:, 2, :] = 1.0
    assert _recognize(OcrEngine(recognition_model=rec, alphabet=K.make_alphabet()), image) == "0"
    image[:, 2, :] = 0.7
    image[:, 3, :] = 0.3
    assert _recognize(OcrEngine(recognition_model=rec, alphabet=K.make_alphabet()), image) == "0"
    assert _recognize(OcrEngine(recognition_model=rec, alphabet=K.make_alphabet(), allowed_chars="123456789"), image) == "1"


def test_engine_errors():  # lib.rs:197,211,254,274 ; detection.rs:141-144 ; recognition.rs:487-493
    eng = OcrEngine()
    inp = eng.prepare_input(ImageSource.from_tensor(np.zeros((1, 8, 8), np.float32), DimOrder.Chw))
    with pytest.raises(ocrs_amd.OcrsError, match="Detection model not loaded"):
        eng.detect_words(inp)
    with pytest.raises(ocrs_amd.OcrsError, match="Recognition model not loaded"):
        eng.recognize_text(inp, [])
    assert abs(eng.detection_threshold() - 0.2) < 1e-7
    sym = Model.from_callable([None, 1, None, None], K.fake_detection_run)
    with pytest.raises(ocrs_amd.OcrsError, match="failed to get model dims"):
        OcrEngine(detection_model=sym).detect_words(inp)
    rec = Model.from_callable(K.FAKE_RECOGNITION_SHAPE, K.fake_recognition_run)
    image = np.zeros((1, 64, 32), np.float32)
    with pytest.raises(ocrs_amd.OcrsError, match="does not match alphabet size"):
        _recognize(OcrEngine(recognition_model=rec), image)  # 96-char default alphabet vs 64 columns
    bad = Model.from_callable(K.FAKE_RECOGNITION_SHAPE, lambda x: np.zeros((3, 3), np.float32))
    with pytest.raises(ocrs_amd.OcrsError, match="expected recognition output to have 3 dims but it has 2"):
        _recognize(OcrEngine(recognition_model=bad, alphabet=K.make_alphabet()), image)

    def boom(x):
        raise RuntimeError("boom")

    with pytest.raises(ocrs_amd.OcrsError, match="model run failed"):
        _recognize(OcrEngine(recognition_model=Model.from_callable(K.FAKE_RECOGNITION_SHAPE, boom),
                             alphabet=K.make_alphabet()), image)


# ------------------------------------------------------------------ Model::run parity
def test_mfma_chain_is_bitwise_fmaf_chain():
    """The premise of the numeric spec: fp32 MFMA == k-ordered fmaf chain."""
    from ocrs_amd import modelfile as mf
    rng = np.random.default_rng(5)
    for cin, cout in [(8, 8), (16, 32), (64, 64), (128, 96), (256, 40)]:
        w = (rng.standard_normal((1, 1, cin, cout)) * 0.3).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        lift_w = rng.standard_normal((1, 1, 1, cin)).astype(np.float32)
        ops = [mf.Op(mf.OP_CONV, 0, 1, kh=1, kw=1, cin=1, cout=cin, weights=(lift_w, np.zeros(cin, np.float32))),
               mf.Op(mf.OP_CONV, 1, 2, relu=0, kh=1, kw=1, cin=cin, cout=cout, weights=(w, b))]
        # output channels > 1 come back NCHW; compare against the oracle's NHWC result
        g = mf.Graph(mf.KIND_DETECTION, [-1, 1, -1, -1], ops, 3, 2)
        x = rng.standard_normal((2, 1, 13, 11)).astype(np.float32)
        got = Model.load_bytes(g.to_bytes()).run(x)
        exp = OracleGraph(g.to_bytes()).run_exact(x)
        assert got.shape == exp.shape and np.array_equal(got, exp), (cin, cout)


@pytest.mark.parametrize("in_hw,n", [((96, 64), 3), ((800, 600), 1)])
def test_detection_model_run_bit_exact(in_hw, n):
    buf = M.detection_model_bytes(in_hw) if in_hw == (800, 600) else M.detection_model_bytes(in_hw, (8, 16, 32, 32))
    rng = np.random.default_rng(11)
    x = (rng.random((n, 1) + in_hw, dtype=np.float32) - 0.5).astype(np.float32)
    m = Model.load_bytes(buf)
    assert m.input_shape() == [None, 1, in_hw[0], in_hw[1]]
    got = m.run(x)
    og = OracleGraph(buf)
    exp = og.run_exact(x)
    assert got.shape == exp.shape == (n, 1) + in_hw
    assert np.array_equal(got, exp)
    assert np.abs(got - og.run_torch(x)).max() < 1e-4  # fp32 tolerance vs the ONNX-operator semantics


def test_onnx_files_load_and_run_bit_exact(tmp_path):
    """`.onnx` files go through ocrs_amd.onnx_import into the HIP executor."""
    from ocrs_amd import modelfile as mf
    from ocrs_amd.onnx_export import export_onnx
    det = mf.build_detection(in_hw=(100, 76), depths=(8, 16, 16, 32), seed=5)
    rec = mf.build_recognition(n_classes=31, in_h=64, seed=6, hidden=64, chans=(32, 32, 64, 64, 64, 64))
    rng = np.random.default_rng(3)
    for g, x in ((det, rng.uniform(-0.5, 0.5, (2, 1, 100, 76))), (rec, rng.uniform(-0.5, 0.5, (3, 1, 64, 120)))):
        path = tmp_path / ("m%d.onnx" % g.kind)
        path.write_bytes(export_onnx(g))
        got = Model.load_file(str(path)).run(x.astype(np.float32))
        assert np.array_equal(got, OracleGraph(g.to_bytes()).run_exact(x.astype(np.float32)))


@pytest.mark.parametrize("n,width", [(3, 100), (5, 300), (1, 50)])
def test_recognition_model_run_bit_exact(n, width):
    buf = M.recognition_model_bytes()
    crops = synth.synthetic_line_crops(1000 + n, n=n, width=min(width, 256))
    x = np.full((n, 1, 64, width), -0.5, np.float32)
    x[:, 0, :, :crops.shape[2]] = crops[:, :, :width]
    m = Model.load_bytes(buf)
    assert m.input_shape() == [None, 1, 64, None]
    got = m.run(x)
    og = OracleGraph(buf)
    exp = og.run_exact(x)
    assert got.shape == exp.shape == (width // 4, n, 97)
    assert np.array_equal(got, exp)
    assert np.abs(got - og.run_torch(x)).max() < 1e-3


# ------------------------------------------------------------------ detection post-processing
def _mask_engine_pair(h, w):
    """Engines whose detection 'model' returns a chosen probability map, with model
    dims == page dims, so that only threshold + components + rects are exercised."""
    box = {}

    def run(x):
        return box["prob"].reshape(1, 1, h, w)

    gpu = OcrEngine(detection_model=Model.from_callable([None, 1, h, w], run))

    class Fake:
        def input_shape(self):
            return [None, 1, h, w]

        def run(self, x):
            return box["prob"].reshape(1, 1, h, w)

    return box, gpu, OP.OcrEngine(detection_model=Fake())


def _adversarial_masks(h, w):
    rng = np.random.default_rng(42)
    yield "empty", np.zeros((h, w), np.uint8)
    yield "full", np.ones((h, w), np.uint8)
    m = np.zeros((h, w), np.uint8)
    m[10:60, 10:100] = 1
    m[20:50, 20:90] = 0       # ring
    m[28:42, 30:80] = 1       # island in the hole: NOT external
    m[32:38, 40:70] = 0
    m[34:36, 50:60] = 1       # island in the island's hole
    yield "nested rings", m
    m = np.zeros((h, w), np.uint8)
    m[5, 5:60] = 1            # 1-px lines, single pixels, diagonals
    m[10:70, 7] = 1
    m[80, 80] = 1
    for i in range(40):
        m[20 + i, 100 + i] = 1
        m[20 + i, 160 - i] = 1
    m[0, :] = 1               # touching the frame
    m[:, w - 1] = 1
    yield "thin", m
    m = (rng.random((h, w)) < 0.35).astype(np.uint8)
    yield "noise35", m
    m = (rng.random((h, w)) < 0.6).astype(np.uint8)
    yield "noise60", m
    m = np.zeros((h, w), np.uint8)
    m[::2, ::2] = 1           # isolated pixels
    yield "dots", m
    m = np.zeros((h, w), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    m[((yy // 7 + xx // 9) % 2) == 0] = 1  # checkerboard of blocks: diagonal 8-connections
    yield "checker", m
    m = np.zeros((h, w), np.uint8)
    for k in range(12):       # rotated blobs
        cy, cx = rng.integers(15, h - 15), rng.integers(30, w - 30)
        ang = rng.random() * np.pi
        d = np.abs((yy - cy) * np.cos(ang) - (xx - cx) * np.sin(ang)) < 5
        e = np.abs((yy - cy) * np.sin(ang) + (xx - cx) * np.cos(ang)) < 22
        m[d & e] = 1
    yield "rotated", m


def test_component_rects_bit_exact_on_adversarial_masks():
    h, w = 120, 200
    box, gpu, ora = _mask_engine_pair(h, w)
    page = np.zeros((1, h, w), np.float32)
    inp = gpu.prepare_input(ImageSource.from_tensor(page, DimOrder.Chw))
    for name, mask in _adversarial_masks(h, w):
        box["prob"] = mask.astype(np.float32)
        got = gpu.detect_words(inp)
        exp = rects_of(ora.detect_words(page))
        assert got.shape == exp.shape, name
        assert np.array_equal(got, exp), name
        # and the low-level pieces the oracle is built from agree on contour discovery order
        if name == "nested rings":
            assert len(clib.find_contours_external(mask)) == 1


def test_detect_text_pixels_and_words_bit_exact_on_synthetic_pages():
    buf = M.detection_model_bytes()
    gpu = OcrEngine(detection_model=Model.load_bytes(buf))
    ora = OP.OcrEngine(detection_model=OracleModel(OracleGraph(buf), "exact"))
    for seed, hw in [(0, (1024, 1024)), (1, (700, 500)), (2, (1300, 900))]:
        px = synth.synthetic_page(seed, hw[0], hw[1], lines=60)
        inp = gpu.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
        oin = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
        assert np.array_equal(gpu.detect_text_pixels(inp), ora.detect_text_pixels(oin)), hw
        got = gpu.detect_words(inp)
        exp = rects_of(ora.detect_words(oin))
        assert len(exp) > 50
        assert got.shape == exp.shape and np.array_equal(got, exp), hw


def test_detect_words_batch_equals_single():
    buf = M.detection_model_bytes()
    gpu = OcrEngine(detection_model=Model.load_bytes(buf))
    inps = [gpu.prepare_input(ImageSource.from_tensor(synth.synthetic_page(s, 512, 512, lines=30), DimOrder.Hwc)) for s in range(3)]
    batch = gpu.detect_words_batch(inps)
    for i, inp in enumerate(inps):
        assert np.array_equal(batch[i], gpu.detect_words(inp))


# ------------------------------------------------------------------ recognition
def test_prepare_recognition_input_bit_exact():
    rec = Model.load_bytes(M.recognition_model_bytes())
    gpu = OcrEngine(recognition_model=rec)
    ora = OP.OcrEngine(recognition_model=OracleModel(OracleGraph(M.recognition_model_bytes()), "exact"))
    px = synth.synthetic_page(4, 400, 600, lines=20)
    inp = gpu.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    oin = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
    rng = np.random.default_rng(9)
    for _ in range(6):
        words, x = [], 30 + int(rng.integers(0, 100))
        y = int(rng.integers(20, 330))
        for _ in range(int(rng.integers(1, 6))):
            ww, hh = int(rng.integers(20, 70)), int(rng.integers(10, 26))
            ang = float(rng.normal(0, 0.06))
            up = (np.float32(np.sin(ang)), np.float32(np.cos(ang)))
            words.append(RotatedRect.new((np.float32(x + ww / 2), np.float32(y + hh / 2 + rng.normal(0, 1.5))), up,
                                         np.float32(ww), np.float32(hh)))
            x += ww + int(rng.integers(4, 12))
        got = gpu.prepare_recognition_input(inp, rects_of(words))
        exp = ora.prepare_recognition_input(oin, words)
        assert got.shape == exp.shape and np.array_equal(got, exp)
    # a line hanging over the page edge (bounds checks of recognition.rs:100,112)
    edge = [RotatedRect.new((np.float32(590.0), np.float32(395.0)), (np.float32(0.0), np.float32(1.0)), np.float32(60.0), np.float32(20.0))]
    assert np.array_equal(gpu.prepare_recognition_input(inp, rects_of(edge)), ora.prepare_recognition_input(oin, edge))


@pytest.mark.parametrize("in_h,chans", [(32, (32, 64, 64, 64, 64, 64)), (16, (32, 64, 64, 64, 64, 64)),
                                         (64, (32, 64, 128, 128, 128, 128))])
def test_recognition_other_input_heights_and_ragged_widths(in_h, chans):
    """The engine's ragged recognition path on models of other input heights: 64 -> 8x16 patches at
    H = 32/16/8, 32 -> 4x32 patches at H = 4, 16 -> H = 2 is not patch-tileable and takes the per-group
    path.  Lines of many different widths (one width group each), tokens and boxes identical."""
    from ocrs_amd import modelfile as mf
    g = mf.build_recognition(n_classes=97, in_h=in_h, seed=21, hidden=64, chans=chans)
    cal = synth.synthetic_line_crops(9, n=8)[:, ::64 // in_h, ::64 // in_h]
    xp = np.full((8, 1, in_h, 300), -0.5, np.float32)
    xp[:, 0, :, :cal.shape[2]] = cal
    g = mf.calibrate_recognition_head(g, lambda buf, x: OracleGraph(buf).run_torch(x), xp)
    rbuf = g.to_bytes()
    gpu = OcrEngine(recognition_model=Model.load_bytes(rbuf))
    ora = OP.OcrEngine(recognition_model=OracleModel(OracleGraph(rbuf), "exact"))
    px = synth.synthetic_page(31, 700, 900, lines=20)
    inp = gpu.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    oin = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
    rng = np.random.default_rng(in_h)
    lines = []
    for i in range(26):  # widths 30 .. 880 px at heights 12 .. 30 -> resized widths across many groups
        ww, hh = 30 + 34 * i, int(rng.integers(12, 31))
        y = 20 + 25 * i
        lines.append([RotatedRect.new((np.float32(10 + ww / 2), np.float32(y)), (np.float32(0.0), np.float32(1.0)),
                                      np.float32(ww), np.float32(hh))])
    got = gpu.recognize_text(inp, [rects_of(l) for l in lines])
    exp = ora.recognize_text(oin, lines)
    assert len(got) == len(exp) == 26
    n_chars = 0
    for a, b in zip(got, exp):
        assert (a is None) == (b is None)
        if a is not None:
            assert str(a) == str(b)
            assert [c.rect for c in a.chars()] == [c.rect.tlbr() for c in b.chars]
            n_chars += len(b.chars)
    assert n_chars > 20


def test_full_pipeline_tokens_boxes_and_text_identical():
    dbuf, rbuf = M.detection_model_bytes(), M.recognition_model_bytes()
    gpu = OcrEngine(detection_model=Model.load_bytes(dbuf), recognition_model=Model.load_bytes(rbuf))
    ora = OP.OcrEngine(detection_model=OracleModel(OracleGraph(dbuf), "exact"),
                       recognition_model=OracleModel(OracleGraph(rbuf), "exact"))
    px = synth.synthetic_page(5, 640, 768, lines=24)
    inp = gpu.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    oin = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
    words = gpu.detect_words(inp)
    owords = ora.detect_words(oin)
    assert np.array_equal(words, rects_of(owords))
    lines = gpu.find_text_lines(inp, words)
    olines = ora.find_text_lines(oin, owords)
    assert len(lines) == len(olines) > 10
    for a, b in zip(lines, olines):
        assert np.array_equal(a, rects_of(b))
    got = gpu.recognize_text(inp, lines)
    exp = ora.recognize_text(oin, olines)
    assert len(got) == len(exp)
    n_chars = 0
    for g, e in zip(got, exp):
        assert (g is None) == (e is None)
        if g is None:
            continue
        assert str(g) == str(e)
        assert [c.rect for c in g.chars()] == [c.rect.tlbr() for c in e.chars]
        n_chars += len(e.chars)
    assert n_chars > 100
    assert gpu.get_text(inp) == ora.get_text(oin)
    # allowed_chars masking (recognition.rs:547-561) on the real graph
    gpu_d = OcrEngine(recognition_model=Model.load_bytes(rbuf), allowed_chars="0123456789")
    ora_d = OP.OcrEngine(recognition_model=OracleModel(OracleGraph(rbuf), "exact"), allowed_chars="0123456789")
    gd = gpu_d.recognize_text(gpu_d.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc)), lines[:6])
    ed = ora_d.recognize_text(oin, olines[:6])
    assert [str(x) if x else None for x in gd] == [str(x) if x else None for x in ed]
    assert all(ch in "0123456789" for x in gd if x for ch in str(x))


# ------------------------------------------------------------------ DecodeMethod::BeamSearch (recognition.rs:198-205,512-514)
def test_beam_search_tokens_match_oracle():
    rbuf = M.recognition_model_bytes()
    from ocrs_amd import DecodeMethod
    px = synth.synthetic_page(6, 160, 420, lines=5, columns=1)
    words = []
    for i in range(5):  # five short lines: [y, x, h, w]
        words.append([RotatedRect.new((np.float32(40 + 70 * i + 35), np.float32(20 + 28 * i)), (np.float32(0.0), np.float32(1.0)),
                                      np.float32(70.0), np.float32(18.0))])
    lines = [rects_of(w) for w in words]
    for width in (1, 7):
        gpu = OcrEngine(recognition_model=Model.load_bytes(rbuf), decode_method=DecodeMethod.BeamSearch(width))
        ora = OP.OcrEngine(recognition_model=OracleModel(OracleGraph(rbuf), "exact"), decode_method=("beam", width))
        inp = gpu.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
        oin = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
        got = gpu.recognize_text(inp, lines)
        exp = ora.recognize_text(oin, words)
        assert [str(x) if x else None for x in got] == [str(x) if x else None for x in exp], width
        for g, e in zip(got, exp):
            if g is not None:
                assert [c.rect for c in g.chars()] == [c.rect.tlbr() for c in e.chars]
    # masked beam search
    gpu = OcrEngine(recognition_model=Model.load_bytes(rbuf), decode_method=DecodeMethod.BeamSearch(4), allowed_chars="abcdefghij")
    ora = OP.OcrEngine(recognition_model=OracleModel(OracleGraph(rbuf), "exact"), decode_method=("beam", 4), allowed_chars="abcdefghij")
    got = gpu.recognize_text(gpu.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc)), lines)
    exp = ora.recognize_text(ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc")), words)
    assert [str(x) if x else None for x in got] == [str(x) if x else None for x in exp]


def test_beam_search_with_callback_model():
    rec = Model.from_callable(K.FAKE_RECOGNITION_SHAPE, K.fake_recognition_run)
    from ocrs_amd import DecodeMethod
    image = np.zeros((1, 64, 32), np.float32)
    image[:, 2, :] = 1.0
    got = _recognize(OcrEngine(recognition_model=rec, alphabet=K.make_alphabet(), decode_method=DecodeMethod.BeamSearch(10)), image)

    class FakeRec:
        def input_shape(self):
            return K.FAKE_RECOGNITION_SHAPE

        def run(self, x):
            return K.fake_recognition_run(x)

    ora = OP.OcrEngine(recognition_model=FakeRec(), alphabet=K.make_alphabet(), decode_method=("beam", 10))
    oin = ora.prepare_input(OP.ImageSource.from_tensor(image, "chw"))
    exp = ora.recognize_text(oin, [[RotatedRect.from_rect(Rect.from_tlhw(0, 0, 64, 32))]])
    assert got == str(exp[0])


# ------------------------------------------------------------------ edge cases
def test_edge_pages_match_oracle():
    dbuf, rbuf = M.detection_model_bytes((160, 128), (8, 16, 32, 32)), M.recognition_model_bytes()
    gpu = OcrEngine(detection_model=Model.load_bytes(dbuf), recognition_model=Model.load_bytes(rbuf))
    ora = OP.OcrEngine(detection_model=OracleModel(OracleGraph(dbuf), "exact"),
                       recognition_model=OracleModel(OracleGraph(rbuf), "exact"))
    cases = {
        "blank": np.full((90, 130, 3), 255, np.uint8),
        "black": np.zeros((90, 130, 3), np.uint8),
        "tiny": np.full((7, 9, 1), 128, np.uint8),
        "one_word": np.full((60, 200, 4), 255, np.uint8),
        "tall": np.full((400, 40, 3), 255, np.uint8),
    }
    cases["one_word"][20:38, 30:120, :3] = 10
    cases["tall"][50:350:30, 5:35] = 0
    for name, px in cases.items():
        inp = gpu.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
        oin = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
        assert np.array_equal(gpu.detect_text_pixels(inp), ora.detect_text_pixels(oin)), name
        words = gpu.detect_words(inp)
        owords = ora.detect_words(oin)
        assert np.array_equal(words, rects_of(owords)), name
        lines = gpu.find_text_lines(inp, words)
        olines = ora.find_text_lines(oin, owords)
        assert len(lines) == len(olines), name
        got = gpu.recognize_text(inp, lines)
        exp = ora.recognize_text(oin, olines)
        assert [str(x) if x else None for x in got] == [str(x) if x else None for x in exp], name
        assert gpu.get_text(inp) == ora.get_text(oin), name
    # empty inputs
    inp = gpu.prepare_input(ImageSource.from_tensor(cases["blank"], DimOrder.Hwc))
    assert gpu.recognize_text(inp, []) == []
    assert gpu.find_text_lines(inp, np.zeros((0, 6), np.float32)) == []
    assert gpu.detect_words_batch([]) == []


def test_thread_safety_concurrent_calls():
    """Model::run is called from several threads at once in the reference (recognition.rs:465-485)."""
    from concurrent.futures import ThreadPoolExecutor
    dbuf = M.detection_model_bytes((160, 128), (8, 16, 32, 32))
    gpu = OcrEngine(detection_model=Model.load_bytes(dbuf))
    pages = [synth.synthetic_page(s, 150, 220, lines=6, columns=1) for s in range(6)]
    inps = [gpu.prepare_input(ImageSource.from_tensor(p, DimOrder.Hwc)) for p in pages]
    ref = [gpu.detect_words(i) for i in inps]
    with ThreadPoolExecutor(6) as ex:
        for _ in range(3):
            got = list(ex.map(gpu.detect_words, inps))
            for a, b in zip(got, ref):
                assert np.array_equal(a, b)


def test_cli_driver_text_json_and_debug_dumps(tmp_path, monkeypatch, capsys):
    """BASELINE configs[0] plumbing (ocrs-cli/src/main.rs:366-497): image file -> text / JSON through the four API
    calls, plus the --text-map / --text-mask / --text-line-images dumps; everything equals the oracle pipeline."""
    import json
    from PIL import Image
    from ocrs_amd import cli, output
    dbuf, rbuf = M.detection_model_bytes(), M.recognition_model_bytes()
    (tmp_path / "det.ocrsm").write_bytes(dbuf)
    (tmp_path / "rec.ocrsm").write_bytes(rbuf)
    px = synth.synthetic_page(8, 600, 800, lines=16)   # 800 x 600 image as in configs[0]
    Image.fromarray(px, "RGB").save(tmp_path / "page.png")
    monkeypatch.chdir(tmp_path)
    common = ["--detect-model", "det.ocrsm", "--rec-model", "rec.ocrsm"]

    ora = OP.OcrEngine(detection_model=OracleModel(OracleGraph(dbuf), "exact"),
                       recognition_model=OracleModel(OracleGraph(rbuf), "exact"))
    oin = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
    owords = ora.detect_words(oin)
    olines = ora.find_text_lines(oin, owords)
    otexts = ora.recognize_text(oin, olines)

    assert cli.main(["page.png", "--text-map", "--text-mask", "--text-line-images"] + common) == 0
    text = capsys.readouterr().out
    assert text.rstrip("\n") == "\n".join(str(t) for t in otexts if t is not None)  # output.rs:88-95
    tm = ora.detect_text_pixels(oin)
    exp_map = (np.clip(tm, 0, 1) * np.float32(255.0)).astype(np.uint8)
    assert np.array_equal(np.asarray(Image.open("text-map.png")), exp_map)
    assert np.array_equal(np.asarray(Image.open("text-mask.png")), (tm > np.float32(0.2)).astype(np.uint8) * 255)
    assert len(list((tmp_path / "lines").glob("line-*.png"))) == len(olines) > 5
    l0 = ora.prepare_recognition_input(oin, olines[0]) + np.float32(0.5)
    assert np.array_equal(np.asarray(Image.open("lines/line-0.png")),
                          (np.clip(l0, 0, 1) * np.float32(255.0)).astype(np.uint8).reshape(l0.shape[-2], l0.shape[-1]))

    assert cli.main(["page.png", "--json", "-o", "out.json"] + common) == 0
    doc = json.loads((tmp_path / "out.json").read_text())
    assert doc["url"] == "page.png" and (doc["image_width"], doc["image_height"]) == (800, 600)
    got_lines = [l["text"] for p in doc["paragraphs"] for l in p["lines"]]
    assert got_lines == [str(t) for t in otexts if t is not None]


def test_recognition_long_short_split_path():
    """A request mixing >= 64 short lines (T <= 160) with long ones is run as two ragged batches on two streams
    (engine.cpp, T_SPLIT); tokens and boxes must not depend on that."""
    from ocrs_amd import modelfile as mf
    g = mf.build_recognition(n_classes=97, in_h=64, seed=23, hidden=64, chans=(32, 64, 64, 64, 64, 64))
    cal = synth.synthetic_line_crops(9, n=8)
    xp = np.full((8, 1, 64, 300), -0.5, np.float32)
    xp[:, 0, :, :cal.shape[2]] = cal
    g = mf.calibrate_recognition_head(g, lambda buf, x: OracleGraph(buf).run_torch(x), xp)
    rbuf = g.to_bytes()
    gpu = OcrEngine(recognition_model=Model.load_bytes(rbuf))
    ora = OP.OcrEngine(recognition_model=OracleModel(OracleGraph(rbuf), "exact"))
    px = synth.synthetic_page(41, 1000, 1000, lines=40)
    inp = gpu.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    oin = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
    rng = np.random.default_rng(5)
    lines = []
    for i in range(72):   # short: 20..150 px wide at height ~20 -> resized width <= 480 -> T <= 120
        ww, hh = int(rng.integers(20, 150)), int(rng.integers(16, 24))
        lines.append([RotatedRect.new((np.float32(20 + (i % 6) * 160 + ww / 2), np.float32(12 + (i // 6) * 26)),
                                      (np.float32(0.0), np.float32(1.0)), np.float32(ww), np.float32(hh))])
    for i in range(5):    # long: 600..950 px wide at height 14 -> clamped towards 2400 -> T up to 600
        ww = 600 + 80 * i
        lines.append([RotatedRect.new((np.float32(20 + ww / 2), np.float32(400 + 30 * i)), (np.float32(0.0), np.float32(1.0)),
                                      np.float32(ww), np.float32(14.0))])
    got = gpu.recognize_text(inp, [rects_of(l) for l in lines])
    exp = ora.recognize_text(oin, lines)
    assert len(got) == len(exp) == 77
    n_chars = 0
    for a, b in zip(got, exp):
        assert (a is None) == (b is None)
        if a is not None:
            assert str(a) == str(b)
            assert [c.rect for c in a.chars()] == [c.rect.tlbr() for c in b.chars]
            n_chars += len(b.chars)
    assert n_chars > 100


def test_bench_call_sequence_batch_apis_match_single_page_path_and_oracle():
    """bench.py's sequence — pixels resident in HBM, prepare_input_device, detect_words_batch,
    find_text_lines_batch_raw, recognize_text_batch_raw, several steps in flight — must give, page for page,
    what the one-page API gives (itself oracle-checked above), and the oracle's text for the first page."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from ocrs_amd import _lib
    L = _lib.lib()
    dbuf, rbuf = M.detection_model_bytes(), M.recognition_model_bytes()
    gpu = OcrEngine(detection_model=Model.load_bytes(dbuf), recognition_model=Model.load_bytes(rbuf))
    pages = [synth.synthetic_page(60 + i, 512, 640, lines=12) for i in range(3)]
    dptrs = []
    for pg in pages:
        p = C.c_void_p()
        _lib.check(L.ocrs_device_malloc(C.c_size_t(pg.nbytes), C.byref(p)))
        _lib.check(L.ocrs_device_upload(p, pg.ctypes.data_as(C.c_void_p), C.c_size_t(pg.nbytes)))
        dptrs.append(p)

    def step(_=None):
        inputs = [gpu.prepare_input_device(p.value, np.uint8, DimOrder.Hwc, 512, 640, 3) for p in dptrs]
        words = gpu.detect_words_batch(inputs)
        rects, loffs, poffs = gpu.find_text_lines_batch_raw(words)
        chars, coffs = gpu.recognize_text_batch_raw(inputs, rects, loffs, poffs)
        return inputs, words, rects, loffs, poffs, chars, coffs

    with ThreadPoolExecutor(max_workers=3) as ex:   # steps in flight, as the bench runs them
        outs = list(ex.map(step, range(3)))
    inputs, words, rects, loffs, poffs, chars, coffs = outs[0]
    for o in outs[1:]:
        assert np.array_equal(o[2], rects) and np.array_equal(o[5], chars) and np.array_equal(o[6], coffs)
    assert inputs[0].shape == (1, 512, 640)
    texts = []
    for pi, pg in enumerate(pages):
        inp = gpu.prepare_input(ImageSource.from_tensor(pg, DimOrder.Hwc))
        assert np.array_equal(inp.image(), inputs[pi].image())
        w1 = gpu.detect_words(inp)
        assert np.array_equal(w1, words[pi])
        l1 = gpu.find_text_lines(inp, w1)
        lo, hi = int(poffs[pi]), int(poffs[pi + 1])
        assert hi - lo == len(l1)
        for k, line in enumerate(l1):
            assert np.array_equal(line, rects[int(loffs[lo + k]):int(loffs[lo + k + 1])])
        t1 = gpu.recognize_text(inp, l1)
        page_text = []
        for k, tl in enumerate(t1):
            a, b = int(coffs[lo + k]), int(coffs[lo + k + 1])
            got = "".join(chr(c) for c in chars["ch"][a:b])
            assert got == ("" if tl is None else "".join(c.char for c in tl.chars()))
            if tl is not None:
                assert [tuple(int(v) for v in (chars["top"][i], chars["left"][i], chars["bottom"][i], chars["right"][i]))
                        for i in range(a, b)] == [tuple(c.rect) for c in tl.chars()]
            page_text.append(got)
        texts.append(page_text)
    ora = OP.OcrEngine(detection_model=OracleModel(OracleGraph(dbuf), "exact"),
                       recognition_model=OracleModel(OracleGraph(rbuf), "exact"))
    oin = ora.prepare_input(OP.ImageSource.from_tensor(pages[0], "hwc"))
    olines = ora.recognize_text(oin, ora.find_text_lines(oin, ora.detect_words(oin)))
    assert [t for t in texts[0] if t] == [str(t) for t in olines if t is not None and str(t)]
    assert Model.load_bytes(rbuf).flops(1, 64, 300) > 1e8
    for p in dptrs:
        _lib.check(L.ocrs_device_free(p))


def test_gpu_beam_search_equals_host_on_random_matrices():
    """kernels_beam.hip (radix-select prefix beam search, one workgroup per line) against the host implementation
    (itself checked against the textbook formulation and the oracle in tests/test_host_cpu.py): identical steps
    (label, position) for widths 1..128, sharp / flat distributions, excluded labels, T up to 600."""
    from ocrs_amd import _lib
    rng = np.random.default_rng(11)

    def logp(T, C, peak):
        z = rng.normal(0, 3, (T, C))
        for t in range(T):
            z[t, (t // 3) % C] += peak
        return (z - np.log(np.exp(z).sum(1, keepdims=True))).astype(np.float32)

    cases = [(40, 12, 1), (40, 12, 3), (60, 20, 10), (80, 97, 25), (50, 97, 100), (30, 5, 100), (120, 97, 7), (1, 97, 100),
             (600, 97, 100), (200, 97, 128), (75, 128, 64)]
    for T, C, w in cases:
        lp = logp(T, C, float(rng.choice([0.3, 3.0, 8.0])))
        if C > 6:
            lp[:, 5] = -np.inf
            lp[T // 2, 1:] = -np.inf
        assert _lib.ctc_beam_search(lp, w, 2) == _lib.ctc_beam_search(lp, w, 0), (T, C, w)
    # quantised scores: many exact ties at the pruning threshold (slot order must decide)
    lp = np.round(logp(60, 97, 0.3) * 2) / 2
    assert _lib.ctc_beam_search(lp.astype(np.float32), 50, 2) == _lib.ctc_beam_search(lp.astype(np.float32), 50, 0)


def test_beam_search_gpu_and_host_paths_of_the_engine_agree():
    """DecodeMethod::BeamSearch(100) through the engine: GPU kernel (option beam_gpu = 1, default) vs host threads
    (0) on the lines of a page — same text, same char boxes."""
    from ocrs_amd import DecodeMethod, _lib
    rbuf = M.recognition_model_bytes()
    px = synth.synthetic_page(6, 300, 640, lines=10, columns=1)
    eng = OcrEngine(detection_model=Model.load_bytes(M.detection_model_bytes()), recognition_model=Model.load_bytes(rbuf),
                    decode_method=DecodeMethod.BeamSearch(100))
    inp = eng.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    lines = eng.find_text_lines(inp, eng.detect_words(inp))
    assert len(lines) >= 5
    res = {}
    try:
        for mode in (1, 0):
            eng.set_option("beam_gpu", mode)
            got = eng.recognize_text(inp, lines)
            res[mode] = [(str(t), [c.rect for c in t.chars()]) if t else None for t in got]
        # and with the request split into sub-requests of ~2 lines by the engine (beam search on the GPU)
        eng.set_option("beam_gpu", 1)
        eng.set_option("rec_max_pixels", 64 * 700 * 2)
        res["split"] = [(str(t), [c.rect for c in t.chars()]) if t else None for t in eng.recognize_text(inp, lines)]
    finally:
        eng.set_option("beam_gpu", 1)
        eng.set_option("rec_max_pixels", 0)
    assert res[1] == res[0]
    assert res["split"] == res[1]
    assert sum(1 for t in res[1] if t) >= 5
