"""GPU parity at the BENCHMARKED configurations (BASELINE.json configs[2] and configs[3]), against golden
fixtures the CPU oracle produced in `exact` mode (tests/golden/make_golden_bench.py; the oracle needs ~16 s of CPU
per page, the fixtures make the comparison free on the GPU box).

  * configs[3]: bench.py's exact call sequence — 16 pages of 1024x1024 (seeds 0..15, 80 lines) resident in HBM,
    prepare_input_device -> detect_words_batch -> find_text_lines_batch_raw -> recognize_text_batch_raw, 6 whole
    steps in flight on 6 host threads — every page's word rects, line grouping, char boxes and text equal to
    the oracle's; the CTC steps (label, pos) of every line through ocrs_engine_recognize_tokens.
  * configs[2]: 2048 crops 64x256 -> width group 300 (T = 75): CTC steps and char boxes of all 2048 lines.
  * both GRU execution modes (persistent kernel / one launch per step) give the same bits.

Bit-exact: integer / index results with ==, float rects with array_equal (ocrs/src/lib.rs:466-488-style exactness).
"""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import models_util as M
from ocrs_amd import DimOrder, ImageSource, Model, OcrEngine, _lib, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_PAGES = 16


@pytest.fixture(scope="module")
def engine():
    _lib.require_gpu()
    dbuf, rbuf = M.detection_model_bytes(), M.recognition_model_bytes()
    eng = OcrEngine(detection_model=Model.load_bytes(dbuf), recognition_model=Model.load_bytes(rbuf))
    eng._digests = (M.digest(dbuf), M.digest(rbuf))
    return eng


def _golden_page(seed, digests):
    path = os.path.join(GOLD, "bench_page_seed%d.npz" % seed)
    if not os.path.exists(path):
        return None
    g = np.load(path)
    assert tuple(g["model_digests"]) == digests, "fixture was made with other model files: re-run make_golden_bench.py"
    return g


def _upload(pages):
    L = _lib.lib()
    dptrs = []
    for pg in pages:
        p = C.c_void_p()
        _lib.check(L.ocrs_device_malloc(C.c_size_t(pg.nbytes), C.byref(p)))
        _lib.check(L.ocrs_device_upload(p, pg.ctypes.data_as(C.c_void_p), C.c_size_t(pg.nbytes)))
        dptrs.append(p)
    return dptrs


def _check_page_against_golden(g, words, rects, loffs, lo, hi, chars, coffs):
    assert np.array_equal(words, g["word_rects"])
    gl = g["line_offsets"]
    assert hi - lo == len(gl) - 1
    base = int(loffs[lo])
    assert np.array_equal(np.asarray(loffs[lo:hi + 1], np.int64) - base, gl)
    assert np.array_equal(rects[base:int(loffs[hi])], g["line_rects"])
    gc, gco = g["chars"], g["char_offsets"]
    cbase = int(coffs[lo])
    assert np.array_equal(np.asarray(coffs[lo:hi + 1], np.int64) - cbase, gco)
    mine = chars[cbase:int(coffs[hi])]
    got = np.stack([mine["ch"].astype(np.int64), mine["top"], mine["left"], mine["bottom"], mine["right"]], axis=1)
    assert np.array_equal(got, gc.astype(np.int64))


def test_full_pipeline_bench_sequence_16_pages_6_steps_in_flight(engine):
    L = _lib.lib()
    pages = [synth.synthetic_page(s, 1024, 1024, lines=80) for s in range(N_PAGES)]
    dptrs = _upload(pages)

    def step(_=None):
        inputs = [engine.prepare_input_device(p.value, np.uint8, DimOrder.Hwc, 1024, 1024, 3) for p in dptrs]
        words = engine.detect_words_batch(inputs)
        rects, loffs, poffs = engine.find_text_lines_batch_raw(words)
        chars, coffs = engine.recognize_text_batch_raw(inputs, rects, loffs, poffs)
        return words, rects, loffs, poffs, chars, coffs

    with ThreadPoolExecutor(max_workers=6) as ex:   # 6 whole steps in flight, as bench.py runs them
        outs = list(ex.map(step, range(12)))
    words, rects, loffs, poffs, chars, coffs = outs[0]
    for o in outs[1:]:   # every in-flight step gives the same bits
        assert all(np.array_equal(a, b) for a, b in zip(o[0], words))
        assert np.array_equal(o[1], rects) and np.array_equal(o[2], loffs) and np.array_equal(o[3], poffs)
        assert np.array_equal(o[4], chars) and np.array_equal(o[5], coffs)
    assert len(loffs) - 1 > 70 * N_PAGES and len(chars) > 2500 * N_PAGES
    checked = 0
    for pi in range(N_PAGES):
        g = _golden_page(pi, engine._digests)
        if g is None:
            continue
        _check_page_against_golden(g, words[pi], rects, loffs, int(poffs[pi]), int(poffs[pi + 1]), chars, coffs)
        checked += 1
    assert checked >= 2
    for p in dptrs:
        _lib.check(L.ocrs_device_free(p))


@pytest.mark.parametrize("seed", [0, 1])
def test_ctc_steps_of_bench_pages_match_golden(engine, seed):
    g = _golden_page(seed, engine._digests)
    assert g is not None
    px = synth.synthetic_page(seed, 1024, 1024, lines=80)
    inp = engine.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    lo = g["line_offsets"]
    lines = [g["line_rects"][lo[i]:lo[i + 1]] for i in range(len(lo) - 1)]
    toks = engine.recognize_tokens(inp, lines)
    to = g["token_offsets"]
    assert len(toks) == len(to) - 1
    for i, t in enumerate(toks):
        assert np.array_equal(np.array(t, np.int32).reshape(-1, 2), g["tokens"][to[i]:to[i + 1]]), i


@pytest.mark.parametrize("name,seed,hh,ww,nl,ncol", [("odd_large", 101, 2200, 3000, 60, 2), ("odd_small", 102, 97, 211, 3, 1)])
def test_pages_of_other_sizes_match_golden(engine, name, seed, hh, ww, nl, ncol):
    """Pages that are not the bench's 1024x1024: a 2200x3000 landscape page (larger than the 800x600 detection input
    in both dimensions: down-scaling, word rects scaled back, long lines) and a 97x211 one (padded up, odd sizes) —
    word rects, line grouping, CTC steps, char boxes against oracle goldens (make_golden_bench.py odd_large/odd_small),
    through the single-page entry points and through the batch ones with both pages of a kind in one request."""
    g = np.load(os.path.join(GOLD, "page_%s.npz" % name))
    assert tuple(g["model_digests"]) == engine._digests
    px = synth.synthetic_page(seed, hh, ww, lines=nl, columns=ncol)
    inp = engine.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    words = engine.detect_words(inp)
    assert np.array_equal(np.array(words, np.float32).reshape(-1, 6), g["word_rects"])
    lines = engine.find_text_lines(inp, words)
    lo = g["line_offsets"]
    assert len(lines) == len(lo) - 1
    for i, l in enumerate(lines):
        assert np.array_equal(np.array(l, np.float32).reshape(-1, 6), g["line_rects"][lo[i]:lo[i + 1]])
    toks = engine.recognize_tokens(inp, lines)
    to = g["token_offsets"]
    for i, t in enumerate(toks):
        assert np.array_equal(np.array(t, np.int32).reshape(-1, 2), g["tokens"][to[i]:to[i + 1]]), i
    # batch entry points, two copies of the page in one request
    inputs = [inp, engine.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))]
    bw = engine.detect_words_batch(inputs)
    rects, loffs, poffs = engine.find_text_lines_batch_raw(bw)
    chars, coffs = engine.recognize_text_batch_raw(inputs, rects, loffs, poffs)
    for pi in range(2):
        _check_page_against_golden(g, bw[pi], rects, loffs, int(poffs[pi]), int(poffs[pi + 1]), chars, coffs)


def test_single_page_requests_in_flight_concurrently(engine):
    """One-page requests (the gate-per-wave GRU kernel's case) from 6 host threads at once, 8 rounds each: every
    result equals the sequential one.  (With several requests in flight the kernels of the others occupy part of the
    chip while a persistent recurrence kernel runs; a placement of that kernel that only appears then once made
    waits for peer workgroups time out — kernels_gru.hip, gru_gates_kernel.)"""
    pages = [synth.synthetic_page(s, 1024, 1024, lines=80) for s in (0, 1, 2)]

    def run(pi):
        inp = engine.prepare_input(ImageSource.from_tensor(pages[pi], DimOrder.Hwc))
        words = engine.detect_words(inp)
        lines = engine.find_text_lines(inp, words)
        return [(str(t), [c.rect for c in t.chars()]) if t else None for t in engine.recognize_text(inp, lines)]

    ref = [run(pi) for pi in range(3)]
    assert all(sum(1 for t in r if t) > 60 for r in ref)
    with ThreadPoolExecutor(max_workers=6) as ex:
        outs = list(ex.map(lambda k: (k % 3, run(k % 3)), range(48)))
    for pi, got in outs:
        assert got == ref[pi]
    # page 0 is also pinned by its golden fixture (chars of every line)
    g = _golden_page(0, engine._digests)
    co = g["char_offsets"]
    for i, t in enumerate(ref[0]):
        exp = "".join(chr(c) for c in g["chars"][co[i]:co[i + 1], 0])
        assert (t[0] if t else "") == exp


def _crops_request(engine):
    n = 2048
    crops = synth.synthetic_line_crops(1000, n=n)
    page = (crops.reshape(1, n * 64, 256) + 0.5).astype(np.float32)
    inp = engine.prepare_input(ImageSource.from_tensor(page, DimOrder.Chw))
    rects = np.zeros((n, 6), np.float32)
    rects[:, 0] = 128.0
    rects[:, 1] = np.arange(n) * 64.0 + 32.0
    rects[:, 2], rects[:, 3] = 0.0, 1.0
    rects[:, 4], rects[:, 5] = 256.0, 64.0
    return inp, rects, n


def test_recognition_only_2048_crops_match_golden(engine):
    """configs[2], exactly as bench.py's recognition-only leg issues it (one request of 2048 lines)."""
    g = np.load(os.path.join(GOLD, "bench_crops_2048.npz"))
    assert tuple(g["model_digests"]) == engine._digests[1:]
    inp, rects, n = _crops_request(engine)
    loffs = np.arange(n + 1, dtype=np.uintp)
    chars, coffs = engine.recognize_text_batch_raw([inp], rects, loffs, np.array([0, n], dtype=np.uintp))
    assert np.array_equal(np.asarray(coffs, np.int64), g["char_offsets"])
    got = np.stack([chars["ch"].astype(np.int64), chars["top"], chars["left"], chars["bottom"], chars["right"]], axis=1)
    assert np.array_equal(got, g["chars"].astype(np.int64))
    toks = engine.recognize_tokens(inp, [rects[i:i + 1] for i in range(n)])
    to = g["token_offsets"]
    flat = np.array([t for ts in toks for t in ts], np.int32).reshape(-1, 2)
    assert np.array_equal(np.cumsum([0] + [len(t) for t in toks]), to)
    assert np.array_equal(flat, g["tokens"])


def test_oversized_recognition_request_is_split_into_sub_requests(engine):
    """A request whose padded line batch exceeds the activation budget ("rec_max_pixels", 2e9 by default) is run as
    consecutive sub-requests; with the budget forced down to ~1/7 of one bench page's lines, and to one line, the
    chars and boxes must be exactly those of the unsplit call."""
    px = synth.synthetic_page(5, 1024, 1024, lines=80)
    inp = engine.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    lines = engine.find_text_lines(inp, engine.detect_words(inp))
    ref = [(str(t), [c.rect for c in t.chars()]) if t else None for t in engine.recognize_text(inp, lines)]
    res = {}
    try:
        for budget in (64 * 1200 * 12, 64 * 50):   # ~12 lines per sub-request; one line per sub-request
            engine.set_option("rec_max_pixels", budget)
            res[budget] = [(str(t), [c.rect for c in t.chars()]) if t else None for t in engine.recognize_text(inp, lines)]
    finally:
        engine.set_option("rec_max_pixels", 0)
    assert sum(1 for t in ref if t) > 60
    for budget, got in res.items():
        assert got == ref, budget


@pytest.mark.parametrize("extra", [512])
def test_more_lines_than_one_grid_holds_second_wave_of_gru_clusters(engine, extra):
    """More row tiles than the slots of a 256-workgroup grid hold (2 560 lines): the persistent GRU kernel is launched with 512 workgroups — the second half becomes
    resident as workgroups of the first exit (clusters never depend on each other).  Chars and boxes must equal the
    per-step kernel's, and the first 2 048 lines (same crops, lines are independent) the golden request's."""
    g = np.load(os.path.join(GOLD, "bench_crops_2048.npz"))
    inp, rects, n = _crops_request(engine)
    idx = np.concatenate([np.arange(n), np.arange(extra) % n])          # the first crops again, as lines 2048..
    rects2 = rects[idx]
    m = n + extra
    loffs = np.arange(m + 1, dtype=np.uintp)
    res = {}
    try:
        for mode in (0, 1):
            engine.set_option("gru_mode", mode)
            res[mode] = engine.recognize_text_batch_raw([inp], rects2, loffs, np.array([0, m], dtype=np.uintp))
    finally:
        engine.set_option("gru_mode", 0)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    chars, coffs = res[0]
    coffs = np.asarray(coffs, np.int64)
    assert np.array_equal(coffs[: n + 1], g["char_offsets"])
    k = int(coffs[n])
    got = np.stack([chars["ch"].astype(np.int64), chars["top"], chars["left"], chars["bottom"], chars["right"]], axis=1)
    assert np.array_equal(got[:k], g["chars"].astype(np.int64))
    rep = min(extra, n)
    assert np.array_equal(got[k: k + int(coffs[rep])], got[: int(coffs[rep])])   # the repeated crops decode to the same text and boxes


def test_gru_modes_give_identical_bits(engine):
    """Persistent recurrence kernel (default) vs one launch per time step: same chars and boxes, on a request with
    ragged lengths (one bench page: T from ~100 to 600) and on the 2048-line request (RT = 4 tiles per wave)."""
    px = synth.synthetic_page(3, 1024, 1024, lines=80)
    inp = engine.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    words = engine.detect_words(inp)
    lines = engine.find_text_lines(inp, words)
    rng = np.random.default_rng(0)
    short = [l[: max(1, int(rng.integers(1, len(l) + 1)))] for l in lines[:40]]   # ragged: prefixes of lines
    req = lines + short
    req2 = lines + short + lines[:70]      # 9-16 row tiles: just beyond the gate-per-wave kernel's range
    cinp, crects, n = _crops_request(engine)
    cl = np.arange(n + 1, dtype=np.uintp)
    res = {}
    try:
        # mode 0 = persistent kernels (general + gate-per-wave for the one-page requests), 1 = per-step launches,
        # 2 = general kernel only (gru_gates 0) with the hand-off forced to write-through stores (gru_local 0),
        # 3 = general kernel only, hand-offs through the XCD's L2 where the placement census allows,
        # 4 = gate-per-wave kernel with forced write-through hand-offs
        for mode in (0, 1, 2, 3, 4):
            engine.set_option("gru_mode", 1 if mode == 1 else 0)
            engine.set_option("gru_local", 0 if mode in (2, 4) else 1)
            engine.set_option("gru_gates", 0 if mode in (2, 3) else 1)
            a = engine.recognize_text(inp, req) + engine.recognize_text(inp, req2)
            b = engine.recognize_text_batch_raw([cinp], crects, cl, np.array([0, n], dtype=np.uintp))
            res[mode] = ([(str(t), [c.rect for c in t.chars()]) if t else None for t in a], b)
    finally:
        engine.set_option("gru_mode", 0)
        engine.set_option("gru_local", 1)
        engine.set_option("gru_gates", 1)
    for mode in (1, 2, 3, 4):
        assert res[0][0] == res[mode][0], mode
        assert np.array_equal(res[0][1][0], res[mode][1][0]) and np.array_equal(res[0][1][1], res[mode][1][1]), mode
    assert sum(1 for t in res[0][0] if t) > 80


@pytest.mark.parametrize("in_hw,depths", [((800, 600), (8, 16, 32, 32, 64, 128, 256)), ((160, 128), (8, 16, 32, 32)),
                                           ((232, 184), (8, 16, 32, 32, 64))])
def test_fused_double_conv_blocks_equal_unfused_and_oracle(in_hw, depths):
    """Detection CNN: the fused LDS-tiled DoubleConv launches (option det_fuse = 1, default) against the unfused
    kernels (det_fuse = 0), bit for bit, on 3 pages through the Model::run seam — tile edges that are not multiples of
    the tile (600 = 18 x 32 + 24; 184 = 5 x 32 + 24), odd sizes under the pools (75 -> 37: the decoder pads `up`), and
    the small model against the oracle's exact chain as well."""
    from oracle.nn import OracleGraph
    dbuf = M.detection_model_bytes(in_hw, depths)
    model = Model.load_bytes(dbuf)
    rng = np.random.default_rng(in_hw[0])
    x = (rng.random((3, 1) + in_hw, dtype=np.float32) - np.float32(0.5))
    x[:, :, 40:60, 30:90] = -0.5   # a dark blob, so that the hand-set "darkness" path fires
    try:
        _lib.set_option("det_fuse", 0)
        ref = model.run(x)
        outs = {}
        for mfma in (2, 1, 0):               # r3: pointwise convs of the fused blocks on MFMA: + C = 32 levels / default / none
            _lib.set_option("det_mfma", mfma)
            _lib.set_option("det_fuse", 2)   # every block shape that has a fused kernel
            outs[(mfma, 2)] = model.run(x)
            _lib.set_option("det_fuse", 1)   # the default: only the shapes where fusion wins
            outs[(mfma, 1)] = model.run(x)
    finally:
        _lib.set_option("det_fuse", 1)
        _lib.set_option("det_mfma", 1)
    got = outs[(1, 1)]   # the defaults
    assert got.shape == ref.shape == (3, 1) + in_hw
    for key, o in outs.items():
        assert np.array_equal(o, ref), key
    if in_hw[0] <= 256:
        assert np.array_equal(got, OracleGraph(dbuf).run_exact(x))


@pytest.mark.parametrize("n", [9, 16])
def test_blocks_without_a_fused_kernel_for_the_request_fall_back_to_the_per_operator_kernels(n):
    """det_mfma = 0 with det_fuse = 1: the (32, 32, 32, 32) and (32, 64, 32, 32) decoder blocks then have ONE fused kernel, the
    row-streaming one, and it declines requests of more than 8 pages.  The executor's query must be made with the request's
    arguments so that those blocks run on the per-operator kernels (round 4 queried with empty arguments, marked the
    operators as done and launched nothing: garbage, silently).  Same bits as det_fuse = 0 and as the defaults."""
    dbuf = M.detection_model_bytes((232, 184), (8, 16, 32, 32, 64))
    model = Model.load_bytes(dbuf)
    rng = np.random.default_rng(n)
    x = (rng.random((n, 1, 232, 184), dtype=np.float32) - np.float32(0.5))
    try:
        ref = model.run(x)                       # defaults
        _lib.set_option("det_fuse", 0)
        unfused = model.run(x)
        _lib.set_option("det_fuse", 1)
        _lib.set_option("det_mfma", 0)
        got = model.run(x)
        _lib.set_option("det_rows", 32)          # and with the row kernels forced on for every request size
        forced = model.run(x)
    finally:
        _lib.set_option("det_fuse", 1)
        _lib.set_option("det_mfma", 1)
        _lib.set_option("det_rows", 1)
    assert np.array_equal(unfused, ref) and np.array_equal(got, ref) and np.array_equal(forced, ref)
