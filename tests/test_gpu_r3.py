"""GPU tests of the round-3 work, all through the C ABI and all against the oracle / its golden fixtures:

  * engine group (several members in one process, pages dealt in contiguous blocks) incl. the RCCL result gather;
  * request coalescing: concurrent one-page calls share launches, nobody's bits change;
  * configs[4] stream shape: 32 distinct pages in requests of 16, 6 in flight;
  * masks / lines beyond the old capacity limits (the reference has none: detection.rs:41-62, recognition.rs:29-55);
  * recurrence fall-backs: a hidden size without a fused kernel, and gru_mode = 1 really dispatching the fused step kernel.
"""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import models_util as M
from ocrs_amd import DimOrder, EngineGroup, ImageSource, Model, OcrEngine, _lib, synth
from oracle import pipeline as OP
from oracle.geometry import RotatedRect
from oracle.nn import OracleGraph, OracleModel
from test_gpu_bench_scale import _check_page_against_golden, _golden_page

pytestmark = pytest.mark.gpu
N_PAGES = 16


@pytest.fixture(scope="module")
def bufs():
    _lib.require_gpu()
    dbuf, rbuf = M.detection_model_bytes(), M.recognition_model_bytes()
    return dbuf, rbuf, (M.digest(dbuf), M.digest(rbuf))


@pytest.fixture(scope="module")
def pages16():
    return [synth.synthetic_page(s, 1024, 1024, lines=80) for s in range(N_PAGES)]


def rects_of(words):
    return np.array([w.to_array() for w in words], np.float32).reshape(-1, 6)


def _group_pipeline(group, pages):
    inputs = group.prepare_input_batch(pages)
    words = group.detect_words_batch(inputs)
    rects, loffs, poffs = group.find_text_lines_batch_raw(words)
    chars, coffs = group.recognize_text_batch_raw(inputs, rects, loffs, poffs)
    return words, rects, loffs, poffs, chars, coffs


def _check_all_golden(out, digests, n=N_PAGES):
    words, rects, loffs, poffs, chars, coffs = out
    checked = 0
    for pi in range(n):
        g = _golden_page(pi, digests)
        if g is None:
            continue
        _check_page_against_golden(g, words[pi], rects, loffs, int(poffs[pi]), int(poffs[pi + 1]), chars, coffs)
        checked += 1
    assert checked >= 2
    return checked


# ------------------------------------------------------------------ engine group
def test_group_of_two_members_on_one_device_gives_golden_bits(bufs, pages16):
    """devices [0, 0]: two engines (two weight replicas, two host threads per call) on the one GPU of the box; with
    ocrs_group_params.shared_block = 1 the pages of a call are split between them in two contiguous blocks.  The per-request gather
    is the host transport (AUTO inside one process)."""
    dbuf, rbuf, digests = bufs
    if True:
        group = EngineGroup([0, 0], dbuf, rbuf, gather="auto", shared_block=1)
        assert len(group) == 2 and group.member(1)[1] == 0
        out = _group_pipeline(group, pages16)
        _check_all_golden(out, digests)
        lg = group.last_gather()
        assert lg["transport"] == "host" and lg["why_host"] == "host transport requested" and lg["bytes"] > 0
        # three calls in flight on the group (each fans out to both members)
        with ThreadPoolExecutor(3) as ex:
            outs = list(ex.map(lambda _: _group_pipeline(group, pages16), range(3)))
        for o in outs:
            assert all(np.array_equal(a, b) for a, b in zip(o[0], out[0]))
            assert np.array_equal(o[4], out[4]) and np.array_equal(o[5], out[5])
        assert 1 <= group.worker_threads() <= 6      # kept between calls, one per share in flight beyond the callers'
        inputs = group.prepare_input_batch(pages16[:2])
        assert [i.shape for i in inputs] == [(1, 1024, 1024)] * 2
        # uneven dealing: 5 pages -> 3 + 2
        out5 = _group_pipeline(group, pages16[:5])
        _check_all_golden(out5, digests, n=5)


def test_group_rccl_gather_one_member(bufs, pages16):
    """gather = rccl on a one-member group: ncclCommInitAll + grouped ncclAllGather from librccl really run (a
    one-rank communicator is all a one-GPU box offers); results travel device -> device -> host and equal the golden
    bits; the raw gather returns exactly the bytes it was given.
    (Order dependence seen in round 6, not resolved: after tests/test_gpu_parity.py — whose oracle leg imports torch, i.e. torch's
    own bundled ROCm libraries — in the SAME process and without the files the full suite runs in between, /opt/rocm's librccl
    fails in its HSA wrapper ("pfn_hsa_system_get_info failed with 4107") and the group reports the host transport with that
    reason; alone, in tests/test_gpu_r3.py as a whole and in the full `-m gpu` suite it binds and runs.  The reason is in the
    assertion message.)"""
    dbuf, rbuf, digests = bufs
    group = EngineGroup([0], dbuf, rbuf, gather="rccl")
    out = _group_pipeline(group, pages16[:4])
    _check_all_golden(out, digests, n=4)
    lg = group.last_gather()
    assert lg["transport"] == "rccl" and lg["why_host"] == "" and lg["bytes"] > 10000, lg
    rng = np.random.default_rng(5)
    for n in (0, 1, 17, 100003):
        blob = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        data, offs = group.gather([blob])
        assert data == blob and offs == [0, n]
        assert group.last_gather()["transport"] == "rccl"
    # the same group calls from several host threads (collectives on one communicator are serialised inside)
    with ThreadPoolExecutor(3) as ex:
        outs = list(ex.map(lambda _: _group_pipeline(group, pages16[:4]), range(3)))
    for o in outs:
        assert np.array_equal(o[4], out[4]) and all(np.array_equal(a, b) for a, b in zip(o[0], out[0]))


def test_models_and_engines_on_an_explicit_device(bufs, pages16):
    dbuf, rbuf, digests = bufs
    det, rec = Model.load_bytes(dbuf, device=0), Model.load_bytes(rbuf, device=0)
    assert det.device() == 0 and rec.device() == 0
    eng = OcrEngine(detection_model=det, recognition_model=rec)
    assert eng.device() == 0
    n = _lib.device_count()
    with pytest.raises(_lib.OcrsError):   # no such device: reported, not fatal
        Model.load_bytes(dbuf, device=n + 3)
    inp = eng.prepare_input(ImageSource.from_tensor(pages16[0], DimOrder.Hwc))
    g = _golden_page(0, digests)
    assert np.array_equal(eng.detect_words(inp), g["word_rects"])


# ------------------------------------------------------------------ request coalescing
def _one_page(engine, page):
    inp = engine.prepare_input(ImageSource.from_tensor(page, DimOrder.Hwc))
    words = engine.detect_words(inp)
    lines = engine.find_text_lines(inp, words)
    return words, [(str(t), [c.rect for c in t.chars()]) if t else None for t in engine.recognize_text(inp, lines)]


def test_concurrent_one_page_calls_are_merged_and_keep_their_bits(bufs, pages16):
    """The reference's call pattern (one page per call, concurrency from threads: ocrs-cli/src/main.rs:420-446): 12
    host threads, 48 one-page pipelines.  With coalescing (default) calls share launches — fewer merged batches than
    requests — and every caller still gets the bits of its own page (golden fixtures); with coalescing off the same."""
    dbuf, rbuf, digests = bufs
    engine = OcrEngine(detection_model=Model.load_bytes(dbuf), recognition_model=Model.load_bytes(rbuf))
    pages = pages16[:4]
    ref = [_one_page(engine, p) for p in pages]
    for pi in range(4):   # the sequential result is the golden one
        g = _golden_page(pi, digests)
        assert np.array_equal(ref[pi][0], g["word_rects"])
        co = g["char_offsets"]
        for i, t in enumerate(ref[pi][1]):
            assert (t[0] if t else "") == "".join(chr(c) for c in g["chars"][co[i]:co[i + 1], 0])
    s0 = engine.coalesce_stats()
    with ThreadPoolExecutor(max_workers=12) as ex:
        outs = list(ex.map(lambda k: (k % 4, _one_page(engine, pages[k % 4])), range(48)))
    for pi, (w, t) in outs:
        assert np.array_equal(w, ref[pi][0]) and t == ref[pi][1]
    s1 = engine.coalesce_stats()
    for stage in ("detect", "recognize"):
        batches, reqs = s1[stage][0] - s0[stage][0], s1[stage][1] - s0[stage][1]
        assert reqs == 48 and batches < reqs, (stage, batches, reqs)
    try:
        engine.set_option("coalesce", 0)
        with ThreadPoolExecutor(max_workers=6) as ex:
            outs = list(ex.map(lambda k: (k % 4, _one_page(engine, pages[k % 4])), range(12)))
        assert engine.coalesce_stats() == s1   # nothing went through the queues
    finally:
        engine.set_option("coalesce", 2)
    for pi, (w, t) in outs:
        assert np.array_equal(w, ref[pi][0]) and t == ref[pi][1]


def test_an_error_in_a_merged_batch_reaches_only_its_caller(bufs, pages16):
    """One caller of a merged recognition batch passes a line with no words (recognition.rs:433 panics there; here
    OCRS_ERR_INVALID_ARGUMENT): its call fails, the calls merged with it succeed with their usual bits."""
    dbuf, rbuf, digests = bufs
    engine = OcrEngine(detection_model=Model.load_bytes(dbuf), recognition_model=Model.load_bytes(rbuf))
    page = pages16[0]
    inp = engine.prepare_input(ImageSource.from_tensor(page, DimOrder.Hwc))
    words = engine.detect_words(inp)
    lines = engine.find_text_lines(inp, words)
    ref = [(str(t) if t else None) for t in engine.recognize_text(inp, lines)]

    def good(_):
        return [(str(t) if t else None) for t in engine.recognize_text(inp, lines)]

    def bad(_):
        with pytest.raises(_lib.OcrsError, match="no words"):
            engine.recognize_text(inp, [lines[0], np.zeros((0, 6), np.float32)])
        return "raised"

    try:
        engine.set_option("coalesce_window_us", 20000)   # make sure the calls meet in one batch
        with ThreadPoolExecutor(max_workers=8) as ex:
            futs = [ex.submit(bad if k % 4 == 1 else good, k) for k in range(16)]
            res = [f.result() for f in futs]
    finally:
        engine.set_option("coalesce_window_us", 300)
    assert res.count("raised") == 4
    assert all(r == ref for r in res if r != "raised")


# ------------------------------------------------------------------ configs[4] stream shape
def test_stream_of_32_distinct_pages_in_requests_of_16_six_in_flight(bufs):
    """BASELINE configs[4] as bench.py --stream-pages runs it on one GPU: distinct pages (seeds 0..31) resident in HBM,
    requests of 16 pages, 6 requests in flight; pages 0..15 against the golden fixtures, and the repeated requests
    against each other (nothing is cached between requests)."""
    dbuf, rbuf, digests = bufs
    engine = OcrEngine(detection_model=Model.load_bytes(dbuf), recognition_model=Model.load_bytes(rbuf))
    L = _lib.lib()
    pages = [synth.synthetic_page(s, 1024, 1024, lines=80) for s in range(32)]
    dptrs = []
    for pg in pages:
        p = C.c_void_p()
        _lib.check(L.ocrs_device_malloc(C.c_size_t(pg.nbytes), C.byref(p)))
        _lib.check(L.ocrs_device_upload(p, pg.ctypes.data_as(C.c_void_p), C.c_size_t(pg.nbytes)))
        dptrs.append(p)

    def request(k):
        mine = dptrs[(k % 2) * 16:(k % 2) * 16 + 16]
        inputs = [engine.prepare_input_device(p.value, np.uint8, DimOrder.Hwc, 1024, 1024, 3) for p in mine]
        words = engine.detect_words_batch(inputs)
        rects, loffs, poffs = engine.find_text_lines_batch_raw(words)
        chars, coffs = engine.recognize_text_batch_raw(inputs, rects, loffs, poffs)
        return words, rects, loffs, poffs, chars, coffs

    with ThreadPoolExecutor(max_workers=6) as ex:
        outs = list(ex.map(request, range(12)))
    _check_all_golden(outs[0], digests)
    for k in range(2, 12):
        a, b = outs[k], outs[k % 2]
        assert all(np.array_equal(x, y) for x, y in zip(a[0], b[0]))
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
    assert not np.array_equal(outs[0][4], outs[1][4])   # the two halves of the stream are different pages
    # pages 16..31 have no committed fixture: page 16 against the oracle's detection stage (seconds of CPU)
    ora = OP.OcrEngine(detection_model=OracleModel(OracleGraph(dbuf), "exact"))
    oin = ora.prepare_input(OP.ImageSource.from_tensor(pages[16], "hwc"))
    assert np.array_equal(outs[1][0][0], rects_of(ora.detect_words(oin)))
    for p in dptrs:
        _lib.check(L.ocrs_device_free(p))


# ------------------------------------------------------------------ beyond the old capacity limits
def _mask_engine_pair(h, w):
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


def test_masks_with_more_components_than_the_scratch_holds():
    """detection.rs:41-62 takes any mask.  1024x1024: isolated pixels on a 2-pixel grid (262 144 components, four times
    the 65 536 the per-page scratch is sized for -> the page's component stage is re-run with worst-case buffers),
    and salt noise of density 0.3 (tens of thousands of small components).  Rects equal the oracle's, in order."""
    h = w = 1024
    box, gpu, ora = _mask_engine_pair(h, w)
    page = np.zeros((1, h, w), np.float32)
    inp = gpu.prepare_input(ImageSource.from_tensor(page, DimOrder.Chw))
    rng = np.random.default_rng(3)
    dots = np.zeros((h, w), np.uint8)
    dots[::2, ::2] = 1
    dots[100:140, 100:300] = 1          # a few real blobs among the dots
    dots[500:520, 40:900] = 1
    noise = (rng.random((h, w)) < 0.3).astype(np.uint8)
    dense = (rng.random((h, w)) < 0.12).astype(np.uint8)   # the density with the most components per pixel
    comb = np.zeros((h, w), np.uint8)    # ONE component whose border (~38 000 points) is far longer than the contour
    comb[100:104, 20:1000] = 1           # kernel's LDS walk buffer (4 096): the walk is then repeated straight into the arena
    comb[104:200, 20:1000:4] = 1
    comb[300:500, 300:700] = 1           # and an ordinary blob next to it
    for name, mask in (("dots", dots), ("noise30", noise), ("noise12", dense), ("comb", comb)):
        box["prob"] = mask.astype(np.float32)
        got = gpu.detect_words(inp)
        exp = rects_of(ora.detect_words(page))
        assert got.shape == exp.shape, (name, got.shape, exp.shape)
        assert np.array_equal(got, exp), name
    # and in a batch next to an ordinary page: only the overflowing page takes the second pass
    box["prob"] = dots.astype(np.float32)
    a = gpu.detect_words(inp)
    assert len(a) > 0


def test_lines_of_more_than_128_words(bufs):
    """recognition.rs:29-55 builds a polygon of 4 points per word for any number of words.  A line of 300 words on a
    4000-pixel-wide page (1 200 polygon vertices; the crop kernel used to refuse more than 512), straight and with
    the words staggered so that one scanline crosses the polygon ~600 times (more than the kernel's LDS list holds)."""
    dbuf, rbuf, _ = bufs
    gpu = OcrEngine(recognition_model=Model.load_bytes(rbuf))
    ora = OP.OcrEngine(recognition_model=OracleModel(OracleGraph(rbuf), "exact"))
    px = synth.synthetic_page(11, 120, 4000, lines=3, columns=1)
    inp = gpu.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    oin = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
    up = (np.float32(0.0), np.float32(1.0))

    def line(stagger):
        words = []
        for i in range(300):
            dy = (stagger if i % 2 else -stagger)
            words.append(RotatedRect.new((np.float32(20 + 13 * i + 5), np.float32(60 + dy)), up, np.float32(10), np.float32(12)))
        return words

    for stagger in (0, 9):
        words = line(stagger)
        got = gpu.prepare_recognition_input(inp, rects_of(words))
        exp = ora.prepare_recognition_input(oin, words)
        assert got.shape == exp.shape and np.array_equal(got, exp), stagger
    got = gpu.recognize_text(inp, [rects_of(line(0)), rects_of(line(9))])
    exp = ora.recognize_text(oin, [line(0), line(9)])
    for a, b in zip(got, exp):
        assert (a is None) == (b is None)
        if a is not None:
            assert str(a) == str(b) and [c.rect for c in a.chars()] == [c.rect.tlbr() for c in b.chars]


# ------------------------------------------------------------------ recurrence fall-backs
def _small_rec_engine(hidden):
    from ocrs_amd import modelfile as mf
    g = mf.build_recognition(n_classes=97, in_h=64, seed=33, hidden=hidden, chans=(32, 64, 64, 64, 64, 64))
    cal = synth.synthetic_line_crops(9, n=8)
    xp = np.full((8, 1, 64, 300), -0.5, np.float32)
    xp[:, 0, :, :cal.shape[2]] = cal
    g = mf.calibrate_recognition_head(g, lambda buf, x: OracleGraph(buf).run_torch(x), xp)
    rbuf = g.to_bytes()
    return rbuf, OcrEngine(recognition_model=Model.load_bytes(rbuf)), OP.OcrEngine(recognition_model=OracleModel(OracleGraph(rbuf), "exact"))


def _some_lines():
    lines = []
    for i in range(10):
        ww, hh = 60 + 70 * i, 14 + (i % 5) * 3
        lines.append([RotatedRect.new((np.float32(10 + ww / 2), np.float32(30 + 40 * i)), (np.float32(0.0), np.float32(1.0)),
                                      np.float32(ww), np.float32(hh))])
    return lines


@pytest.mark.parametrize("hidden", [32, 64])
def test_recurrence_paths_by_hidden_size_and_mode(hidden):
    """hidden = 32 has neither a persistent nor a fused step kernel: hidden GEMM + gate kernel per time step (that
    branch was lost in round 2 and read uninitialised memory).  hidden = 64 with gru_mode = 1 must run the FUSED
    step kernel (one launch per step, no separate gate kernel), with gru_mode = 0 the persistent one (one launch per
    layer).  All equal to the oracle's exact chain."""
    rbuf, gpu, ora = _small_rec_engine(hidden)
    px = synth.synthetic_page(8, 460, 760, lines=10, columns=1)
    inp = gpu.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    oin = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
    lines = _some_lines()
    exp = ora.recognize_text(oin, lines)
    assert sum(1 for t in exp if t is not None) >= 5

    def run(mode):
        gpu.set_option("gru_mode", mode)
        gpu.enable_timing(2)
        gpu.kernel_stats(reset=True)
        got = gpu.recognize_text(inp, [rects_of(l) for l in lines])
        ks = gpu.kernel_stats(reset=True)
        gpu.enable_timing(0)
        for a, b in zip(got, exp):
            assert (a is None) == (b is None)
            if a is not None:
                assert str(a) == str(b) and [c.rect for c in a.chars()] == [c.rect.tlbr() for c in b.chars]
        return ks["gemm_gru_hidden_mfma"]["launches"], ks["gru_gates"]["launches"]

    try:
        hid0, gates0 = run(0)
        hid1, gates1 = run(1)
    finally:
        gpu.set_option("gru_mode", 0)
    if hidden == 32:
        assert gates0 > 0 and gates0 == hid0 and (hid1, gates1) == (hid0, gates0)   # two launches per step either way
    else:
        assert gates0 == 0 and hid0 == 2                 # persistent: one launch per GRU layer
        assert gates1 == 0 and hid1 > 2 * 20             # fused step kernel: one launch per time step and layer


# ------------------------------------------------------------------ conv patches spanning several images of a width group
def test_narrow_lines_sharing_conv_patches():
    """Width groups of 50 / 100 / 150 px with many lines each: at the conv layers the images are 25 / 12 (50 / 25,
    75 / 37) columns wide, so one 16- or 32-column patch of `conv3x3_ragged` covers two to three images of the
    group's strip, odd widths included (the pooled layer pads each image to an even width).  Same tokens and boxes
    as the oracle, and the same with every image tiled on its own (conv_flat = 0) and with the first two convs as
    separate kernels (conv12_fuse = 0)."""
    rbuf, gpu, ora = _small_rec_engine(64)
    px = synth.synthetic_page(12, 460, 760, lines=10, columns=1)
    inp = gpu.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    oin = ora.prepare_input(OP.ImageSource.from_tensor(px, "hwc"))
    lines = []
    for i in range(23):  # resized widths 38..48 -> 50, 80..96 -> 100, 120..144 -> 150 at height 64
        ww = (12, 15, 25, 30, 40, 45)[i % 6]
        lines.append([RotatedRect.new((np.float32(20 + 30 * (i % 20) + ww / 2), np.float32(30 + 18 * i)),
                                      (np.float32(0.0), np.float32(1.0)), np.float32(ww), np.float32(20.0))])
    exp = ora.recognize_text(oin, lines)
    assert sum(1 for t in exp if t is not None and len(t.chars)) >= 10
    try:
        # conv12_fuse: conv1 + pool + conv2 + pool as one kernel (its patches leave an empty column between images)
        for flat, fuse12 in ((1, 1), (0, 1), (1, 0), (0, 0)):
            gpu.set_option("conv_flat", flat)
            gpu.set_option("conv12_fuse", fuse12)
            got = gpu.recognize_text(inp, [rects_of(l) for l in lines])
            for a, b in zip(got, exp):
                assert (a is None) == (b is None)
                if a is not None:
                    assert str(a) == str(b) and [c.rect for c in a.chars()] == [c.rect.tlbr() for c in b.chars]
    finally:
        gpu.set_option("conv_flat", 1)
        gpu.set_option("conv12_fuse", 1)
