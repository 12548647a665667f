"""GPU tests of the round-4 work, all through the C ABI and against the oracle / its golden fixtures:

  * the engine group's RCCL gather with TWO and EIGHT members on the one GPU of the box, through a test double of
    librccl (tests/stubs/rccl_stub.cpp, loaded with OCRS_RCCL_LIB): slot layout, length prefixes, per-member streams and
    syncs, several gathers in flight — code that a one-GPU box could otherwise never execute;
  * the final result gather (ocrs_group_final_gather) and its fall-backs;
  * default dealing between members that share a device (whole calls, rotating);
  * the device binding of an entry point lasts for the call only (needs two GPUs; skipped otherwise).
"""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import stub_util
from ocrs_amd import DimOrder, EngineGroup, ImageSource, Model, OcrEngine, _lib
from test_gpu_r3 import _check_all_golden, _group_pipeline, bufs, pages16  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rccl_stub(monkeypatch):
    path = stub_util.rccl_stub_path()
    monkeypatch.setenv("OCRS_RCCL_LIB", path)
    lib = C.CDLL(path)          # the same mapping group.cpp's dlopen gets: its counters are the ones we read
    def stats():
        out = (C.c_uint64 * 5)()
        lib.ocrs_rccl_stub_stats(out)
        return dict(zip(("inits", "gathers", "groups", "max_ranks", "bytes"), [int(v) for v in out]))
    return stats


@pytest.mark.parametrize("members", [2, 8])
def test_group_rccl_gather_with_several_members_through_the_librccl_double(bufs, pages16, rccl_stub, members):
    """gather = rccl on [0] * G with the double: every per-request gather runs gather_rccl with G ranks — the golden bits
    of pages 0-15 come back, raw gathers are byte-exact (incl. empty and unequal payloads), three calls in flight."""
    dbuf, rbuf, digests = bufs
    before = rccl_stub()
    if True:
        group = EngineGroup([0] * members, dbuf, rbuf, gather="rccl", shared_block=16 // members)
        out = _group_pipeline(group, pages16)
        _check_all_golden(out, digests)
        lg = group.last_gather()
        assert lg["transport"] == "rccl" and lg["why_host"] == "" and lg["bytes"] > 100000
        rng = np.random.default_rng(members)
        for sizes in ([0] * members, [1] + [0] * (members - 1), [17, 100003] + [5] * (members - 2),
                      list(rng.integers(0, 70000, members))):
            payloads = [rng.integers(0, 256, int(n), dtype=np.uint8).tobytes() for n in sizes]
            data, offs = group.gather(payloads)
            assert data == b"".join(payloads) and offs == [0] + list(np.cumsum([len(p) for p in payloads]))
            assert group.last_gather()["transport"] == "rccl"
        with ThreadPoolExecutor(3) as ex:
            outs = list(ex.map(lambda _: _group_pipeline(group, pages16), range(3)))
        for o in outs:
            assert all(np.array_equal(a, b) for a, b in zip(o[0], out[0]))
            assert np.array_equal(o[4], out[4]) and np.array_equal(o[5], out[5])
        # uneven shares: 5 pages
        out5 = _group_pipeline(group, pages16[:5])
        _check_all_golden(out5, digests, n=5)
    after = rccl_stub()
    assert after["inits"] == before["inits"] + 1 and after["max_ranks"] >= members
    assert after["gathers"] - before["gathers"] >= members * (2 * 5 + 4)     # G all-gathers per gather, 2 gathers per pipeline
    assert after["groups"] - before["groups"] >= 2 * 5 + 4


def test_final_gather_uses_rccl_when_it_can_and_an_rccl_failure_reaches_only_that_call(bufs, pages16, rccl_stub, monkeypatch):
    """AUTO: per-request gathers on the host, the FINAL gather through RCCL (here the double, 4 members).  A collective that
    fails is an error of that call (OCRS_ERR_DEVICE), the group keeps working."""
    dbuf, rbuf, digests = bufs
    group = EngineGroup([0, 0, 0, 0], dbuf, rbuf, gather="auto")
    out = _group_pipeline(group, pages16[:4])
    _check_all_golden(out, digests, n=4)
    assert group.last_gather()["transport"] == "host"
    texts = [("member %d: " % m).encode() + bytes(range(m * 10, m * 10 + 7)) * (m + 1) for m in range(4)]
    data, offs = group.final_gather(texts, "auto")
    assert data == b"".join(texts) and offs[-1] == len(data)
    lg = group.last_gather()
    assert lg["transport"] == "rccl" and lg["why_host"] == ""
    assert group.final_gather(texts, "host")[0] == data and group.last_gather()["transport"] == "host"
    n = rccl_stub()["gathers"]
    monkeypatch.setenv("OCRS_RCCL_STUB_FAIL_GATHER", str(n + 2))     # the second member's call of the next gather
    with pytest.raises(_lib.OcrsError) as ei:
        group.final_gather(texts, "rccl")
    assert "RCCL error" in str(ei.value)
    monkeypatch.delenv("OCRS_RCCL_STUB_FAIL_GATHER")
    assert group.final_gather(texts, "rccl")[0] == data and group.last_gather()["transport"] == "rccl"
    out2 = _group_pipeline(group, pages16[:4])
    assert np.array_equal(out2[4], out[4])


def test_members_sharing_a_device_take_whole_calls_in_turn(bufs, pages16):
    """Default dealing on [0, 0]: a 16-page call is not split below group_shared_block = 16 pages — one member takes it,
    the next call goes to the other member (worker threads are kept, none is created per call)."""
    dbuf, rbuf, digests = bufs
    group = EngineGroup([0, 0], dbuf, rbuf)
    e0, e1 = group.member(0)[0], group.member(1)[0]
    for e in (e0, e1):
        e.enable_timing(1)
    outs = [_group_pipeline(group, pages16) for _ in range(2)]
    _check_all_golden(outs[0], digests)
    assert np.array_equal(outs[1][4], outs[0][4])
    t0, t1 = e0.stage_times(), e1.stage_times()
    # both members worked (prepare / detect / recognize of one call each, give or take the rotation's start)
    assert sum(n for _, n in t0.values()) > 0 and sum(n for _, n in t1.values()) > 0
    assert group.worker_threads() <= 2


@pytest.mark.skipif(_lib.device_count() < 2 if os.path.exists(_lib.LIB_PATH) else True, reason="needs two GPUs")
def test_an_entry_point_leaves_the_callers_hip_device_as_it_found_it(bufs, pages16):
    """DeviceScope must restore the calling thread's device (torch or the caller may use another GPU)."""
    dbuf, rbuf, _ = bufs
    hip = C.CDLL("libamdhip64.so")
    det, rec = Model.load_bytes(dbuf, device=0), Model.load_bytes(rbuf, device=0)
    eng = OcrEngine(detection_model=det, recognition_model=rec)
    assert hip.hipSetDevice(1) == 0
    inp = eng.prepare_input(ImageSource.from_tensor(pages16[0], DimOrder.Hwc))
    eng.detect_words(inp)
    d = C.c_int(-1)
    assert hip.hipGetDevice(C.byref(d)) == 0 and d.value == 1
    group = EngineGroup([0, 1], dbuf, rbuf)
    _group_pipeline(group, pages16[:4])
    assert hip.hipGetDevice(C.byref(d)) == 0 and d.value == 1
    assert hip.hipSetDevice(0) == 0


# ------------------------------------------------------------------ the reference's own images as parity fixtures
REF_IMAGES = ("why-rust", "polar-bears", "rust-book")


def _ref_case(name):
    from test_golden import _ref_fixture
    g, dbuf, rbuf = _ref_fixture(name)
    eng = OcrEngine(detection_model=Model.load_bytes(dbuf), recognition_model=Model.load_bytes(rbuf))
    return g, eng


def _bits_sum(a):
    return int(np.frombuffer(np.ascontiguousarray(a).tobytes(), np.uint32).sum(dtype=np.uint64))


@pytest.mark.parametrize("name", REF_IMAGES)
def test_reference_image_through_the_one_page_api_equals_the_oracle_golden(name):
    """ocrs-cli/test-data/{why-rust,polar-bears}.png and ocrs/examples/rust-book.jpg, loaded as the CLI loads them
    (into_rgb8), through prepare_input -> detect_text_pixels / detect_words -> find_text_lines -> recognize: grey page,
    probability map (checksum of the f32 bits), mask bits, word rects, line grouping, CTC steps, char boxes and text are
    the oracle's (tests/golden/make_golden_reference_images.py).  polar-bears is 242 rows high: the pad branch of
    detection.rs:159-160; the masks are glyph-shaped (holes, touching letters, specks)."""
    g, eng = _ref_case(name)
    px = np.ascontiguousarray(g["pixels"])
    inp = eng.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    assert _bits_sum(inp.image()) == int(g["grey_bits_sum"][0])
    prob = eng.detect_text_pixels(inp)
    assert prob.shape == tuple(g["mask_shape"])
    assert np.array_equal(np.packbits(prob > np.float32(eng.detection_threshold())), g["mask"])
    assert _bits_sum(prob) == int(g["prob_bits_sum"][0])
    words = eng.detect_words(inp)
    assert np.array_equal(words, g["word_rects"])
    lines = eng.find_text_lines(inp, words)
    assert np.array_equal(np.cumsum([0] + [len(l) for l in lines]), g["line_offsets"])
    assert np.array_equal(np.concatenate(lines), g["line_rects"])
    toks = eng.recognize_tokens(inp, lines)
    assert np.array_equal(np.array([t for ts in toks for t in ts], np.int32).reshape(-1, 2), g["tokens"])
    assert np.array_equal(np.cumsum([0] + [len(t) for t in toks]), g["token_offsets"])
    assert eng.get_text(inp) == str(g["text"][0])


@pytest.mark.parametrize("name", REF_IMAGES)
def test_reference_image_through_the_batch_api_equals_the_oracle_golden(name):
    """The same image three times in one batch request (detect_words_batch -> find_text_lines_batch ->
    recognize_text_batch), two requests in flight: every copy gives the golden word rects, line grouping and char boxes."""
    from test_gpu_bench_scale import _check_page_against_golden
    g, eng = _ref_case(name)
    px = np.ascontiguousarray(g["pixels"])

    def request(_):
        inputs = [eng.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc)) for _ in range(3)]
        words = eng.detect_words_batch(inputs)
        rects, loffs, poffs = eng.find_text_lines_batch_raw(words)
        chars, coffs = eng.recognize_text_batch_raw(inputs, rects, loffs, poffs)
        return words, rects, loffs, poffs, chars, coffs

    with ThreadPoolExecutor(2) as ex:
        outs = list(ex.map(request, range(2)))
    for words, rects, loffs, poffs, chars, coffs in outs:
        for pi in range(3):
            _check_page_against_golden(g, words[pi], rects, loffs, int(poffs[pi]), int(poffs[pi + 1]), chars, coffs)


def test_mixed_load_soak_20_seconds():
    """tools/soak.py as a test: one-page pipelines (coalesced), 16-page batch pipelines and engine-group calls at once
    for 20 s, every result compared with the sequential reference; any error or differing byte fails."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "errors: none" in r.stdout, (r.stdout + r.stderr)[-3000:]


# ------------------------------------------------------------------ row-streaming DoubleConv blocks (kernels_det_stream.hip)
@pytest.mark.parametrize("in_hw,n,depths", [((96, 64), 3, (8, 16, 32, 32)), ((131, 157), 2, (8, 16, 32, 32)), ((61, 59), 1, (8, 16)),
                                            ((240, 121), 5, (8, 16, 32)), ((176, 272), 2, (8, 16, 32, 32, 64)), ((402, 250), 1, (8, 16, 32, 32, 64))])
def test_streaming_detection_blocks_equal_the_tiled_blocks_and_the_oracle(in_hw, n, depths):
    """option det_stream: the full-resolution DoubleConv blocks as row-streaming register kernels (a wave per 64-column
    strip, DPP lane shifts for the horizontal taps, the weights as a tape through the SGPRs).  Same bits as the LDS-tiled
    blocks (det_stream 0) and as the oracle's exact chain, for every segment height (8 / 14 / 32 rows per wave; 1 = chosen
    from the request size), on sizes whose last strip / last segment are partial, odd sizes (the decoder's ConvTranspose
    output is one row / column short of the skip tensor: zero-padded `up` channels) and an image narrower than a strip."""
    import models_util as M
    from oracle.nn import OracleGraph
    buf = M.detection_model_bytes(in_hw, depths)
    rng = np.random.default_rng(5)
    x = (rng.random((n, 1) + in_hw, dtype=np.float32) - 0.5).astype(np.float32)
    m = Model.load_bytes(buf)
    exp = OracleGraph(buf).run_exact(x)
    try:
        # (wave kernels of the 8-channel level, workgroup kernels of the 8-64-channel levels): the decoder block at full
        # resolution has both, the workgroup kernel goes first when it is on
        for mode in ((0, 0), (8, 8), (14, 14), (32, 32), (1, 1), (8, 0), (32, 0), (1, 0), (0, 20), (0, 1)):
            _lib.set_option("det_stream", mode[0])
            _lib.set_option("det_rows", mode[1])
            got = m.run(x)
            assert got.shape == exp.shape and np.array_equal(got, exp), mode
    finally:
        _lib.set_option("det_stream", 1)
        _lib.set_option("det_rows", 1)


def test_streaming_detection_blocks_on_random_sizes():
    """The same comparison on eight random model-input sizes (61-330 pixels a side, 1-4 pages): strips, segments and the
    ConvTranspose's zero-padded last row / column fall differently every time."""
    import models_util as M
    from oracle.nn import OracleGraph
    rng = np.random.default_rng(2024)
    try:
        for _ in range(8):
            in_hw = (int(rng.integers(61, 331)), int(rng.integers(61, 331)))
            n = int(rng.integers(1, 5))
            buf = M.detection_model_bytes(in_hw, (8, 16, 32, 32) if min(in_hw) >= 100 else (8, 16, 32))
            x = (rng.random((n, 1) + in_hw, dtype=np.float32) - 0.5).astype(np.float32)
            m = Model.load_bytes(buf)
            exp = OracleGraph(buf).run_exact(x)
            for mode in ((1, 1), (8, 8), (32, 32), (14, 0), (0, 14)):
                _lib.set_option("det_stream", mode[0])
                _lib.set_option("det_rows", mode[1])
                assert np.array_equal(m.run(x), exp), (in_hw, n, mode)
    finally:
        _lib.set_option("det_stream", 1)
        _lib.set_option("det_rows", 1)


# ------------------------------------------------------------------ component labelling, four pixels per thread
@pytest.mark.parametrize("h,w", [(100, 1032), (100, 2052), (100, 260), (33, 64), (50, 8)])
def test_quad_pixel_component_kernels_equal_the_byte_kernels_and_the_oracle(h, w):
    """option ccl_quad: labelling and root compaction with four mask pixels per thread (a wave's run segment is 256 pixels,
    a block's 1 024).  Widths that end inside a wave / cross the 256- and 1 024-pixel boundaries / are narrower than one
    thread group; masks whose runs and diagonal contacts straddle those boundaries.  Same rects in the same order as the
    one-pixel-per-thread kernels and the oracle."""
    from test_gpu_parity import _adversarial_masks, _mask_engine_pair, rects_of
    box, gpu, ora = _mask_engine_pair(h, w)
    page = np.zeros((1, h, w), np.float32)
    inp = gpu.prepare_input(ImageSource.from_tensor(page, DimOrder.Chw))
    rng = np.random.default_rng(w)
    masks = list(_adversarial_masks(h, w)) if h >= 100 and w >= 200 else [("empty", np.zeros((h, w), np.uint8)), ("full", np.ones((h, w), np.uint8))]
    m = np.zeros((h, w), np.uint8)
    for y in range(2, h - 2, 4):                       # long runs ending / starting exactly at the segment boundaries
        for b in (64, 256, 1024, 2048):
            if b + 3 < w:
                m[y, max(b - 40, 0):b] = 1             # ends at b - 1
                m[y + 1, b:min(b + 37, w)] = 1         # starts at b: diagonal contact across the boundary
    masks.append(("boundary runs", m))
    m = (rng.random((h, w)) < 0.5).astype(np.uint8)
    m[:, 250:262] = (rng.random((h, 12)) < 0.8) if w > 262 else m[:, 250:262]
    masks.append(("noise50", m))
    m = np.ones((h, w), np.uint8)
    m[h // 2, :] = 0                                   # one background line across every boundary, touching the frame
    m[3:h - 3:5, 1:w - 1] = 0                          # and lines that do not
    masks.append(("background lines", m))
    try:
        for name, mask in masks:
            box["prob"] = mask.astype(np.float32)
            exp = rects_of(ora.detect_words(page))
            for quad in (1, 0):
                gpu.set_option("ccl_quad", quad)
                got = gpu.detect_words(inp)
                assert got.shape == exp.shape and np.array_equal(got, exp), (name, quad)
    finally:
        gpu.set_option("ccl_quad", 1)


def test_quad_pixel_component_kernels_on_page_sized_masks():
    """The same comparison on 1024 x 1024 masks: a thresholded synthetic text page (word blobs), the page with 2 % salt noise
    on top (thousands of one-pixel components between the words) and a dense random texture cropped to text lines."""
    from ocrs_amd import synth
    from test_gpu_parity import _mask_engine_pair, rects_of
    h = w = 1024
    box, gpu, ora = _mask_engine_pair(h, w)
    page = np.zeros((1, h, w), np.float32)
    inp = gpu.prepare_input(ImageSource.from_tensor(page, DimOrder.Chw))
    rng = np.random.default_rng(77)
    grey = synth.synthetic_page(5, h, w, lines=70).astype(np.float32).mean(axis=2)
    text = (grey < 128).astype(np.uint8)
    salt = text | (rng.random((h, w)) < 0.02).astype(np.uint8)
    texture = ((rng.random((h, w)) < 0.55) & (text > 0)).astype(np.uint8)
    try:
        for name, mask in (("text", text), ("text + salt", salt), ("texture", texture)):
            box["prob"] = mask.astype(np.float32)
            exp = rects_of(ora.detect_words(page))
            assert len(exp) > 100, name
            for quad in (1, 0):
                gpu.set_option("ccl_quad", quad)
                got = gpu.detect_words(inp)
                assert got.shape == exp.shape and np.array_equal(got, exp), (name, quad)
    finally:
        gpu.set_option("ccl_quad", 1)


def test_streaming_detection_kernels_at_bench_scale_give_the_golden_word_rects(bufs, pages16):
    """The 16 bench pages (800 x 600 detection input, full seven-level U-Net) through detect_words_batch with the row-streaming
    kernels forced on for every request size and each rows-per-wave / rows-per-workgroup choice, with them off, and in the
    default configuration (16 pages: wave kernels + tiled blocks; 8 pages: wave + workgroup kernels): always the word rects of
    the golden fixtures the oracle made."""
    from test_gpu_bench_scale import _golden_page
    dbuf, rbuf, digests = bufs
    eng = OcrEngine(detection_model=Model.load_bytes(dbuf))
    inputs = [eng.prepare_input(ImageSource.from_tensor(p, DimOrder.Hwc)) for p in pages16]
    golden = [_golden_page(pi, digests) for pi in range(len(pages16))]
    assert sum(g is not None for g in golden) >= 2
    try:
        for mode in ((1, 1), (8, 8), (32, 32), (14, 20), (0, 0), (1, 0), (0, 1)):
            eng.set_option("det_stream", mode[0])
            eng.set_option("det_rows", mode[1])
            for lo, hi in ((0, 16), (0, 8), (8, 11)):
                words = eng.detect_words_batch(inputs[lo:hi])
                for pi in range(lo, hi):
                    if golden[pi] is not None:
                        assert np.array_equal(words[pi - lo], golden[pi]["word_rects"]), (mode, lo, hi, pi)
    finally:
        eng.set_option("det_stream", 1)
        eng.set_option("det_rows", 1)

