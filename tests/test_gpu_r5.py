"""GPU tests of the round-5 work, through the C ABI:

  * per-engine tuning options: two engines in one process differ, a process default never reaches an existing engine;
  * relaxed numerics (ocrs_engine_params.numerics): tokens and boxes against the exact engine, log-probs within a stated tolerance;
  * memory: pool statistics stay bounded over requests of varied sizes, results equal the sequential run's.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import models_util as M
from ocrs_amd import DimOrder, ImageSource, Model, OcrEngine, _lib, numerics_report as NR, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# Relaxed numerics, recognition log-probs: |relaxed - exact| on the values that are finite in both.  The relaxed contraction
# is fp32-accurate term by term (bf16 x 3 split, fp32 accumulate) but sums in another order, and the gates use the 1-ulp
# hardware exp / rcp: differences are those of two correct fp32 evaluations, amplified by the 600-step recurrence.
LOGPROB_TOL = 2e-3


@pytest.fixture(scope="module")
def pair():
    _lib.require_gpu()
    det, rec = Model.load_bytes(M.detection_model_bytes()), Model.load_bytes(M.recognition_model_bytes())
    return (OcrEngine(detection_model=det, recognition_model=rec), OcrEngine(detection_model=det, recognition_model=rec, numerics="relaxed"))


def test_two_engines_in_one_process_own_their_options(pair):
    """ocrs_engine_set_option changes one engine; ocrs_set_option (the process default) only what is created afterwards; two
    engines with different kernel selections run concurrently and return the same bits."""
    exact, _ = pair
    det, rec = Model.load_bytes(M.detection_model_bytes()), Model.load_bytes(M.recognition_model_bytes())
    other = OcrEngine(detection_model=det, recognition_model=rec, options={"gru_mode": 1, "det_stream": 0, "det_rows": 0, "ccl_quad": 0, "conv12_fuse": 0})
    assert other.get_option("gru_mode") == 1 and exact.get_option("gru_mode") == 0
    assert other.get_option("numerics") == 0 and pair[1].get_option("numerics") == 1
    with pytest.raises(_lib.OcrsError):
        other.set_option("numerics", 1)                      # fixed at creation
    with pytest.raises(_lib.OcrsError):
        other.set_option("gru_waves", 16)                    # removed in round 5
    try:
        _lib.set_option("det_fuse", 0)
        assert exact.get_option("det_fuse") == 1            # an existing engine keeps its copy
        third = OcrEngine(detection_model=det)
        assert third.get_option("det_fuse") == 0            # a new one starts from the default
    finally:
        _lib.set_option("det_fuse", 1)
    px = synth.synthetic_page(21, 700, 900, lines=40)
    def run(eng):
        inp = eng.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
        words = eng.detect_words(inp)
        return words, eng.recognize_tokens(inp, eng.find_text_lines(inp, words))
    ref = run(exact)
    with ThreadPoolExecutor(4) as ex:
        outs = list(ex.map(run, [exact, other, exact, other]))
    for w, t in outs:
        assert np.array_equal(w, ref[0]) and t == ref[1]
    assert len(ref[1]) > 20


def test_relaxed_numerics_keep_boxes_and_tokens_and_stay_within_the_logprob_tolerance(pair):
    """Two bench pages and 256 crops through an exact and a relaxed engine: word boxes identical (detection runs the same
    kernels in both modes), CTC tokens and char boxes identical — or the flips printed and at most 1 line in 1 000 —, log-probs
    within LOGPROB_TOL.  The relaxed engine really runs other kernels: its log-probs are NOT bit-identical."""
    exact, relaxed = pair
    pages = [synth.synthetic_page(s, 1024, 1024, lines=80) for s in (0, 5)]
    rep = NR.compare_pixels(exact, relaxed, pages)
    inp, lines = NR.crops_request(exact, synth, n=256)
    rc = NR.compare_page(exact, relaxed, inp, lines=lines)
    tot = NR.merge([rep, rc])
    print("relaxed vs exact:", {k: v for k, v in tot.items() if k != "flipped"}, tot["flipped"][:3])
    assert tot["lines"] > 350 and tot["tokens"] > 5000
    assert rep["box_flips"] == 0 and rep["max_abs_dprob_map"] == 0.0
    assert tot["nonfinite_mismatch"] == 0
    assert tot["token_flip_lines"] <= max(1, tot["lines"] // 1000) and tot["label_flip_lines"] == 0, tot["flipped"]
    assert tot["char_box_flips"] <= tot["token_flip_lines"] * 4
    assert 0.0 < tot["max_abs_dlogprob"] < LOGPROB_TOL
    # the other kernel selections under relaxed / reduced numerics (unfused conv1 + conv2, per-image conv patches, per-step GRU):
    # same tokens as the exact engine on page 0
    det, rec = Model.load_bytes(M.detection_model_bytes()), Model.load_bytes(M.recognition_model_bytes())
    i0 = exact.prepare_input(ImageSource.from_tensor(pages[0], DimOrder.Hwc))
    lines0 = exact.find_text_lines(i0, exact.detect_words(i0))
    want = exact.recognize_tokens(i0, lines0)
    for mode in ("relaxed", "reduced"):
        for opts in ({"conv12_fuse": 0}, {"conv_flat": 0}, {"gru_mode": 1}, {"gru_gates": 0, "gru_local": 0}):
            alt = OcrEngine(detection_model=det, recognition_model=rec, numerics=mode, options=opts)
            got = alt.recognize_tokens(i0, lines0)
            differ = sum(1 for a, b in zip(got, want) if [x[0] for x in a] != [x[0] for x in b])
            assert differ <= (0 if mode == "relaxed" else 1), (mode, opts, differ)
    # the exact engine is still exact: page 0 against the oracle's golden fixture
    g = np.load(os.path.join(GOLD, "bench_page_seed0.npz"))
    i0 = exact.prepare_input(ImageSource.from_tensor(pages[0], DimOrder.Hwc))
    assert np.array_equal(exact.detect_words(i0), g["word_rects"])


def test_pools_stay_bounded_over_requests_of_varied_sizes():
    """Pages of random sizes (200-1 600 pixels a side, 1-120 lines) from four threads for a few seconds, with small caps on the
    cached bytes: device and pinned memory in use return to their idle level, the caches respect their caps (blocks above them
    go back to the driver on the trimmer thread), and every result equals the sequential run's."""
    _lib.require_gpu()
    det, rec = Model.load_bytes(M.detection_model_bytes((320, 256), (8, 16, 32, 32, 64))), Model.load_bytes(M.recognition_model_bytes())
    eng = OcrEngine(detection_model=det, recognition_model=rec)
    rng = np.random.default_rng(11)
    cases = []
    for s in range(24):
        h, w = int(rng.integers(200, 1601)), int(rng.integers(200, 1601))
        cases.append(synth.synthetic_page(100 + s, h, w, lines=int(rng.integers(1, max(2, min(120, h // 14)))), columns=1 + (w > 900)))
    def run(px):
        inp = eng.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
        words = eng.detect_words(inp)
        return words, eng.recognize_tokens(inp, eng.find_text_lines(inp, words))
    ref = [run(px) for px in cases]
    _lib.pool_trim()
    base = _lib.pool_stats()          # weights only: nothing cached, no activation arena
    cap_dev, cap_pin = 256 << 20, 8 << 20
    _lib.pool_configure(device_cached_cap=cap_dev, pinned_cached_cap=cap_pin)
    try:
        peak_cached = 0
        with ThreadPoolExecutor(4) as ex:
            for rep in range(3):
                outs = list(ex.map(run, cases))
                for (w, t), (rw, rt) in zip(outs, ref):
                    assert np.array_equal(w, rw) and t == rt
                st = _lib.pool_stats()
                peak_cached = max(peak_cached, st["device_cached"])
                assert st["device_cached"] <= cap_dev and st["pinned_cached"] <= cap_pin, st
        _lib.pool_trim()                 # cached blocks and the conv stacks' shared arena (kept at its high-water mark) given back
        st = _lib.pool_stats()
        assert st["device_live"] <= base["device_live"] + (1 << 20) and st["pinned_live"] <= base["pinned_live"] + (1 << 16), (base, st)
        assert st["device_cached"] == 0
        assert st["device_driver_frees"] > base["device_driver_frees"]      # the cap was reached and blocks went back
        assert st["device_cap"] == cap_dev and st["pinned_cap"] == cap_pin
    finally:
        _lib.pool_configure(device_cached_cap=base["device_cap"], pinned_cached_cap=base["pinned_cap"])


# ------------------------------------------------------------------ rotated / skewed pages end to end
ROT_CASES = [(n, a) for n in ("why-rust", "polar-bears", "rust-book") for a in (3, -3, 10, -10, 90)]


def _rot_case(name, angle):
    import sys
    import zlib
    sys.path.insert(0, GOLD)
    from make_golden_rotated import rotated_pixels
    g = np.load(os.path.join(GOLD, "rotated", "%s_%+d.npz" % (name, angle)))
    base = np.load(os.path.join(GOLD, "reference", name + ".npz"))
    px = rotated_pixels(base["pixels"], name, angle)
    if zlib.crc32(px.tobytes()) != int(g["pixel_crc"][0]) or px.shape != tuple(g["pixel_shape"]):
        pytest.skip("this PIL build resamples differently from the one that made the fixture")
    dbuf, rbuf = M.detection_model_bytes(ink=tuple(g["ink"])), M.recognition_model_bytes()
    assert [M.digest(dbuf), M.digest(rbuf)] == list(g["model_digests"]), "synthetic model files changed: re-run make_golden_rotated.py"
    return g, px, OcrEngine(detection_model=Model.load_bytes(dbuf), recognition_model=Model.load_bytes(rbuf))


def _bits_sum(a):
    return int(np.frombuffer(np.ascontiguousarray(a).tobytes(), np.uint32).sum(dtype=np.uint64))


@pytest.mark.parametrize("name,angle", ROT_CASES)
def test_rotated_page_through_the_one_page_api_equals_the_oracle_golden(name, angle):
    """The reference's images rotated by +-3, +-10 and 90 degrees (tests/golden/make_golden_rotated.py): word rects with real
    `up` vectors, lines grouped along a slope (or in vertical columns at 90 degrees), slanted line polygons, crops resampled
    from rotated boxes, char boxes cut from slanted polygons — every stage equal to the oracle's exact golden."""
    g, px, eng = _rot_case(name, angle)
    inp = eng.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    assert _bits_sum(inp.image()) == int(g["grey_bits_sum"][0])
    prob = eng.detect_text_pixels(inp)
    assert prob.shape == tuple(g["mask_shape"])
    assert np.array_equal(np.packbits(prob > np.float32(eng.detection_threshold())), g["mask"])
    assert _bits_sum(prob) == int(g["prob_bits_sum"][0])
    words = eng.detect_words(inp)
    assert np.array_equal(words, g["word_rects"])
    up = words[:, 2:4]
    if angle != 90 and len(words) > 20:
        assert np.count_nonzero(np.abs(up[:, 0]) > 0.02) > len(words) // 2      # the rects really are rotated
    lines = eng.find_text_lines(inp, words)
    assert np.array_equal(np.cumsum([0] + [len(l) for l in lines]), g["line_offsets"])
    assert np.array_equal(np.concatenate(lines) if lines else np.zeros((0, 6), np.float32), g["line_rects"])
    for i, (shape, bits) in enumerate(zip(g["crop_shapes"], g["crop_bits_sums"])):
        crop = eng.prepare_recognition_input(inp, lines[i])
        assert crop.shape == tuple(shape) and _bits_sum(crop) == int(bits), (name, angle, i)
    toks = eng.recognize_tokens(inp, lines)
    assert np.array_equal(np.array([t for ts in toks for t in ts], np.int32).reshape(-1, 2), g["tokens"])
    assert np.array_equal(np.cumsum([0] + [len(t) for t in toks]), g["token_offsets"])
    assert eng.get_text(inp) == str(g["text"][0])


@pytest.mark.parametrize("name,angle", [("why-rust", 3), ("polar-bears", -10), ("rust-book", 90), ("polar-bears", 90)])
def test_rotated_page_through_the_batch_api_equals_the_oracle_golden(name, angle):
    """Three copies of a rotated page in one batch request: golden word rects, line grouping and char boxes for every copy."""
    from test_gpu_bench_scale import _check_page_against_golden
    g, px, eng = _rot_case(name, angle)
    inputs = [eng.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc)) for _ in range(3)]
    words = eng.detect_words_batch(inputs)
    rects, loffs, poffs = eng.find_text_lines_batch_raw(words)
    chars, coffs = eng.recognize_text_batch_raw(inputs, rects, loffs, poffs)
    for pi in range(3):
        _check_page_against_golden(g, words[pi], rects, loffs, int(poffs[pi]), int(poffs[pi + 1]), chars, coffs)


def test_relaxed_numerics_on_rotated_pages_and_with_beam_search(tmp_path):
    """The numerics modes beyond the upright greedy case: a rotated reference image (slanted polygons, resampled crops) and
    DecodeMethod::BeamSearch on the GPU — relaxed gives the exact engine's text, reduced at most one differing line; and the
    CLI's --numerics switch reaches the engine."""
    from ocrs_amd import DecodeMethod
    g, px, exact = _rot_case("polar-bears", -10)
    dbuf, rbuf = M.detection_model_bytes(ink=tuple(g["ink"])), M.recognition_model_bytes()
    det, rec = Model.load_bytes(dbuf), Model.load_bytes(rbuf)
    inp = exact.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    lines = exact.find_text_lines(inp, exact.detect_words(inp))
    want = [str(t) if t else None for t in exact.recognize_text(inp, lines)]
    assert "\n".join(t for t in want if t is not None) == str(g["text"][0])
    for mode, allowed in (("relaxed", 0), ("reduced", 1)):
        eng = OcrEngine(detection_model=det, recognition_model=rec, numerics=mode)
        got = [str(t) if t else None for t in eng.recognize_text(inp, lines)]
        assert sum(1 for a, b in zip(got, want) if a != b) <= allowed, mode
    beam_exact = OcrEngine(detection_model=det, recognition_model=rec, decode_method=DecodeMethod.BeamSearch(20))
    beam_relaxed = OcrEngine(detection_model=det, recognition_model=rec, decode_method=DecodeMethod.BeamSearch(20), numerics="relaxed")
    be = [str(t) if t else None for t in beam_exact.recognize_text(inp, lines[:20])]
    br = [str(t) if t else None for t in beam_relaxed.recognize_text(inp, lines[:20])]
    assert sum(1 for a, b in zip(be, br) if a != b) <= 1 and sum(1 for t in be if t) >= 10
    # the CLI with --numerics relaxed on the same pixels (PNG on disk, synthetic default models of the CLI)
    from PIL import Image
    from ocrs_amd import cli
    Image.fromarray(synth.synthetic_page(3, 300, 640, lines=10, columns=1), "RGB").save(tmp_path / "page.png")
    outs = {}
    for mode in ("exact", "relaxed"):
        assert cli.main([str(tmp_path / "page.png"), "--numerics", mode, "-o", str(tmp_path / (mode + ".txt"))]) == 0
        outs[mode] = open(tmp_path / (mode + ".txt")).read()
    assert outs["exact"] == outs["relaxed"] and len(outs["exact"]) > 20


# ------------------------------------------------------------------ the recurrence of the relaxed modes (kernels_gru_split.hip)
def _rec_model(hidden):
    from ocrs_amd import modelfile as mf
    from oracle.nn import OracleGraph
    g = mf.build_recognition(n_classes=97, in_h=64, seed=40 + hidden, hidden=hidden, chans=(32, 64, 64, 64, 64, 64))
    cal = synth.synthetic_line_crops(9, n=8)
    xp = np.full((8, 1, 64, 300), -0.5, np.float32)
    xp[:, 0, :, :cal.shape[2]] = cal
    return mf.calibrate_recognition_head(g, lambda buf, x: OracleGraph(buf).run_torch(x), xp).to_bytes()


@pytest.mark.parametrize("hidden", [64, 128, 256])
def test_split_recurrence_by_hidden_size_and_request_shape(hidden):
    """numerics = relaxed / reduced run the recurrence with the state cut into 3 / 2 bf16 planes, in one persistent launch per
    layer, for the hidden sizes the exact persistent kernel serves.  Requests of one line, of 10 ragged lines and of 330 lines
    (21 row tiles: several per wave, lengths 8 .. 170 steps): labels equal to the exact engine's but for near-ties (relaxed: at most
    1 line of the 330, reduced: 1 in 100), log-probs within the tolerance — and NOT the bits of the same mode with one fp32 launch per step (gru_mode = 1),
    which shares every other kernel: the split kernel really ran."""
    _lib.require_gpu()
    rec = Model.load_bytes(_rec_model(hidden))
    exact = OcrEngine(recognition_model=rec)
    px = synth.synthetic_page(8, 460, 760, lines=10, columns=1)
    inp = exact.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    def rect(i, ww, hh):   # one upright word per line: centre, up vector, width, height
        return np.array([[10 + ww / 2, 30 + 40 * (i % 10), 0.0, 1.0, ww, hh]], np.float32)
    ten = [rect(i, 60 + 70 * i, 14 + (i % 5) * 3) for i in range(10)]
    many = [rect(i, 30 + (i * 37) % 700, 14 + (i % 5) * 3) for i in range(330)]
    # (random weights: many near-ties between classes; with every log-prob within tol a label can only flip where the exact margin is < 2 tol)
    for mode, tol, flip_rate in (("relaxed", LOGPROB_TOL, 0.004), ("reduced", 20 * LOGPROB_TOL, 0.01)):
        split = OcrEngine(recognition_model=rec, numerics=mode)
        steps = OcrEngine(recognition_model=rec, numerics=mode, options={"gru_mode": 1})
        for lines in (ten[:1], ten, many):
            split.enable_timing(2)
            split.kernel_stats(reset=True)
            got = split.recognize_tokens(inp, lines)
            ks = split.kernel_stats(reset=True)
            split.enable_timing(0)
            # one launch per layer and sub-request (the 330 lines go as two), like the input projections
            assert ks["gemm_gru_hidden_mfma"]["launches"] == ks["gemm_gru_input_mfma"]["launches"] in (2, 4) and ks["gru_gates"]["launches"] == 0
            want = exact.recognize_tokens(inp, lines)
            differ = sum(1 for a, b in zip(got, want) if [x[0] for x in a] != [x[0] for x in b])
            flips_allowed = int(flip_rate * len(lines)) if len(lines) > 100 else 0
            le, ls, lp = exact.recognize_logits(inp, lines), split.recognize_logits(inp, lines), steps.recognize_logits(inp, lines)
            worst, same_bits = 0.0, True
            for a, b, c in zip(le, ls, lp):
                assert a.shape == b.shape == c.shape and np.array_equal(np.isfinite(a), np.isfinite(b))
                fin = np.isfinite(a)
                worst = max(worst, float(np.max(np.abs(a[fin] - b[fin]))) if fin.any() else 0.0)
                same_bits = same_bits and np.array_equal(b, c)
            assert worst < tol, (mode, len(lines), worst)
            assert differ <= flips_allowed, (mode, len(lines), differ, worst)
            assert not same_bits, (mode, len(lines))
        assert sum(len(t) for t in want) > 300


# ------------------------------------------------------------------ relaxed modes under concurrency
@pytest.mark.parametrize("mode", ["relaxed", "reduced"])
def test_relaxed_results_do_not_depend_on_what_runs_beside_them(mode):
    """One-page requests of varied sizes from six threads against the sequential run of the SAME engine: every token equal.
    (Round 5 found the line crops of a request changing — 64-byte pieces, the lanes 48..63 of a wave — while another request's
    bf16-split conv kernels ran on the same CUs; an engine with numerics != exact therefore makes its device run ONE kernel at a
    time: DeviceContext::serialize, DESIGN.md 6.5.  Without that this test fails within a second.)  Also: a line's result does
    not depend on how many rows share its launches (the input projection takes the split kernel for every M)."""
    _lib.require_gpu()
    import threading
    det, rec = Model.load_bytes(M.detection_model_bytes()), Model.load_bytes(M.recognition_model_bytes())
    eng = OcrEngine(detection_model=det, recognition_model=rec, numerics=mode)
    rng = np.random.default_rng(7)
    pre = []
    for s in range(18):
        h, w = int(rng.integers(200, 1801)), int(rng.integers(300, 2201))
        px = synth.synthetic_page(700 + s, h, w, lines=int(rng.integers(1, max(2, min(90, h // 14)))), columns=1 + (w > 1200))
        inp = eng.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
        words = eng.detect_words(inp)
        pre.append((inp, words, eng.find_text_lines(inp, words)))
    ref = [eng.recognize_tokens(i, l) for i, _, l in pre]
    assert sum(len(r) for r in ref) > 300
    # alone or together: the same bits (log-probs of the first lines of page 0 on their own and inside the whole page's request)
    i0, _, l0 = max(pre, key=lambda t: len(t[2]))
    whole = eng.recognize_logits(i0, l0)
    for k in (0, len(l0) // 2):
        assert np.array_equal(eng.recognize_logits(i0, [l0[k]])[0], whole[k])
    bad = []

    def worker(k):
        for it in range(20):
            j = (it * 7 + k) % len(pre)
            inp, words, lines = pre[j]
            if eng.recognize_tokens(inp, lines) != ref[j] or eng.detect_words(inp).tobytes() != words.tobytes():
                bad.append(j)

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not bad, (mode, sorted(set(bad)))


@pytest.mark.parametrize("hidden", [64, 128, 256])
def test_blocked_handoff_of_the_exact_recurrence_kernels_by_hidden_size(hidden):
    """The exact persistent kernels (general and gate-per-wave) hand the state over through the blocked exchange buffer of round 5:
    for every hidden size they serve, requests of 1 line (one step tile, one row live), 10 ragged lines (gate-per-wave kernel: every
    tile its own cluster) and 330 lines (general kernel, several tiles per wave, tiles with idle rows) give the bits of the
    per-step kernels (gru_mode = 1), with L2 and with write-through hand-offs."""
    _lib.require_gpu()
    rec = Model.load_bytes(_rec_model(hidden))
    step = OcrEngine(recognition_model=rec, options={"gru_mode": 1})
    px = synth.synthetic_page(8, 460, 760, lines=10, columns=1)
    inp = step.prepare_input(ImageSource.from_tensor(px, DimOrder.Hwc))
    def rect(i, ww, hh):
        return np.array([[10 + ww / 2, 30 + 40 * (i % 10), 0.0, 1.0, ww, hh]], np.float32)
    ten = [rect(i, 60 + 70 * i, 14 + (i % 5) * 3) for i in range(10)]
    many = [rect(i, 30 + (i * 37) % 700, 14 + (i % 5) * 3) for i in range(330)]
    for opts in ({}, {"gru_local": 0}, {"gru_gates": 0}):
        eng = OcrEngine(recognition_model=rec, options=opts)
        for lines in (ten[:1], ten, many):
            want, got = step.recognize_logits(inp, lines), eng.recognize_logits(inp, lines)
            assert all(np.array_equal(a, b) for a, b in zip(want, got)), (hidden, opts, len(lines))
            eng.enable_timing(2)
            eng.kernel_stats(reset=True)
            eng.recognize_tokens(inp, lines)
            ks = eng.kernel_stats(reset=True)
            eng.enable_timing(0)
            assert ks["gemm_gru_hidden_mfma"]["launches"] == ks["gemm_gru_input_mfma"]["launches"]   # persistent: one launch per layer
