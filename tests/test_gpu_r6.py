"""Round 6 GPU tests: the standing concurrency canary, the isolation regime of the bf16-MFMA kernels and its safe switch.

The canary (tools/hazard_canary.py) calls every stage of the pipeline through the C ABI from six threads and compares EVERY
element of every stage's output — grey page, probability map, word rects, every line crop, every log-probability, every CTC
step — with what the same engine computed for the same input alone on the device.  The reference's contract is concurrent
`Model::run` (recognition.rs:465-485); that results do not depend on what runs beside them is a test here, not an absence of
reports.
"""
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import models_util as M  # noqa: E402
from ocrs_amd import DimOrder, ImageSource, Model, OcrEngine, _lib, synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _assert_clean(rep, min_checks):
    assert not rep["errors"], rep["errors"]
    assert rep["mismatching_checks"] == 0, {k: v for k, v in rep["classes"].items() if v["bad_checks"]}
    assert rep["sequential_pass_after_differs_on_pages"] == 0
    for cls, n in min_checks.items():
        assert rep["classes"][cls]["checks"] >= n, (cls, rep["classes"][cls])


def test_exact_mode_canary_every_stage_every_element_six_threads_twenty_seconds():
    """Exact engine, isolation untouched (every call on its own stream, conv stacks and recurrences on the shared ones): 20 s of
    six threads, every victim class — prepare, resize + detection CNN, threshold + components + contours, line crops, conv stack
    + projections + recurrence + head, CTC — equal to the quiet run, element for element."""
    import hazard_canary as HC
    _lib.require_gpu()
    rep = HC.run(numerics="exact", isolation="auto", seconds=20.0, threads=6)
    assert rep["isolation"]["mode"] == "free"
    _assert_clean(rep, {"prepare": 20, "text_map": 20, "words": 20, "crop": 500, "logits": 20, "tokens": 20})


@pytest.mark.parametrize("mode", ["relaxed", "reduced"])
def test_relaxed_modes_canary_under_the_default_isolation(mode):
    """numerics != exact with the default policy (one stream per device while such an engine exists): the same canary, clean."""
    import hazard_canary as HC
    _lib.require_gpu()
    rep = HC.run(numerics=mode, isolation="auto", seconds=8.0, threads=6)
    assert rep["isolation"]["mode"] == "serial" and rep["isolation"]["relaxed_engines"] == 1
    _assert_clean(rep, {"crop": 100, "logits": 5})
    assert _lib.isolation()["mode"] == "free" and _lib.isolation()["relaxed_engines"] == 0   # the engine is gone, so is the regime


def test_creating_and_destroying_a_relaxed_engine_under_load_drains_instead_of_mixing_regimes():
    """Three threads keep an EXACT engine busy while relaxed engines are created, used and destroyed: the regime change waits for
    the requests in flight (round 5 flipped a flag under them: requests of the two regimes overlapped, two persistent recurrences
    could be half-resident together).  Every result equals the quiet run; nothing times out."""
    _lib.require_gpu()
    det, rec = Model.load_bytes(M.detection_model_bytes()), Model.load_bytes(M.recognition_model_bytes())
    exact = OcrEngine(detection_model=det, recognition_model=rec)
    pre = []
    for s in range(5):
        p = synth.synthetic_page(60 + s, 600 + 100 * s, 800 + 120 * s, lines=12 + 9 * s, columns=1)
        inp = exact.prepare_input(ImageSource.from_tensor(p, DimOrder.Hwc))
        words = exact.detect_words(inp)
        lines = exact.find_text_lines(inp, words)
        pre.append((inp, words, lines, exact.recognize_tokens(inp, lines)))
    bad, seen_modes, stop = [], set(), threading.Event()

    def load(k):
        it = 0
        while not stop.is_set():
            inp, words, lines, tok = pre[(it + k) % len(pre)]
            if exact.recognize_tokens(inp, lines) != tok or exact.detect_words(inp).tobytes() != words.tobytes():
                bad.append((k, it))
            it += 1

    ths = [threading.Thread(target=load, args=(k,)) for k in range(3)]
    [t.start() for t in ths]
    try:
        for rnd in range(4):
            rel = OcrEngine(detection_model=det, recognition_model=rec, numerics="relaxed" if rnd % 2 == 0 else "reduced")
            seen_modes.add(_lib.isolation()["mode"])
            inp, words, lines, _ = pre[rnd % len(pre)]
            a = rel.recognize_tokens(inp, lines)
            assert rel.recognize_tokens(inp, lines) == a
            del rel
            seen_modes.add(_lib.isolation()["mode"])
    finally:
        stop.set()
        [t.join() for t in ths]
    assert seen_modes == {"serial", "free"}
    assert not bad, bad[:5]


def test_detection_batch_of_mixed_page_sizes_equals_the_single_page_results_and_goldens():
    """detect_words_batch over pages of SIX sizes in one request — 1024², 700x500, 1300x900, 2200x3000, 97x211 and the 242-row
    shape of polar-bears.png (the pad branch) — equals detect_words page by page (and, for the 1024² bench page and the two
    odd-size pages, the oracle goldens of tests/golden/); the model runs once over the batch, the size-dependent kernels
    once per size.  Then the same pages as one-page calls from six threads: the coalescer merges whatever waits, results
    byte-equal.  (detection.rs:131-171: any image per call; rounds 1-5 needed one size per batch.)"""
    _lib.require_gpu()
    gold = os.path.join(ROOT, "tests", "golden")
    det, rec = Model.load_bytes(M.detection_model_bytes()), Model.load_bytes(M.recognition_model_bytes())
    eng = OcrEngine(detection_model=det, recognition_model=rec)
    specs = [(0, 1024, 1024, 80, 2), (31, 700, 500, 25, 1), (32, 1300, 900, 50, 1), (101, 2200, 3000, 60, 2), (102, 97, 211, 3, 1),
             (33, 242, 817, 9, 1), (1, 1024, 1024, 80, 2), (34, 700, 500, 12, 1)]
    px = [synth.synthetic_page(s, h, w, lines=nl, columns=c) if s not in (0, 1) else synth.synthetic_page(s, 1024, 1024, lines=80)
          for s, h, w, nl, c in specs]
    inputs = [eng.prepare_input(ImageSource.from_tensor(p, DimOrder.Hwc)) for p in px]
    single = [eng.detect_words(i) for i in inputs]
    assert sum(len(w) for w in single) > 1500
    batch = eng.detect_words_batch(inputs)
    for k, (a, b) in enumerate(zip(single, batch)):
        assert a.tobytes() == b.tobytes(), specs[k]
    # the goldens that exist for these very pages
    g0 = np.load(os.path.join(gold, "bench_page_seed0.npz"))
    assert np.array_equal(batch[0], g0["word_rects"])
    assert np.array_equal(batch[3], np.load(os.path.join(gold, "page_odd_large.npz"))["word_rects"])
    assert np.array_equal(batch[4], np.load(os.path.join(gold, "page_odd_small.npz"))["word_rects"])
    # a different order of the same pages, and sub-batches: page results do not depend on their neighbours
    perm = [5, 2, 7, 0, 4, 1, 6, 3]
    for k, b in zip(perm, eng.detect_words_batch([inputs[k] for k in perm])):
        assert b.tobytes() == single[k].tobytes()
    # the whole pipeline on the mixed batch
    rects, loffs, poffs = eng.find_text_lines_batch_raw(batch)
    chars, coffs = eng.recognize_text_batch_raw(inputs, rects, loffs, poffs)
    texts = []
    for i, inp in enumerate(inputs):
        lines = eng.find_text_lines(inp, single[i])
        want = [str(t) if t else "" for t in eng.recognize_text(inp, lines)]
        got = ["".join(chr(c) for c in chars["ch"][int(coffs[li]):int(coffs[li + 1])]) for li in range(int(poffs[i]), int(poffs[i + 1]))]
        assert got == want, specs[i]
        texts.append(want)
    assert sum(len(t) for t in texts) > 200
    # one-page calls of mixed sizes from six threads (merged by the coalescer)
    bad = []

    def worker(k):
        for it in range(12):
            j = (it * 3 + k) % len(inputs)
            if eng.detect_words(inputs[j]).tobytes() != single[j].tobytes():
                bad.append((k, it, j))

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not bad, bad[:5]
    merged = eng.coalesce_stats()["detect"]
    assert merged[1] >= merged[0] >= 1


def test_group_replay_returns_the_recorded_results_without_touching_the_gpu():
    """ocrs_group_set_replay (the host-side pre-flight hook of bench.py --replay): a four-member group on device 0 records the
    results of eight pages, then replays them — every call returns the recorded bytes in page order whatever the dealing, takes
    about the configured share times, and launches nothing (the engine's stage timers stay at zero)."""
    import time
    from ocrs_amd import EngineGroup
    _lib.require_gpu()
    dbuf, rbuf = M.detection_model_bytes((160, 128), (8, 16, 32, 32)), M.recognition_model_bytes()
    group = EngineGroup([0, 0, 0, 0], dbuf, rbuf, shared_block=2)
    pages = [synth.synthetic_page(80 + s, 192, 256, lines=6, columns=1) for s in range(8)]

    def pipeline(pp):
        inputs = group.prepare_input_batch(pp)
        words = group.detect_words_batch(inputs)
        rects, loffs, poffs = group.find_text_lines_batch_raw(words)
        chars, coffs = group.recognize_text_batch_raw(inputs, rects, loffs, poffs)
        return words, rects, loffs, poffs, chars, coffs

    group.set_replay(1)
    want = pipeline(pages)
    assert sum(len(w) for w in want[0]) > 20 and len(want[4]) > 20
    group.set_replay(2, (0.01, 0.03, 0.05))
    eng0 = group.member(0)[0]
    eng0.enable_timing(1)
    eng0.stage_times(reset=True)
    before = [group.member_stats(m)["shares"] for m in range(4)]
    t0 = time.perf_counter()
    got = pipeline(pages)
    dt = time.perf_counter() - t0
    assert all(np.array_equal(a, b) for a, b in zip(got[0], want[0]))
    for a, b in zip(got[1:], want[1:]):
        assert np.array_equal(a, b)
    assert 0.09 <= dt < 0.5, dt                                   # the three shares' sleeps, the members side by side
    assert all(v[0] == 0 for v in eng0.stage_times(reset=False).values())   # no launch went through member 0's timers
    eng0.enable_timing(0)
    assert [group.member_stats(m)["shares"] - before[m] for m in range(4)] == [3, 3, 3, 3]
    # another order and a subset: results follow the PAGES, not the positions
    sub = [pages[5], pages[0], pages[6]]
    g2 = pipeline(sub)
    for k, j in enumerate((5, 0, 6)):
        assert np.array_equal(g2[0][k], want[0][j])
    # a page the group never recorded is refused, not invented
    with pytest.raises(_lib.OcrsError):
        pipeline([synth.synthetic_page(99, 192, 256, lines=6, columns=1)])
    group.set_replay(0)
    assert all(np.array_equal(a, b) for a, b in zip(pipeline(pages)[0], want[0]))


@pytest.mark.parametrize("cin,cout,relu", [(64, 64, 1), (64, 128, 0), (32, 256, 1), (48, 192, 1), (64, 100, 0), (16, 320, 1)])
def test_tiled_gemm_row_wise_epilogue_is_the_fmaf_chain_on_ragged_rows_and_partial_column_tiles(cin, cout, relu):
    """gemm_tiled_kernel's epilogue (round 6: a wave turns its accumulators through LDS and stores whole rows as float4) on the
    shapes that select each of its forms: 19 173 rows (not a multiple of the 128-row tile: the last tile's row guard), full
    128- and 64-column tiles (row-wise form), a 192- and a 320-column output (row-wise tiles beside a partial one in the same
    launch), 100 columns (direct form only), K = 48 / 16 (single-chunk loop) and K = 32 / 64 (chunk-pair loop), with and
    without ReLU — bit for bit the oracle's k-ascending fmaf chain from the bias."""
    from oracle.nn import OracleGraph
    from ocrs_amd import modelfile as mf
    _lib.require_gpu()
    rng = np.random.default_rng(cin * 1000 + cout)
    w = (rng.standard_normal((1, 1, cin, cout)) * 0.3).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    lift_w = rng.standard_normal((1, 1, 1, cin)).astype(np.float32)
    ops = [mf.Op(mf.OP_CONV, 0, 1, kh=1, kw=1, cin=1, cout=cin, weights=(lift_w, np.zeros(cin, np.float32))),
           mf.Op(mf.OP_CONV, 1, 2, relu=relu, kh=1, kw=1, cin=cin, cout=cout, weights=(w, b))]
    g = mf.Graph(mf.KIND_DETECTION, [-1, 1, -1, -1], ops, 3, 2)
    x = rng.standard_normal((3, 1, 77, 83)).astype(np.float32)       # 3 * 77 * 83 = 19 173 rows: beyond gemm_small's 16 384
    got = Model.load_bytes(g.to_bytes()).run(x)
    exp = OracleGraph(g.to_bytes()).run_exact(x)
    assert got.shape == exp.shape and np.array_equal(got, exp)
    if relu:
        assert (got == 0).any() and (got > 0).any()
