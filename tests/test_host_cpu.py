"""CPU-side checks of the product: the C-ABI library loads and exports every
symbol include/ocrs_amd.h declares, and the host-only entry points
(find_text_lines, ImageSource validation, error reporting) agree with the
oracle.  No GPU compute is invoked here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import kat_util as K
import ocrs_amd
from ocrs_amd import _lib
from oracle.geometry import Rect, RotatedRect
from oracle.layout import find_text_lines as oracle_find_text_lines

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ocrs_amd import build
    build.build()
    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "ocrs_amd.h")).read()
    declared = set(re.findall(r"OCRS_API[^;(]*?\b(ocrs_\w+)\s*\(", hdr))
    assert declared == set(_lib.DECLARED_SYMBOLS)
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_header_cites_reference_for_every_entry_point():
    hdr = open(os.path.join(ROOT, "include", "ocrs_amd.h")).read()
    assert hdr.count(".rs:") >= 25


def test_image_source_check_bytes(lib):  # preprocess.rs:274-321
    for ln, w, h, err in [(100, 10, 10, None), (50, 10, 10, "length"), (128, 8, 8, "channel"), (0, 0, 10, "channel")]:
        if err is None:
            ocrs_amd.ImageSource.from_bytes(bytes(ln), (w, h))
        else:
            with pytest.raises(ocrs_amd.ImageSourceError, match=err):
                ocrs_amd.ImageSource.from_bytes(bytes(ln), (w, h))


def test_image_source_from_tensor():  # preprocess.rs:323-360
    a = np.arange(25, dtype=np.uint8).reshape(1, 5, 5)
    ocrs_amd.ImageSource.from_tensor(a, ocrs_amd.DimOrder.Chw)
    with pytest.raises(ocrs_amd.ImageSourceError):
        ocrs_amd.ImageSource.from_tensor(a, ocrs_amd.DimOrder.Hwc)
    with pytest.raises(ocrs_amd.ImageSourceError):
        ocrs_amd.ImageSource.from_tensor(np.zeros((0, 5, 5), np.uint8), ocrs_amd.DimOrder.Chw)


def _host_find_text_lines(lib, words):
    a = np.ascontiguousarray(np.array([w.to_array() for w in words], np.float32).reshape(-1, 6))
    lr = C.POINTER(C.c_float)()
    lo = C.POINTER(C.c_size_t)()
    nl = C.c_size_t(0)
    _lib.check(lib.ocrs_engine_find_text_lines(None, None, a.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(len(a)),
                                               C.byref(lr), C.byref(lo), C.byref(nl)))
    offs = [lo[i] for i in range(nl.value + 1)]
    flat = np.ctypeslib.as_array(lr, shape=(max(len(a), 1) * 6,))[: len(a) * 6].reshape(-1, 6).copy()
    lib.ocrs_buffer_free(lr)
    lib.ocrs_buffer_free(lo)
    return [flat[offs[i]:offs[i + 1]] for i in range(nl.value)]


def _same_lines(got, exp):
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        ea = np.array([w.to_array() for w in e], np.float32).reshape(-1, 6)
        assert g.shape == ea.shape
        assert np.array_equal(g, ea)


def test_find_text_lines_kat(lib):  # layout_analysis.rs:294-350
    left = K.gen_rect_grid((0, 0), (10, 5), (5, 5), (3, 2))
    lb = K.union_rects(left)
    right = K.gen_rect_grid((0, lb[3] + 20), (10, 5), (5, 5), (3, 2))
    words = K.xorshift_shuffle([RotatedRect.from_rect(Rect(*r)) for r in left + right], 1234)
    got = _host_find_text_lines(lib, words)
    assert len(got) == 20 and all(len(l) == 5 for l in got)
    _same_lines(got, oracle_find_text_lines(words))


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_find_text_lines_matches_oracle_on_random_layouts(lib, seed):
    rng = np.random.default_rng(seed)
    words = []
    cols = int(rng.integers(1, 4))
    for c in range(cols):
        x0 = 20 + c * 330
        y = 20
        for _ in range(int(rng.integers(8, 25))):
            h = int(rng.integers(10, 22))
            x = x0 + int(rng.integers(0, 20))
            for _ in range(int(rng.integers(1, 8))):
                w = int(rng.integers(15, 60))
                if x + w > x0 + 300:
                    break
                ang = float(rng.normal(0, 0.03))
                up = (np.float32(np.sin(ang)), np.float32(np.cos(ang)))
                words.append(RotatedRect.new((np.float32(x + w / 2), np.float32(y + h / 2)), up, np.float32(w + 6), np.float32(h + 6)))
                x += w + int(rng.integers(4, 14))
            y += h + int(rng.integers(4, 30))
    order = rng.permutation(len(words))
    words = [words[i] for i in order]
    _same_lines(_host_find_text_lines(lib, words), oracle_find_text_lines(words))


@pytest.mark.parametrize("seed", [0, 1])
def test_find_text_lines_matches_oracle_on_bench_scale_pages(lib, seed):
    """~700 word boxes as the detection stage produces them for a 1024x1024 bench page (tests/golden/make_golden.py
    documents how the fixture was made): the layout search pops ~48k heap entries here, with score ties, so this
    is where an inexact emulation of Rust's BinaryHeap order would show."""
    a = np.load(os.path.join(os.path.dirname(__file__), "golden", "bench_page_words_seed%d.npy" % seed))
    words = [RotatedRect.from_array(r) for r in a]
    assert len(words) > 600
    got = _host_find_text_lines(lib, words)
    exp = oracle_find_text_lines(words)
    assert 70 <= len(exp) <= 90
    _same_lines(got, exp)


@pytest.mark.parametrize("shift", [(40000.0, 0.0), (0.0, 33000.0), (-33500.0, -100.0)])
def test_find_text_lines_beyond_16_bit_coordinates(lib, shift):
    """The layout search keeps 16-bit coordinates and indices when a page's words fit them and 32-bit ones otherwise:
    the same random layout moved past +-32 767 in x or y (the wide path), against the oracle on the moved words."""
    rng = np.random.default_rng(11)
    words = []
    for c in range(2):
        x0 = 30 + c * 420
        y = 25
        for _ in range(18):
            h = int(rng.integers(10, 22))
            x = x0 + int(rng.integers(0, 20))
            for _ in range(int(rng.integers(2, 8))):
                w = int(rng.integers(15, 60))
                if x + w > x0 + 380:
                    break
                words.append(RotatedRect.new((np.float32(x + w / 2 + shift[0]), np.float32(y + h / 2 + shift[1])),
                                             (np.float32(0.0), np.float32(1.0)), np.float32(w + 6), np.float32(h + 6)))
                x += w + int(rng.integers(4, 14))
            y += h + int(rng.integers(4, 30))
    words = [words[i] for i in rng.permutation(len(words))]
    _same_lines(_host_find_text_lines(lib, words), oracle_find_text_lines(words))


def test_find_text_lines_empty(lib):
    assert _host_find_text_lines(lib, []) == []


def test_model_file_errors_are_reported_not_fatal(lib):
    h = C.c_void_p()
    st = lib.ocrs_model_load_bytes(C.c_char_p(b"garbage"), C.c_size_t(7), C.byref(h))
    assert st == 8 and b"model file" in lib.ocrs_last_error()
    st = lib.ocrs_model_load_file(b"/nonexistent/model.ocrsm", C.byref(h))
    assert st == 8 and b"cannot open" in lib.ocrs_last_error()


# ---------------------------------------------------------------- output.rs:217-250, text_items.rs:148-166
def _gen_text_chars(text, width):
    from ocrs_amd import TextChar
    return [TextChar(ch, (0, i * width, 25, i * width + width)) for i, ch in enumerate(text)]


def test_format_json_output_matches_reference_golden(lib):
    import json
    from ocrs_amd import TextLine, output
    lines = [TextLine(_gen_text_chars("line one", 10)), None, TextLine(_gen_text_chars("line two", 10))]
    got = json.loads(output.format_json_output("image.jpeg", (256, 256), lines))
    exp = json.load(open(os.path.join(ROOT, "tests", "golden", "reference", "format-json-expected.json")))["expected"]
    assert got == exp
    # textual: serde_json (no `preserve_order`) writes every object's keys alphabetically, as the golden file has them
    text = output.format_json_output("image.jpeg", (256, 256), lines)
    assert text == json.dumps(exp, indent=2, ensure_ascii=False, sort_keys=True)
    assert text.index('"image_height"') < text.index('"image_width"') < text.index('"paragraphs"') < text.index('"url"')
    assert output.format_text_output(lines).split("\n") == ["line one", "line two"]


def test_item_rotated_rect(lib):  # text_items.rs:148-166
    from ocrs_amd import TextWord, rotated_rect_corners
    word = TextWord(_gen_text_chars("foo", 10))
    assert word.bounding_rect() == (0, 0, 25, 30)
    rr = word.rotated_rect()
    assert (rr[2], rr[3]) == (0.0, -1.0)  # up_axis() == Vec2::from_yx(-1., 0.)
    assert rotated_rect_corners(rr) == [[30.0, 25.0], [0.0, 25.0], [0.0, 0.0], [30.0, 0.0]]


def test_committed_bench_line_keeps_the_driver_contract():
    """profiles/r1_bench_default.json is the last `python bench.py` line measured on an MI355X; the keys the
    driver and the judge read must be there with the right types (a schema check, not a performance check)."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r1_bench_default.json")
    d = json.load(open(path))
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["higher_is_better"] is True and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or isinstance(r["traffic"], (int, float))
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert abs(d["value"] - d["config"]["pages_per_step_per_gpu"] * 1000.0 / d["ms_per_step"]) / d["value"] < 1e-3


def test_no_gpu_means_a_loud_error_not_a_cpu_fallback(lib):
    """The product has no CPU path: without a HIP device, loading a model (the first thing that needs HBM) must
    raise, not quietly compute somewhere else."""
    if _lib.device_count() > 0:
        pytest.skip("a HIP device is visible")
    from ocrs_amd import modelfile as mf
    buf = mf.build_recognition(hidden=8, chans=(4, 8, 8, 8, 8, 8), n_classes=5, in_h=32).to_bytes()
    with pytest.raises(ocrs_amd.OcrsError, match="(?i)device|hip"):
        ocrs_amd.Model.load_bytes(buf)
    with pytest.raises(ocrs_amd.OcrsError):
        _lib.require_gpu()


def test_malformed_model_files_are_rejected_at_load(lib):
    """A stale or corrupted .ocrsm must fail with OCRS_ERR_IO at load (before any device work), not index out of
    bounds later: slot numbers, weight-tensor sizes and blob ranges are validated (model.cpp HipModel::load)."""
    import struct
    import models_util as M
    from ocrs_amd import Model
    from ocrs_amd._lib import OcrsError
    good = bytearray(M.detection_model_bytes((160, 128), (8, 16, 32, 32)))
    hdr, opsz = struct.calcsize("<8sII4qIIIIQ"), struct.calcsize("<I9iII16Q")

    def expect_io(buf):
        with pytest.raises(OcrsError) as e:
            Model.load_bytes(bytes(buf))
        assert e.value.status_name == "IO", e.value

    bad = bytearray(good); struct.pack_into("<i", bad, hdr + 4, 1 << 20)          # op 0: in0 far outside the slot table
    expect_io(bad)
    bad = bytearray(good); struct.pack_into("<i", bad, hdr + 3 * opsz + 12, -5)   # op 3: negative output slot
    expect_io(bad)
    bad = bytearray(good); struct.pack_into("<i", bad, hdr + opsz + 7 * 4, 999)   # op 1 (1x1 conv): cin no longer matches its weights
    expect_io(bad)
    bad = bytearray(good); struct.pack_into("<Q", bad, hdr + 48 + 8, 2 ** 62)     # op 0: weight count overflows the blob
    expect_io(bad)
    expect_io(good[: len(good) // 2])                                               # truncated blob
    bad = bytearray(good); struct.pack_into("<I", bad, 8 + 4 + 4 + 32 + 8, 10 ** 6)  # header: output slot outside the table
    expect_io(bad)


# ---------------------------------------------------------------- recognition.rs:512-514 (decode_beam)
def _random_logp(rng, T, C, peak):
    z = rng.normal(0, 3, (T, C))
    for t in range(T):
        z[t, (t // 3) % C] += peak
    return (z - np.log(np.exp(z).sum(1, keepdims=True))).astype(np.float32)


def test_fast_beam_search_equals_the_textbook_formulation(lib):
    """The engine's prefix beam search (closed-form insertion keys, nth_element) against the same function written as
    the algorithm is usually stated (label trie + per-step candidate map + stable sort), through the C ABI:
    identical steps (label, position) for widths 1..100, with excluded (-inf) labels, sharp and flat distributions."""
    from ocrs_amd import _lib
    rng = np.random.default_rng(7)
    cases = [(40, 12, 1), (40, 12, 3), (60, 20, 10), (80, 97, 25), (50, 97, 100), (30, 5, 100), (120, 97, 7), (1, 97, 100)]
    for T, C, w in cases:
        lp = _random_logp(rng, T, C, float(rng.choice([0.3, 3.0, 8.0])))
        if C > 6:
            lp[:, 5] = -np.inf
            lp[T // 2, 1:] = -np.inf     # a step where only the blank is allowed
        fast, ref = _lib.ctc_beam_search(lp, w, 0), _lib.ctc_beam_search(lp, w, 1)
        assert fast == ref, (T, C, w)
    assert _lib.ctc_beam_search(np.zeros((0, 97), np.float32), 10, 0) == []


def test_beam_search_host_equals_oracle(lib):
    """... and against the oracle's Python restatement (oracle/pipeline.py::ctc_beam_search), which evaluates the
    same fixed float64 log-sum-exp (beam_math.hpp) operation for operation."""
    from ocrs_amd import _lib
    from oracle import pipeline as OP
    rng = np.random.default_rng(8)
    for T, C, w in [(30, 8, 1), (30, 8, 4), (40, 20, 7), (25, 97, 30), (20, 97, 100)]:
        lp = _random_logp(rng, T, C, float(rng.choice([0.3, 2.0, 8.0])))
        got = _lib.ctc_beam_search(lp, w, 0)
        assert got == [(int(a), int(b)) for a, b in OP.ctc_beam_search(lp, w)], (T, C, w)
    # the fixed log-sum-exp stays within a few ulp of libm's
    import math
    for _ in range(2000):
        a, b = float(rng.uniform(-60, 0)), float(rng.uniform(-60, 0))
        assert abs(OP.beam_lse(a, b) - (max(a, b) + math.log1p(math.exp(-abs(a - b))))) < 1e-14


def _gru_plan(lib, lengths, hidden=256):
    a = np.ascontiguousarray(np.asarray(lengths, np.int32))
    ncl, waves = C.c_int32(0), C.c_int32(0)
    tiles = np.full(512, -2, np.int16)
    st = lib.ocrs_gru_tile_plan(a.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(len(a)), C.c_int(hidden), C.byref(ncl),
                                C.byref(waves), tiles.ctypes.data_as(C.POINTER(C.c_int16)))
    return st, ncl.value, tiles, waves.value


_ROUND = (0, 63, 101, 150, 200)   # kRoundCost[0] of kernels_gru.hip: 0.1 us per round with n live tiles (measured)


def _wave_cost(lens):
    """the cost model of gru_assign_tiles (kernels_gru.hip): a round with n live tiles costs _ROUND[n]"""
    ls = sorted(lens, reverse=True) + [0]
    return sum((ls[i] - ls[i + 1]) * _ROUND[i + 1] for i in range(len(ls) - 1))


@pytest.mark.parametrize("n_lines", [1, 16, 17, 77, 1232, 2048, 2049, 4096, 8192])
def test_gru_tile_plan_covers_every_tile_once_and_balances(lib, n_lines):
    """Host side of the persistent GRU kernel (kernels_gru.hip::gru_assign_tiles): every 16-line tile of the
    length-sorted batch goes to exactly one slot (4 waves per cluster, at most 4 tiles each), longest first; the slowest
    slot of the deal is no slower (cost model of the kernel) than under the contiguous deal it replaced."""
    rng = np.random.default_rng(n_lines)
    lengths = np.sort(rng.integers(25, 601, n_lines))[::-1]
    st, ncl, tiles, W = _gru_plan(lib, lengths)
    if n_lines > 4096:
        assert st == 9                                         # OCRS_ERR_CAPACITY: the general kernel holds 4 096 lines
        return
    assert st == 0 and W == 4
    S, per = 4, 4                                              # slots per cluster, tiles per slot
    ntiles = (n_lines + 15) // 16
    assert 1 <= ncl <= (8 if ntiles <= 32 * per else 16)
    used = tiles[: per * S * ncl].reshape(-1, per)
    assert np.all(tiles[per * S * ncl:] == -1)                 # the whole 512-entry buffer is initialised
    flat = used[used >= 0]
    assert sorted(flat.tolist()) == list(range(ntiles))
    tl = [int(lengths[16 * k]) for k in range(ntiles)]
    costs = []
    for row in used:
        idx = [int(t) for t in row if t >= 0]
        assert all(t == -1 for t in row[len(idx):])           # filled from the front
        assert idx == sorted(idx)                              # longest (lowest index) first
        costs.append(_wave_cost([tl[t] for t in idx]))
    rt = -(-ntiles // (S * ncl))
    contiguous = [_wave_cost([tl[k] for k in range((s // S) * S * rt + s % S, min(ntiles, (s // S + 1) * S * rt), S)])
                  for s in range(S * ncl)]
    assert max(costs) <= max(contiguous) + 1e-6


def test_gru_tile_plan_rejects_what_the_kernel_cannot_hold(lib):
    assert _gru_plan(lib, np.full(4097, 50, np.int32))[0] == 9      # OCRS_ERR_CAPACITY: > 4096 lines at H = 256 (general kernel)
    assert _gru_plan(lib, np.full(64, 50, np.int32), hidden=96)[0] == 9  # unsupported hidden size
    assert _gru_plan(lib, [10, 20])[0] == 1                  # OCRS_ERR_INVALID_ARGUMENT: not descending


# ---------------------------------------------------------------- request coalescer (coalesce.hpp), host only
@pytest.mark.parametrize("threads,max_active,window_us", [(1, 2, 300), (12, 2, 300), (12, 1, 0), (24, 3, 2000)])
def test_coalescer_runs_every_request_once_and_merges_under_load(lib, threads, max_active, window_us):
    """The leader/follower queue behind the one-page entry points (ocrs-cli/src/main.rs:420-446 is one page per call):
    every request runs exactly once with its own result, requests of incompatible kinds never share a batch, a batch
    stays within the page budget, errors reach exactly their callers, and with many callers batches carry several
    requests."""
    out = (C.c_uint64 * 5)()
    per = 40
    st = lib.ocrs_coalescer_selftest(threads, per, max_active, 8, C.c_long(window_us), 7, out)
    assert st == 0, lib.ocrs_last_error()
    batches, reqs, errors, wrong, max_pages = [int(v) for v in out]
    n = threads * per
    assert reqs == n and wrong == 0
    assert errors == len([i for i in range(1, n + 1) if i % 7 == 0])
    assert max_pages <= 8
    if threads == 1:
        assert batches == n            # a lone caller is never delayed into a batch
    else:
        assert batches < n             # concurrent callers share batches


# ---------------------------------------------------------------- engine group: dealing and packing (host side)
def test_group_deals_pages_in_contiguous_blocks(lib):
    """block = max(ceil(n / G), min(min_block = 8, n)); page i -> member (i / block) mod G: a 16-page call on 8
    members uses two of them (8 pages each), 10 000 pages on 8 members 1 250 each, in order."""
    for n, g in [(0, 1), (7, 3), (16, 8), (5, 8), (10000, 8), (16, 2), (24, 2), (9, 2), (64, 8), (65, 8)]:
        mo = (C.c_size_t * max(n, 1))()
        pp = (C.c_size_t * g)()
        assert lib.ocrs_group_deal(C.c_size_t(n), C.c_size_t(g), C.c_size_t(0), mo, pp) == 0
        block = max(-(-n // g), min(8, n)) if n else 1
        assert [mo[i] for i in range(n)] == [(i // block) % g for i in range(n)]
        assert [pp[m] for m in range(g)] == [sum(1 for i in range(n) if (i // block) % g == m) for m in range(g)]
        assert all(mo[i] <= mo[i + 1] for i in range(n - 1))          # contiguous, in order
        if n >= 8 * g:
            assert max(pp) - min(pp) <= block and min(pp) > 0            # large calls use every member
        elif n:
            assert min(v for v in pp if v) >= min(8, n) or sum(1 for v in pp if v) == 1 or n % block
    assert lib.ocrs_group_deal(C.c_size_t(4), C.c_size_t(0), C.c_size_t(0), None, None) == 1
    mo = (C.c_size_t * 6)()   # an explicit min_block (ocrs_group_params.min_block)
    assert lib.ocrs_group_deal(C.c_size_t(6), C.c_size_t(3), C.c_size_t(1), mo, None) == 0 and list(mo) == [0, 0, 1, 1, 2, 2]


def test_group_host_gather_concatenates_member_payloads_in_member_order(lib):
    """A group without models needs no GPU: the host transport of the result gather packs the members' payloads in
    member order with their boundaries (the RCCL transport must deliver the same bytes: tests/test_gpu_r4.py)."""
    from ocrs_amd import EngineGroup
    g = EngineGroup([0, 0, 0], gather="host")
    assert len(g) == 3 and [g.member(i)[1] for i in range(3)] == [0, 0, 0]
    rng = np.random.default_rng(1)
    payloads = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (24 * 7, 0, 4096 + 5)]
    data, offs = g.gather(payloads)
    assert data == b"".join(payloads) and offs == [0, 168, 168, 168 + 4101]
    lg = g.last_gather()
    assert lg["transport"] == "host" and lg["bytes"] == len(data) and lg["why_host"]
    g2 = EngineGroup([0, 0], gather="rccl")       # RCCL asked for, impossible with a repeated device: host, and says why
    assert g2.gather([b"a", b"bc"])[0] == b"abc" and "more than once" in g2.last_gather()["why_host"]
    with pytest.raises(ocrs_amd.OcrsError):
        EngineGroup([], gather="host")


def test_group_auto_gather_is_host_per_request_and_rccl_problems_are_never_fatal(lib, monkeypatch):
    """AUTO = host for the per-request gathers (inside one process the results are on the host already); the FINAL gather
    asks for RCCL on a group of two or more members and falls back, with the reason, when librccl cannot be loaded or
    refuses the communicator.  libocrs_amd.so itself has no RCCL dependency (dlopen on first use)."""
    import subprocess
    from ocrs_amd import EngineGroup, _lib
    deps = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in deps
    g = EngineGroup([0, 1], gather="auto")
    assert g.gather([b"xy", b"z"]) == (b"xyz", [0, 2, 3])
    lg = g.last_gather()
    assert lg["transport"] == "host" and lg["why_host"] == "host transport requested"
    # final gather, library missing
    monkeypatch.setenv("OCRS_RCCL_LIB", "/nonexistent/librccl.so")
    assert g.final_gather([b"xy", b"z"], "auto") == (b"xyz", [0, 2, 3])
    lg = g.last_gather()
    assert lg["transport"] == "host" and "librccl unavailable" in lg["why_host"] and "/nonexistent/librccl.so" in lg["why_host"]
    assert g.final_gather([b"", b"q"], "rccl") == (b"q", [0, 0, 1])      # asked for explicitly: still not an error
    assert g.last_gather()["transport"] == "host"
    # a library that is not an RCCL
    monkeypatch.setenv("OCRS_RCCL_LIB", "libm.so.6")
    g3 = EngineGroup([0, 1], gather="rccl")
    assert g3.gather([b"1", b"2"])[0] == b"12" and "lacks ncclCommInitAll" in g3.last_gather()["why_host"]
    # the communicator is refused (test double with failure injection; no HIP call happens before the refusal)
    import stub_util
    monkeypatch.setenv("OCRS_RCCL_LIB", stub_util.rccl_stub_path())
    monkeypatch.setenv("OCRS_RCCL_STUB_FAIL_INIT", "1")
    g4 = EngineGroup([0, 1, 2, 3], gather="auto")
    assert g4.final_gather([b"a", b"", b"bc", b"d"], "auto") == (b"abcd", [0, 1, 1, 3, 4])
    assert "ncclCommInitAll failed" in g4.last_gather()["why_host"]
    # one member: nothing to move between devices
    g1 = EngineGroup([0], gather="auto")
    assert g1.final_gather([b"solo"], "auto")[0] == b"solo" and g1.last_gather()["why_host"] == "one member: nothing to gather"
    with pytest.raises(ocrs_amd.OcrsError):      # an unknown transport is an argument error
        _lib.check(lib.ocrs_group_final_gather(g1._h, C.c_int(7), None, None, None, None))


def test_every_option_of_the_library_is_documented_in_the_header(lib):
    """ocrs_set_option names: the table in common.cpp, ocrs_option_name and the comment block of include/ocrs_amd.h list the
    same options, at most twelve of them (round 5: the experiment switches are gone); the per-engine configuration fields
    are fields of ocrs_engine_params, not options; no getenv outside the once-per-process initialisers."""
    import re
    from ocrs_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "ocrs_amd", "csrc", "common.cpp")).read()
    names = re.findall(r'\{"([a-z0-9_]+)",\s*"OCRS_[A-Z0-9_]+",', src)
    assert 8 <= len(names) <= 12
    assert _lib.option_names() == names
    hdr = open(os.path.join(root, "include", "ocrs_amd.h")).read()
    missing = [n for n in names if '"%s"' % n not in hdr]
    assert not missing, missing
    fields = re.findall(r'\{"([a-z0-9_]+)",\s*nullptr,', src)
    params = re.search(r"typedef struct ocrs_engine_params \{(.*?)\} ocrs_engine_params;", hdr, re.S).group(1)
    assert fields and all(re.search(r"\b%s;" % f, params) for f in fields), fields
    # the table is indexed by `enum Option`: entry i must be the option enumerator i names (a det_tail / gru_waves swap once
    # made one option read the other's value)
    chp = open(os.path.join(root, "ocrs_amd", "csrc", "common.hpp")).read()
    enum = re.search(r"enum Option \{(.*?)OPT_COUNT \}", chp, re.S).group(1)
    enum = re.sub(r"//[^\n]*", "", enum)
    ids = [x.split("=")[0].strip() for x in enum.replace("\n", " ").split(",") if x.strip()]
    ids = [i for i in ids if i != "OPT_PUBLIC_COUNT"]
    assert ["OPT_" + n.upper() for n in names + fields] == ids
    assert lib.ocrs_set_option(b"coalesce", C.c_long(1)) == 1 and lib.ocrs_set_option(b"no_such_option", C.c_long(1)) == 1   # not process options
    # options that round 5 removed with their kernels are still ACCEPTED (and ignored): a caller built against the older header works
    assert lib.ocrs_set_option(b"det_tail", C.c_long(1)) == 0 and lib.ocrs_set_option(b"gru_waves", C.c_long(16)) == 0
    assert _lib.option_names() == names
    # values are validated: the kernels index tables with some of them
    assert lib.ocrs_set_option(b"det_fuse", C.c_long(7)) == 1 and lib.ocrs_set_option(b"det_rows", C.c_long(13)) == 1
    assert b"out of range" in lib.ocrs_last_error()
    assert lib.ocrs_set_option(b"det_rows", C.c_long(20)) == 0 and lib.ocrs_set_option(b"det_rows", C.c_long(1)) == 0
    # environment variables are read by initialisers that run once (options, pool cap, the RCCL library name), never on a launch path
    for f in sorted(os.listdir(os.path.join(root, "ocrs_amd", "csrc"))):
        body = open(os.path.join(root, "ocrs_amd", "csrc", f)).read()
        n = len(re.findall(r"\bgetenv\(", body))
        assert n <= {"common.cpp": 2, "group.cpp": 1}.get(f, 0), (f, n)


def test_numa_placement_helpers_parse_bind_and_restore(lib, tmp_path):
    """numa.hpp behind ocrs_group_member_stats: sysfs cpu lists, the node of a PCI device from a (fake) sysfs tree, and the
    scope that binds a member's share to its GPU's node and puts the thread's mask back — silently doing nothing when the
    host does not say (node -1, no file, CPUs outside the thread's mask)."""
    n, arr = C.c_size_t(0), (C.c_int32 * 32)()
    assert lib.ocrs_numa_parse_cpulist(b"0-3,8,10-11\n", arr, 32, C.byref(n)) == 0 and list(arr)[: n.value] == [0, 1, 2, 3, 8, 10, 11]
    assert lib.ocrs_numa_parse_cpulist(b"", None, 0, C.byref(n)) == 0 and n.value == 0
    assert lib.ocrs_numa_parse_cpulist(b"0-255", None, 0, C.byref(n)) == 0 and n.value == 256     # cpus may be NULL: count only
    for bad in (b"3-1", b"a", b"1,,2", b"1-", b"70000"):
        assert lib.ocrs_numa_parse_cpulist(bad, None, 0, C.byref(n)) == 1, bad
    mine = sorted(os.sched_getaffinity(0))
    root = tmp_path / "sys"
    dev = root / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node1 = root / "devices" / "system" / "node" / "node1"
    node1.mkdir(parents=True)
    (node1 / "cpulist").write_text("%d\n" % mine[0])
    node, inside, after = C.c_int(-9), C.c_int(0), C.c_int(0)
    assert lib.ocrs_numa_bind_selftest(str(root).encode(), b"0000:C1:00.0", C.byref(node), C.byref(inside), C.byref(after)) == 0
    assert (node.value, inside.value, after.value) == (1, 1, len(mine))          # bound to the one CPU inside, restored after
    assert sorted(os.sched_getaffinity(0)) == mine
    (node1 / "cpulist").write_text("4000-4001\n")                                  # CPUs this thread may not use: no binding
    assert lib.ocrs_numa_bind_selftest(str(root).encode(), b"0000:c1:00.0", C.byref(node), C.byref(inside), C.byref(after)) == 0
    assert (node.value, inside.value, after.value) == (1, -1, len(mine))
    (dev / "numa_node").write_text("-1\n")                                         # single-node hosts say -1
    assert lib.ocrs_numa_bind_selftest(str(root).encode(), b"0000:c1:00.0", C.byref(node), C.byref(inside), C.byref(after)) == 0
    assert (node.value, inside.value) == (-1, -1)
    assert lib.ocrs_numa_bind_selftest(str(root).encode(), b"0000:99:00.0", C.byref(node), C.byref(inside), C.byref(after)) == 0
    assert (node.value, inside.value, after.value) == (-1, -1, len(mine))         # unknown device
