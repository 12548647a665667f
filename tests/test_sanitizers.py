"""Race / memory-safety evidence for the host C++ (the reference gets it from Rust's `Send + Sync`,
detection.rs:67, recognition.rs:316).  tests/sanitize/host_harness.cpp is built from the PRODUCT'S sources with clang++ under
-fsanitize=thread and -fsanitize=address,undefined and driven through: the request coalescer, the engine group's worker
pool and fan-out, the layout analysis on the batch pool (fuzz pages + bench pages), beam search, text items.  Any
sanitizer report fails.  Also here: the randomised layout check against the oracle (tools/fuzz_layout.py as a test).
CPU only."""
import ctypes as C
import os
import subprocess
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "ocrs_amd", "csrc")
OUT = os.path.join(HERE, "sanitize", "_build")
SOURCES = [os.path.join(HERE, "sanitize", "host_harness.cpp")] + [os.path.join(CSRC, f) for f in ("layout.cpp", "ctc_beam.cpp", "text_items.cpp", "jpeg_host.cpp")]
HEADERS = [os.path.join(CSRC, f) for f in ("coalesce.hpp", "coalesce_selftest.hpp", "host_pool.hpp", "geometry.hpp", "engine.hpp", "beam_math.hpp", "jpeg.hpp")]
REPORT_MARKS = ("WARNING: ThreadSanitizer", "ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:")


def harness(kind):
    """kind: 'thread' | 'address,undefined' -> path of the built harness."""
    exe = os.path.join(OUT, "host_harness_" + kind.replace(",", "_"))
    newest = max(os.path.getmtime(p) for p in SOURCES + HEADERS)
    if not os.path.exists(exe) or os.path.getmtime(exe) < newest:
        os.makedirs(OUT, exist_ok=True)
        # ROCm's clang: gcc 11's libtsan has no interceptor for pthread_cond_clockwait (std::condition_variable::wait_until
        # on the steady clock, coalesce.hpp) and then reports a false "double lock" plus false races on everything the
        # mutex protects
        cxx = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else "g++"
        cmd = [cxx, "-std=c++17", "-g", "-O1", "-fno-omit-frame-pointer", "-fsanitize=" + kind, "-fno-sanitize-recover=undefined",
               "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + SOURCES + ["-o", exe, "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, "building the sanitizer harness failed:\n" + r.stderr[-3000:]
    return exe


def run(kind, *args):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([harness(kind)] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=600)
    out = r.stdout + r.stderr
    assert not any(m in out for m in REPORT_MARKS), out[-6000:]
    assert r.returncode == 0, out[-3000:]
    return r.stdout


SAN = ["thread", "address,undefined"]


@pytest.mark.parametrize("kind", SAN)
@pytest.mark.parametrize("mode", ["coalescer", "shares", "beam", "text_items", "numa"])
def test_host_code_is_clean_under_sanitizers(kind, mode):
    assert mode + ": ok" in run(kind, mode)


def _fnv(h, b):
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _layout_inputs():
    from fuzz_pages import fuzz_page, words_array
    pages = [words_array(fuzz_page(s)) for s in range(40)]
    for s in (0, 1):     # two bench pages (~690 words each): the big max_empty_rects searches
        pages.append(np.ascontiguousarray(np.load(os.path.join(HERE, "golden", "bench_page_words_seed%d.npy" % s)), np.float32).reshape(-1, 6))
    pages.append(np.zeros((0, 6), np.float32))
    return pages


@pytest.mark.parametrize("kind", SAN)
def test_layout_analysis_on_the_batch_pool_is_clean_and_gives_the_librarys_lines(kind, tmp_path):
    """find_text_lines over 43 pages (each twice, on 8 threads) under the sanitizers; the hash of all lines equals the
    one computed through libocrs_amd.so's ocrs_engine_find_text_lines_batch."""
    from ocrs_amd import _lib
    pages = _layout_inputs()
    path = tmp_path / "pages.bin"
    with open(path, "wb") as f:
        f.write(np.uint32(len(pages)).tobytes())
        for a in pages:
            f.write(np.uint32(len(a)).tobytes())
            f.write(a.tobytes())
    out = run(kind, "layout", path, 8)
    got = [l for l in out.splitlines() if l.startswith("layout:")][0].split()[-1]
    # the same through the product library
    L = _lib.lib()
    flat = np.ascontiguousarray(np.concatenate(pages))
    woffs = np.cumsum([0] + [len(a) for a in pages]).astype(np.uintp)
    lr, lo, po = C.POINTER(C.c_float)(), C.POINTER(C.c_size_t)(), C.POINTER(C.c_size_t)()
    _lib.check(L.ocrs_engine_find_text_lines_batch(None, C.c_size_t(len(pages)), flat.ctypes.data_as(C.POINTER(C.c_float)),
                                                   woffs.ctypes.data_as(C.POINTER(C.c_size_t)), C.byref(lr), C.byref(lo), C.byref(po)))
    h = 14695981039346656037
    for p in range(len(pages)):
        hp = 14695981039346656037
        for li in range(po[p], po[p + 1]):
            a, b = lo[li], lo[li + 1]
            hp = _fnv(hp, np.uint64(b - a).tobytes())
            hp = _fnv(hp, np.ctypeslib.as_array(lr, shape=(max(int(woffs[-1]), 1) * 6,))[a * 6:b * 6].tobytes())
        h = _fnv(h, np.uint64(hp).tobytes())
    for ptr in (lr, lo, po):
        L.ocrs_buffer_free(ptr)
    assert got == "%016x" % h


@pytest.mark.parametrize("kind", ["address,undefined"])
def test_jpeg_parser_and_huffman_decoders_survive_corrupt_streams_under_asan(kind, tmp_path):
    """jpeg_host.cpp parses bytes from files: intact streams of every flavour, every prefix length class, bit flips in headers,
    tables and entropy-coded data — each must return coefficients or an error, with no out-of-bounds access, overflow or
    leak; the intact ones hash like the library's own decode."""
    import io
    from PIL import Image
    from ocrs_amd import synth
    rng = np.random.default_rng(11)
    px = synth.synthetic_page(2, 96, 128, lines=6, columns=1)
    intact = []
    for ss, prog, rst in ((0, False, 0), (2, False, 3), (1, True, 0), (2, True, 4)):
        b = io.BytesIO()
        Image.fromarray(px).save(b, "JPEG", quality=70, subsampling=ss, progressive=prog, **({"restart_marker_blocks": rst} if rst else {}))
        intact.append(b.getvalue())
    streams = list(intact)
    for data in intact:
        for cut in sorted(set(int(v) for v in rng.integers(0, len(data), 40))):
            streams.append(data[:cut])
        for _ in range(120):
            a = bytearray(data)
            for _ in range(int(rng.integers(1, 6))):
                a[int(rng.integers(2, len(a)))] ^= 1 << int(rng.integers(0, 8))
            streams.append(bytes(a))
    path = tmp_path / "streams.bin"
    with open(path, "wb") as f:
        f.write(np.uint32(len(streams)).tobytes())
        for sdata in streams:
            f.write(np.uint32(len(sdata)).tobytes())
            f.write(sdata)
    out = run(kind, "jpeg", path)
    line = [l for l in out.splitlines() if l.startswith("jpeg:")][0]
    assert int(line.split()[1]) >= len(intact)


def _fuzz_range(lo_hi):
    lo, hi = lo_hi
    sys.path.insert(0, ROOT)
    from fuzz_pages import fuzz_page, words_array
    from ocrs_amd import _lib
    from oracle.layout import find_text_lines as oracle_ftl
    L = _lib.lib()
    bad = []
    for seed in range(lo, hi):
        words = fuzz_page(seed)
        a = words_array(words)
        lr, lo_, nl = C.POINTER(C.c_float)(), C.POINTER(C.c_size_t)(), C.c_size_t(0)
        _lib.check(L.ocrs_engine_find_text_lines(None, None, a.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(len(a)), C.byref(lr),
                                                 C.byref(lo_), C.byref(nl)))
        offs = [lo_[i] for i in range(nl.value + 1)]
        flat = np.ctypeslib.as_array(lr, shape=(max(len(a), 1) * 6,))[: len(a) * 6].reshape(-1, 6).copy()
        L.ocrs_buffer_free(lr)
        L.ocrs_buffer_free(lo_)
        exp = oracle_ftl(words)
        got = [flat[offs[i]:offs[i + 1]] for i in range(nl.value)]
        if not (len(got) == len(exp) and all(np.array_equal(g, np.array([w.to_array() for w in e], np.float32).reshape(-1, 6))
                                             for g, e in zip(got, exp))):
            bad.append(seed)
    return bad


def test_layout_fuzz_200_random_pages_equal_the_oracle():
    """tools/fuzz_layout.py as a test: 200 random pages (columns of small / tall / wide / rotated words) through
    layout.cpp give the oracle's lines bit for bit (order, membership, rect bits)."""
    chunks = [(s, s + 25) for s in range(5000, 5200, 25)]
    with ProcessPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as ex:
        bad = [s for part in ex.map(_fuzz_range, chunks) for s in part]
    assert not bad, "layout differs from the oracle on fuzz seeds %s" % bad
