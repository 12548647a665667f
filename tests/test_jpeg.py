"""JPEG hand-off (the reference decodes on the host, ocrs-cli/src/main.rs:312-333): host Huffman
decoding + GPU dequantisation / IDCT / upsampling / colour conversion.  The bar: the decoded RGB8 pixels equal what
libjpeg-turbo (PIL — the decoder the oracle's fixtures were made with) produces, byte for byte.

CPU (-m "not gpu"): the HOST half (marker parsing, sequential and progressive entropy decoding, restart intervals) feeding
the numpy restatement of the per-sample half (oracle/jpeg.py) must reproduce PIL on the reference's own JPEG and on
encodings of synthetic pages / the reference's PNGs in every sampling mode; malformed and unsupported streams are refused.
GPU (-m gpu): the HIP kernels give the same bytes; prepare_input_jpeg gives the grey page of prepare_input(PIL pixels);
the whole pipeline on ocrs/examples/rust-book.jpg from its FILE BYTES gives the oracle's golden words, lines and text.
"""
import io
import itertools
import os

import numpy as np
import pytest

import ocrs_amd
from ocrs_amd import _lib, synth
from oracle import jpeg as OJ

PIL_Image = pytest.importorskip("PIL.Image")
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference")


def _encode(px, **kw):
    b = io.BytesIO()
    PIL_Image.fromarray(px).save(b, "JPEG", **kw)
    return b.getvalue()


def _pil(data):
    return np.asarray(PIL_Image.open(io.BytesIO(data)).convert("RGB"))    # == image::open(..).into_rgb8()


def _cases():
    rng = np.random.default_rng(0)
    imgs = {"page": synth.synthetic_page(3, 512, 640, lines=30), "odd": synth.synthetic_page(5, 211, 97, lines=5, columns=1),
            "noise": rng.integers(0, 256, (61, 83, 3), dtype=np.uint8), "w2": rng.integers(0, 256, (9, 2, 3), dtype=np.uint8),
            "w4": rng.integers(0, 256, (17, 4, 3), dtype=np.uint8), "polar": np.load(os.path.join(G, "polar-bears.npz"))["pixels"]}
    for name, px in imgs.items():
        for q, ss, prog, rst in itertools.product((30, 90, 100), (0, 1, 2), (False, True), (0, 5)):
            yield "%s q%d ss%d %s rst%d" % (name, q, ss, "prog" if prog else "seq", rst), _encode(
                px, quality=q, subsampling=ss, progressive=prog, optimize=(q == 90), **({"restart_marker_blocks": rst} if rst else {}))
        grey = np.asarray(PIL_Image.fromarray(px).convert("L"))
        for prog in (False, True):
            yield "%s grey %s" % (name, "prog" if prog else "seq"), _encode(grey, quality=80, progressive=prog)


CASES = list(_cases())


def test_host_entropy_decoder_and_oracle_reproduce_pil_on_the_reference_jpeg():
    """ocrs/examples/rust-book.jpg (1200x1600, PROGRESSIVE 4:4:4, ten scans) from its file bytes."""
    g = np.load(os.path.join(G, "rust-book.npz"))
    data = g["file_bytes"].tobytes()
    info = _lib.jpeg_info(data)
    assert (info["height"], info["width"], info["components"], info["progressive"]) == (1600, 1200, 3, True)
    geom, quant, coef = _lib.jpeg_coefficients(data)
    rgb = OJ.decode_from_coefficients(geom, quant, coef)
    assert np.array_equal(rgb, _pil(data))
    assert np.array_equal(rgb, g["pixels"])          # the pixels the golden fixture was made from


def test_host_entropy_decoder_and_oracle_reproduce_pil_in_every_mode():
    """4:4:4 / 4:2:2 / 4:2:0 / grey x sequential / progressive x restart intervals x qualities, odd sizes, planes of width
    <= 2 (libjpeg's fancy upsampling falls back to replication there), optimised Huffman tables."""
    bad = []
    for name, data in CASES:
        geom, quant, coef = _lib.jpeg_coefficients(data)
        if not np.array_equal(OJ.decode_from_coefficients(geom, quant, coef), _pil(data)):
            bad.append(name)
    assert not bad and len(CASES) > 200, bad[:10]


def test_malformed_and_unsupported_streams_are_refused_not_crashed():
    data = _encode(synth.synthetic_page(1, 64, 64, lines=2, columns=1), quality=80)
    for broken in (b"", b"\xff\xd8", b"\x89PNG\r\n\x1a\n" + b"0" * 64, data[:20], data[: len(data) // 3].replace(b"\xff\xc0", b"\xff\xc9")):
        with pytest.raises(_lib.OcrsError) as ei:
            _lib.jpeg_info(broken)
        assert ei.value.status == 6 and "JPEG" in str(ei.value)      # OCRS_ERR_IMAGE_SOURCE
    cmyk = io.BytesIO()
    PIL_Image.new("CMYK", (16, 16)).save(cmyk, "JPEG")
    with pytest.raises(_lib.OcrsError):
        _lib.jpeg_info(cmyk.getvalue())
    # a truncated scan still decodes (zeros past the end, as libjpeg does) — no error, no crash
    assert _lib.jpeg_info(data[: len(data) - 40])["height"] == 64
    # random garbage after a valid header: every byte pattern must come back as pixels or as an error
    rng = np.random.default_rng(3)
    head = data[: data.index(b"\xff\xda") + 14]
    for _ in range(50):
        junk = head + rng.integers(0, 256, 300, dtype=np.uint8).tobytes() + b"\xff\xd9"
        try:
            _lib.jpeg_coefficients(junk)
        except _lib.OcrsError:
            pass


# ------------------------------------------------------------------------------------------------ GPU
def test_decompression_bombs_are_refused_before_any_allocation():
    """A tiny file whose SOF claims 65535 x 65535 (12 GiB of samples), a stream with hundreds of scans, a sequential stream
    whose scan is repeated: OCRS_ERR_IMAGE_SOURCE at once (the reference's `image` crate enforces a 512 MiB allocation limit)."""
    import time
    data = _encode(synth.synthetic_page(1, 64, 64, lines=2, columns=1), quality=80)
    sof = data.index(b"\xff\xc0")
    huge = data[: sof + 5] + b"\xff\xff\xff\xff" + data[sof + 9:]
    t0 = time.perf_counter()
    for fn in (_lib.jpeg_info, _lib.jpeg_coefficients):
        with pytest.raises(_lib.OcrsError, match="too large") as ei:
            fn(huge)
        assert ei.value.status == 6
    assert time.perf_counter() - t0 < 1.0
    big_ok = data[: sof + 5] + bytes([0x2E, 0xE0, 0x2E, 0xE0]) + data[sof + 9:]      # 12000 x 12000 x 3 = 412 MiB: within budget
    assert _lib.jpeg_info(big_ok)["height"] == 12000
    # the scan of a sequential stream repeated: the second one names components that are already decoded
    sos, eoi = data.index(b"\xff\xda"), data.rindex(b"\xff\xd9")
    with pytest.raises(_lib.OcrsError, match="two scans"):
        _lib.jpeg_coefficients(data[:eoi] + data[sos:eoi] + b"\xff\xd9")
    # a progressive stream with its last scan repeated 600 times
    prog = _encode(synth.synthetic_page(1, 64, 64, lines=2, columns=1), quality=80, progressive=True)
    last, eoi = prog.rindex(b"\xff\xda"), prog.rindex(b"\xff\xd9")
    with pytest.raises(_lib.OcrsError, match="too many scans"):
        _lib.jpeg_coefficients(prog[:eoi] + prog[last:eoi] * 600 + b"\xff\xd9")
    assert _lib.jpeg_coefficients(prog[:eoi] + prog[last:eoi] * 3 + b"\xff\xd9") is not None   # a few repeats are legal refinements or harmless


@pytest.mark.gpu
def test_gpu_decode_equals_pil_and_the_oracle_in_every_mode():
    _lib.require_gpu()
    bad, total_px, total_coef = [], 0, 0
    for name, data in CASES:
        rgb, coef_bytes = _lib.jpeg_decode_rgb(data)
        total_px += rgb.shape[0] * rgb.shape[1]
        total_coef += coef_bytes
        if not np.array_equal(rgb, _pil(data)):
            bad.append(name)
    assert not bad, bad[:10]


@pytest.mark.gpu
def test_gpu_decode_of_the_reference_jpeg_and_of_bench_pages():
    _lib.require_gpu()
    g = np.load(os.path.join(G, "rust-book.npz"))
    data = g["file_bytes"].tobytes()
    rgb, coef_bytes = _lib.jpeg_decode_rgb(data)
    assert np.array_equal(rgb, g["pixels"]) and np.array_equal(rgb, _pil(data))
    assert coef_bytes < rgb.size                 # fewer bytes crossed PCIe than the decoded pixels would have
    for seed in range(16):                       # the 16 bench pages, 4:2:0 quality 90 (what scanners write)
        px = synth.synthetic_page(seed, 1024, 1024, lines=80)
        data = _encode(px, quality=90, subsampling=2)
        rgb, coef_bytes = _lib.jpeg_decode_rgb(data)
        assert np.array_equal(rgb, _pil(data)), seed
        assert coef_bytes < 0.5 * rgb.size, (seed, coef_bytes)


@pytest.mark.gpu
def test_prepare_input_jpeg_gives_the_page_of_prepare_input_on_the_decoded_pixels_and_the_golden_text():
    """The drop-in: file bytes in, OcrInput out.  Same grey page bits as prepare_input(into_rgb8(decode(file))), and the
    whole pipeline on rust-book.jpg from its bytes reproduces the oracle's golden fixture (made from PIL's pixels)."""
    from test_golden import _ref_fixture
    from ocrs_amd import DimOrder, ImageSource, Model, OcrEngine
    g, dbuf, rbuf = _ref_fixture("rust-book")
    eng = OcrEngine(detection_model=Model.load_bytes(dbuf), recognition_model=Model.load_bytes(rbuf))
    data = g["file_bytes"].tobytes()
    inp, coef_bytes = eng.prepare_input_jpeg(data)
    ref = eng.prepare_input(ImageSource.from_tensor(np.ascontiguousarray(g["pixels"]), DimOrder.Hwc))
    assert inp.shape == ref.shape == (1, 1600, 1200)
    assert np.array_equal(inp.image(), ref.image())
    words = eng.detect_words(inp)
    assert np.array_equal(words, g["word_rects"])
    lines = eng.find_text_lines(inp, words)
    assert np.array_equal(np.concatenate(lines), g["line_rects"])
    assert eng.get_text(inp) == str(g["text"][0])
    with pytest.raises(ocrs_amd.OcrsError):
        eng.prepare_input_jpeg(b"\x89PNG not a jpeg")


# ------------------------------------------------------------------------------------------------ flavours PIL cannot write
def _parse_dht(data):
    """{(class, id): (bits[16], vals)} of a JPEG stream's DHT segments."""
    out, i = {}, 2
    while i < len(data):
        if data[i] != 0xFF:
            i += 1
            continue
        m = data[i + 1]
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            i += 2
            continue
        L = (data[i + 2] << 8) | data[i + 3]
        if m == 0xC4:
            s, e = i + 4, i + 2 + L
            while s < e:
                tc, th = data[s] >> 4, data[s] & 15
                bits = list(data[s + 1:s + 17])
                n = sum(bits)
                out[(tc, th)] = (bits, list(data[s + 17:s + 17 + n]))
                s += 17 + n
        if m == 0xDA:
            break
        i += 2 + L
    return out


def _codes(bits, vals):
    code, k, table = 0, 0, {}
    for l in range(1, 17):
        for _ in range(bits[l - 1]):
            table[vals[k]] = (code, l)
            code += 1
            k += 1
        code <<= 1
    return table


class _BitWriter:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, value, length):
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49,
          56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _reencode_non_interleaved(geom, quant, coef, dht, restart=0, dqt16=True):
    """A baseline JPEG with ONE SCAN PER COMPONENT (each walks its own block grid), optional restart intervals and 16-bit
    quantisation tables, from decoded coefficients — stream flavours no PIL option produces."""
    width, height, ncomp, hmax, vmax = [int(v) for v in geom[:5]]
    comps = [[int(v) for v in geom[7 + 7 * i:14 + 7 * i]] for i in range(ncomp)]   # h, v, tq, cw, ch, bw, bh
    out = bytearray(b"\xff\xd8")
    out += b"\xff\xe0\x00\x10JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00"
    for tq in sorted(set(c[2] for c in comps)):
        q = [int(quant[tq * 64 + ZIGZAG[z]]) for z in range(64)]
        if dqt16:
            out += b"\xff\xdb" + (2 + 1 + 128).to_bytes(2, "big") + bytes([0x10 | tq]) + b"".join(v.to_bytes(2, "big") for v in q)
        else:
            out += b"\xff\xdb" + (2 + 1 + 64).to_bytes(2, "big") + bytes([tq]) + bytes(q)
    out += b"\xff\xc0" + (8 + 3 * ncomp).to_bytes(2, "big") + b"\x08" + height.to_bytes(2, "big") + width.to_bytes(2, "big") + bytes([ncomp])
    for i, c in enumerate(comps):
        out += bytes([i + 1, (c[0] << 4) | c[1], c[2]])
    for (tc, th), (bits, vals) in sorted(dht.items()):
        out += b"\xff\xc4" + (2 + 17 + len(vals)).to_bytes(2, "big") + bytes([(tc << 4) | th]) + bytes(bits) + bytes(vals)
    if restart:
        out += b"\xff\xdd\x00\x04" + restart.to_bytes(2, "big")
    first = 0
    for i, c in enumerate(comps):
        _h, _v, _tq, cw, ch, bw, bh = c
        blocks = coef[first:first + bw * bh].reshape(bh, bw, 64)
        first += bw * bh
        t = 0 if i == 0 else min(1, max(k[1] for k in dht))
        dc, ac = _codes(*dht[(0, t)]), _codes(*dht[(1, t)])
        out += b"\xff\xda\x00\x08\x01" + bytes([i + 1, (t << 4) | t]) + b"\x00\x3f\x00"
        w, pred, n_mcu, rst = _BitWriter(), 0, 0, 0
        for by in range((ch + 7) // 8):
            for bx in range((cw + 7) // 8):
                if restart and n_mcu and n_mcu % restart == 0:
                    w.flush()
                    w.out += bytes([0xFF, 0xD0 + (rst & 7)])
                    rst += 1
                    pred = 0
                b = blocks[by, bx]
                d = int(b[0]) - pred
                pred = int(b[0])
                s = abs(d).bit_length()
                w.put(*dc[s])
                if s:
                    w.put(d if d > 0 else d + (1 << s) - 1, s)
                run = 0
                last = max([z for z in range(1, 64) if b[ZIGZAG[z]]] or [0])
                for z in range(1, last + 1):
                    v = int(b[ZIGZAG[z]])
                    if v == 0:
                        run += 1
                        continue
                    while run > 15:
                        w.put(*ac[0xF0])
                        run -= 16
                    s = abs(v).bit_length()
                    w.put(*ac[(run << 4) | s])
                    w.put(v if v > 0 else v + (1 << s) - 1, s)
                    run = 0
                if last < 63:
                    w.put(*ac[0x00])
                n_mcu += 1
        w.flush()
        out += w.out
    out += b"\xff\xd9"
    return bytes(out)


REENC = [(2, 0, True), (1, 7, False), (0, 3, True), (2, 11, True)]


def _check_reencoded(ss, restart, dqt16, gpu):
    px = synth.synthetic_page(9, 203, 157, lines=6, columns=1)
    src = _encode(px, quality=85, subsampling=ss)
    geom, quant, coef = _lib.jpeg_coefficients(src)
    data = _reencode_non_interleaved(geom, quant, coef, _parse_dht(src), restart=restart, dqt16=dqt16)
    ref = _pil(data)
    assert np.array_equal(ref, _pil(src))            # the re-encoding is lossless: libjpeg sees the same image
    if gpu:
        rgb, _ = _lib.jpeg_decode_rgb(data)
        assert np.array_equal(rgb, ref)
    else:
        g2, q2, c2 = _lib.jpeg_coefficients(data)
        assert np.array_equal(OJ.decode_from_coefficients(g2, q2, c2), ref)


@pytest.mark.parametrize("ss,restart,dqt16", REENC)
def test_single_component_scans_restart_intervals_and_16_bit_tables(ss, restart, dqt16):
    """Baseline streams with one scan per component (each walking only the blocks that cover ITS plane, not the padded MCU
    grid), restart intervals inside such scans and 16-bit DQT segments: re-encoded here from decoded coefficients, decoded by
    libjpeg (PIL) and by the host decoder + oracle: the same pixels."""
    _check_reencoded(ss, restart, dqt16, gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("ss,restart,dqt16", REENC)
def test_gpu_decode_of_single_component_scans_restart_intervals_and_16_bit_tables(ss, restart, dqt16):
    _lib.require_gpu()
    _check_reencoded(ss, restart, dqt16, gpu=True)
