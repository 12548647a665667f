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
