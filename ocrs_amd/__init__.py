"""ocrs_amd — MI355X-native engine behind the robertknight/ocrs `OcrEngine` API.

Python mirror (ctypes) of the reference's public surface (ocrs/src/lib.rs:29-31,
111-301): `OcrEngine`, `OcrEngineParams`, `ImageSource`, `DimOrder`,
`DecodeMethod`, `TextLine`/`TextWord`/`TextChar`, plus `Model` for the
`trait Model` seam (ocrs/src/model.rs:6-17).  All compute happens in
libocrs_amd.so (HIP, gfx950); this module only marshals arrays.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import OcrsError, check, lib

__all__ = ["OcrEngine", "OcrEngineParams", "ImageSource", "ImageSourceError", "DimOrder", "DecodeMethod", "Model",
           "OcrInput", "TextLine", "TextWord", "TextChar", "OcrsError", "DEFAULT_ALPHABET", "EngineGroup"]

# lib.rs:34 (with the EUR sign the comment at lib.rs:33 asks for)
DEFAULT_ALPHABET = " 0123456789!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~€ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"


class DimOrder:  # preprocess.rs:50-57
    Hwc = 0
    Chw = 1


class DecodeMethod:  # recognition.rs:198-205
    Greedy = ("greedy", 0)

    @staticmethod
    def BeamSearch(width):
        return ("beam", int(width))


class ImageSourceError(ValueError):  # preprocess.rs:38-46
    pass


class ImageSource:
    """preprocess.rs:61-124 — a borrowed view of u8 or f32 pixels, HWC or CHW."""

    def __init__(self, data, order):
        self.data = data
        self.order = order

    @staticmethod
    def from_bytes(buf, dimensions):
        width, height = dimensions
        ch = C.c_uint32(0)
        st = lib().ocrs_image_source_check_bytes(C.c_size_t(len(buf)), C.c_uint32(width), C.c_uint32(height), C.byref(ch))
        if st != 0:
            raise ImageSourceError(lib().ocrs_last_error().decode())
        arr = np.frombuffer(buf, dtype=np.uint8).reshape(height, width, ch.value)
        return ImageSource(arr, DimOrder.Hwc)

    @staticmethod
    def from_tensor(data, order):
        data = np.asarray(data)
        if data.ndim != 3:
            raise ImageSourceError("expected a 3-dimensional image tensor")
        chans = data.shape[2] if order == DimOrder.Hwc else data.shape[0]
        if chans not in (1, 3, 4):
            raise ImageSourceError("channel count is not 1, 3 or 4")
        if data.dtype not in (np.uint8, np.float32):
            raise ImageSourceError("pixels must be uint8 in [0,255] or float32 in [0,1]")
        return ImageSource(data, order)


class Model:
    """`trait Model` (model.rs:6-17): either an `.ocrsm` fixed graph executed by
    the HIP executor, or a Python callable (the reference's tests inject fake
    models through the same seam, lib.rs:339-422)."""

    def __init__(self, handle, keepalive=None):
        self._h = handle
        self._keep = keepalive

    @staticmethod
    def load_file(path):
        """`*.ocrsm` containers are read by the library; `*.onnx` files (the format the real
        ocrs models are published in, README.md:96-102) are lowered by `onnx_import` first."""
        if str(path).lower().endswith(".onnx"):
            from .onnx_import import import_onnx
            return Model.load_bytes(import_onnx(str(path)).to_bytes())
        h = C.c_void_p()
        check(lib().ocrs_model_load_file(str(path).encode(), C.byref(h)))
        return Model(h)

    @staticmethod
    def load_bytes(buf, device=None):
        """device=None: the process default (ocrs_set_device); an int places the weights on that HIP device."""
        h = C.c_void_p()
        if device is None:
            check(lib().ocrs_model_load_bytes(C.c_char_p(bytes(buf)), C.c_size_t(len(buf)), C.byref(h)))
        else:
            check(lib().ocrs_model_load_bytes_on_device(C.c_char_p(bytes(buf)), C.c_size_t(len(buf)), C.c_int(int(device)),
                                                        C.byref(h)))
        return Model(h)

    def device(self):
        d = C.c_int(-1)
        check(lib().ocrs_model_device(self._h, C.byref(d)))
        return d.value

    @staticmethod
    def from_callable(input_shape, fn):
        """input_shape: NCHW with None for symbolic dims; fn(np [N,C,H,W] f32) -> np (<= 4 dims)."""
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        libc.malloc.argtypes = [C.c_size_t]
        err = []

        def _run(user, inp, in_shape, out, out_shape, out_ndim):
            try:
                shp = [in_shape[i] for i in range(4)]
                x = np.ctypeslib.as_array(inp, shape=(int(np.prod(shp)),)).reshape(shp).copy()
                y = np.ascontiguousarray(fn(x), dtype=np.float32)
                if y.ndim < 1 or y.ndim > 4:
                    return 2
                p = libc.malloc(max(y.nbytes, 4))
                C.memmove(p, y.ctypes.data, y.nbytes)
                out[0] = C.cast(p, C.POINTER(C.c_float))
                for i, d in enumerate(y.shape):
                    out_shape[i] = d
                out_ndim[0] = y.ndim
                return 0
            except Exception as e:  # surfaces as ModelRunError::RunFailed
                err.append(e)
                return 1

        cb = _lib.RUN_FN(_run)
        shape = (C.c_int64 * 4)(*[-1 if d is None else int(d) for d in input_shape])
        h = C.c_void_p()
        check(lib().ocrs_model_from_callback(shape, cb, None, C.byref(h)))
        return Model(h, keepalive=(cb, err))

    def input_shape(self):
        dims = (C.c_int64 * 4)()
        fixed = (C.c_uint8 * 4)()
        check(lib().ocrs_model_input_shape(self._h, dims, fixed))
        return [int(dims[i]) if fixed[i] else None for i in range(4)]

    def run(self, nchw, timing=False):
        x = np.ascontiguousarray(nchw, dtype=np.float32)
        assert x.ndim == 4
        shape = (C.c_int64 * 4)(*x.shape)
        out = C.POINTER(C.c_float)()
        oshape = (C.c_int64 * 4)()
        ond = C.c_int(0)
        opts = _lib.RunOptions(1 if timing else 0)
        check(lib().ocrs_model_run(self._h, x.ctypes.data_as(C.POINTER(C.c_float)), shape, C.byref(opts), C.byref(out),
                                   oshape, C.byref(ond)))
        shp = [int(oshape[i]) for i in range(ond.value)]
        y = np.ctypeslib.as_array(out, shape=(int(np.prod(shp)),)).reshape(shp).copy()
        lib().ocrs_buffer_free(out)
        return y

    def flops(self, n, h, w):
        shape = (C.c_int64 * 4)(n, 1, h, w)
        f = C.c_double(0)
        check(lib().ocrs_model_flops(self._h, shape, C.byref(f)))
        return f.value

    def __del__(self):
        try:
            if self._h:
                lib().ocrs_model_free(self._h)
                self._h = None
        except Exception:
            pass


NUMERICS = {"exact": 0, "relaxed": 1, "reduced": 2}   # ocrs_numerics


class OcrEngineParams:  # lib.rs:38-71
    def __init__(self, detection_model=None, recognition_model=None, debug=False, decode_method=DecodeMethod.Greedy,
                 alphabet=None, allowed_chars=None, numerics="exact", coalesce=0, coalesce_pages=0, coalesce_window_us=0,
                 layout_threads=0, rec_max_pixels=0, options=None):
        # numerics / coalesce*: ocrs_engine_params fields without a reference counterpart (0 = default); while an engine with
        # numerics != "exact" is alive, every call on its device runs its kernels one at a time (include/ocrs_amd.h);
        # options: {name: value} applied to the new engine with ocrs_engine_set_option
        self.numerics = numerics
        self.coalesce = coalesce
        self.coalesce_pages = coalesce_pages
        self.coalesce_window_us = coalesce_window_us
        self.layout_threads = layout_threads
        self.rec_max_pixels = rec_max_pixels
        self.options = dict(options or {})
        self.detection_model = detection_model
        self.recognition_model = recognition_model
        self.debug = debug
        self.decode_method = decode_method
        self.alphabet = alphabet
        self.allowed_chars = allowed_chars


class OcrInput:
    """lib.rs:125-128 — the prepared grey page, resident in HBM."""

    def __init__(self, handle):
        self._h = handle

    @property
    def shape(self):
        h, w = C.c_int(0), C.c_int(0)
        check(lib().ocrs_page_dims(self._h, C.byref(h), C.byref(w)))
        return (1, h.value, w.value)

    def image(self):
        _, h, w = self.shape
        out = np.empty((1, h, w), np.float32)
        check(lib().ocrs_page_image(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def __del__(self):
        try:
            if self._h:
                lib().ocrs_page_free(self._h)
                self._h = None
        except Exception:
            pass


class TextChar:  # text_items.rs:48-54
    __slots__ = ("char", "rect")

    def __init__(self, char, rect):
        self.char = char
        self.rect = rect  # (top, left, bottom, right)


class _TextItem:
    def __init__(self, chars):
        self._chars = chars

    def chars(self):
        return self._chars

    def bounding_rect(self):
        a = np.array([c.rect for c in self._chars])
        return (int(a[:, 0].min()), int(a[:, 1].min()), int(a[:, 2].max()), int(a[:, 3].max()))

    def rotated_rect(self):
        """text_items.rs:18-30 -> (center.x, center.y, up.x, up.y, width, height)."""
        a = np.ascontiguousarray(np.array([c.rect for c in self._chars], np.int32).reshape(-1, 4))
        out = (C.c_float * 6)()
        check(lib().ocrs_text_item_rotated_rect(a.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(len(a)), out))
        return np.array(list(out), np.float32)

    def __str__(self):
        return "".join(c.char for c in self._chars)


def rotated_rect_corners(rect6):
    """RotatedRect::corners -> [[x, y] x 4] floats."""
    r = (C.c_float * 6)(*[float(v) for v in rect6])
    out = (C.c_float * 8)()
    check(lib().ocrs_rotated_rect_corners(r, out))
    return [[out[2 * i], out[2 * i + 1]] for i in range(4)]


class TextWord(_TextItem):  # text_items.rs:92-107
    pass


class TextLine(_TextItem):  # text_items.rs:61-82
    def __init__(self, chars):
        assert chars, "Text lines must not be empty"
        super().__init__(chars)

    def words(self):
        out, cur = [], []
        for c in self._chars:
            if c.char == " ":
                if cur:
                    out.append(TextWord(cur))
                cur = []
            else:
                cur.append(c)
        if cur:
            out.append(TextWord(cur))
        return out


def _rects_to_array(rects):
    a = np.ascontiguousarray(np.asarray(rects, dtype=np.float32).reshape(-1, 6))
    return a


def _pack_lines(lines):
    offs = [0]
    flat = []
    for l in lines:
        a = _rects_to_array(l)
        flat.append(a)
        offs.append(offs[-1] + len(a))
    rects = np.concatenate(flat) if flat else np.zeros((0, 6), np.float32)
    return np.ascontiguousarray(rects, np.float32), np.array(offs, dtype=np.uintp)


class OcrEngine:
    """lib.rs:111-301.  Word rects are rows of 6 floats
    (center.x, center.y, up.x, up.y, width, height)."""

    def __init__(self, params=None, **kw):
        params = params or OcrEngineParams(**kw)
        self._params = params
        p = _lib.EngineParams()
        p.detection_model = params.detection_model._h if params.detection_model else None
        p.recognition_model = params.recognition_model._h if params.recognition_model else None
        p.debug = 1 if params.debug else 0
        p.decode_method = 0 if params.decode_method[0] == "greedy" else 1
        p.beam_width = params.decode_method[1]
        p.alphabet = params.alphabet.encode("utf-8") if params.alphabet is not None else None
        p.allowed_chars = params.allowed_chars.encode("utf-8") if params.allowed_chars is not None else None
        p.numerics = NUMERICS[params.numerics]
        p.coalesce = int(params.coalesce)
        p.coalesce_pages = int(params.coalesce_pages)
        p.coalesce_window_us = int(params.coalesce_window_us)
        p.layout_threads = int(params.layout_threads)
        p.rec_max_pixels = int(params.rec_max_pixels)
        self._h = C.c_void_p()
        check(lib().ocrs_engine_new(C.byref(p), C.byref(self._h)))
        for k, v in params.options.items():
            self.set_option(k, v)

    def set_option(self, name, value):
        """ocrs_engine_set_option: this engine's copy of a tuning option (results never depend on one)."""
        check(lib().ocrs_engine_set_option(self._h, name.encode(), C.c_long(int(value))))

    def get_option(self, name):
        v = C.c_long(0)
        check(lib().ocrs_engine_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    @classmethod
    def _borrowed(cls, handle, keep):
        """An engine owned by somebody else (an EngineGroup member): same methods, never freed here."""
        self = cls.__new__(cls)
        self._params = None
        self._h = handle
        self._keep = keep
        self._owned = False
        return self

    def device(self):
        d = C.c_int(-1)
        check(lib().ocrs_engine_device(self._h, C.byref(d)))
        return d.value

    def coalesce_stats(self):
        """{stage: (merged batches run, caller requests they carried)} — ocrs_engine_coalesce_stats."""
        det, rec = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
        check(lib().ocrs_engine_coalesce_stats(self._h, det, rec))
        return {"detect": (int(det[0]), int(det[1])), "recognize": (int(rec[0]), int(rec[1]))}

    # ---- lib.rs:183-187
    def prepare_input(self, image):
        a = np.ascontiguousarray(image.data)
        if image.order == DimOrder.Hwc:
            h, w, c = a.shape
        else:
            c, h, w = a.shape
        h_out = C.c_void_p()
        check(lib().ocrs_engine_prepare_input(self._h, a.ctypes.data_as(C.c_void_p), 0 if a.dtype == np.uint8 else 1,
                                              image.order, h, w, c, C.byref(h_out)))
        return OcrInput(h_out)

    def prepare_input_batch_raw(self, host_ptrs, dtype, order, h, w, c):
        """n equally sized host images given as raw pointers (e.g. pinned buffers from ocrs_host_malloc): one call,
        one wait."""
        n = len(host_ptrs)
        arr = (C.c_void_p * n)(*host_ptrs)
        out = (C.c_void_p * n)()
        check(lib().ocrs_engine_prepare_input_batch(self._h, arr, C.c_size_t(n), 0 if dtype == np.uint8 else 1, order,
                                                    h, w, c, out))
        return [OcrInput(C.c_void_p(out[i])) for i in range(n)]

    def prepare_input_device(self, d_ptr, dtype, order, h, w, c):
        """Pixels already in HBM (a raw device pointer), e.g. from bench.py."""
        h_out = C.c_void_p()
        check(lib().ocrs_engine_prepare_input_device(self._h, C.c_void_p(d_ptr), 0 if dtype == np.uint8 else 1, order,
                                                     h, w, c, C.byref(h_out)))
        return OcrInput(h_out)

    def prepare_input_jpeg(self, data):
        """prepare_input fed with a JPEG file's bytes (ocrs-cli/src/main.rs:312-333 decodes on the host first): Huffman
        decoding on the host, IDCT / upsampling / colour conversion / grey conversion on the GPU.  -> (OcrInput, bytes that
        crossed PCIe).  Raises OcrsError (IMAGE_SOURCE) for flavours the hand-off does not cover."""
        h_out = C.c_void_p()
        cb = C.c_size_t(0)
        buf = C.create_string_buffer(bytes(data), len(data))
        check(lib().ocrs_engine_prepare_input_jpeg(self._h, buf, C.c_size_t(len(data)), C.byref(h_out), C.byref(cb)))
        return OcrInput(h_out), cb.value

    # ---- lib.rs:193-199
    def detect_words(self, inp):
        rects = C.POINTER(C.c_float)()
        n = C.c_size_t(0)
        check(lib().ocrs_engine_detect_words(self._h, inp._h, C.byref(rects), C.byref(n)))
        out = np.ctypeslib.as_array(rects, shape=(max(n.value, 1) * 6,))[: n.value * 6].reshape(-1, 6).copy()
        lib().ocrs_buffer_free(rects)
        return out

    def detect_words_batch(self, inputs):
        n = len(inputs)
        pages = (C.c_void_p * n)(*[i._h for i in inputs])
        rects = C.POINTER(C.c_float)()
        offs = (C.c_size_t * (n + 1))()
        check(lib().ocrs_engine_detect_words_batch(self._h, pages, C.c_size_t(n), C.byref(rects), offs))
        total = offs[n]
        flat = np.ctypeslib.as_array(rects, shape=(max(total, 1) * 6,))[: total * 6].reshape(-1, 6).copy()
        lib().ocrs_buffer_free(rects)
        return [flat[offs[i]:offs[i + 1]] for i in range(n)]

    # ---- lib.rs:207-213
    def detect_text_pixels(self, inp):
        _, h, w = inp.shape
        out = np.empty((h, w), np.float32)
        check(lib().ocrs_engine_detect_text_pixels(self._h, inp._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    # ---- lib.rs:222-228
    def find_text_lines(self, inp, words):
        a = _rects_to_array(words)
        lr = C.POINTER(C.c_float)()
        lo = C.POINTER(C.c_size_t)()
        nl = C.c_size_t(0)
        check(lib().ocrs_engine_find_text_lines(self._h, inp._h if inp is not None else None,
                                                a.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(len(a)), C.byref(lr),
                                                C.byref(lo), C.byref(nl)))
        offs = [lo[i] for i in range(nl.value + 1)]
        flat = np.ctypeslib.as_array(lr, shape=(max(len(a), 1) * 6,))[: len(a) * 6].reshape(-1, 6).copy()
        lib().ocrs_buffer_free(lr)
        lib().ocrs_buffer_free(lo)
        return [flat[offs[i]:offs[i + 1]] for i in range(nl.value)]

    def find_text_lines_batch_raw(self, words_per_page):
        """Threaded over pages.  Returns (rects [n,6], line_offsets, page_line_offsets) as numpy arrays."""
        n = len(words_per_page)
        woffs = np.zeros(n + 1, dtype=np.uintp)
        for i, w in enumerate(words_per_page):
            woffs[i + 1] = woffs[i] + len(w)
        allw = np.ascontiguousarray(np.concatenate([_rects_to_array(w) for w in words_per_page]) if n else np.zeros((0, 6), np.float32))
        lr = C.POINTER(C.c_float)()
        lo = C.POINTER(C.c_size_t)()
        po = C.POINTER(C.c_size_t)()
        check(lib().ocrs_engine_find_text_lines_batch(self._h, C.c_size_t(n), allw.ctypes.data_as(C.POINTER(C.c_float)),
                                                      woffs.ctypes.data_as(C.POINTER(C.c_size_t)), C.byref(lr), C.byref(lo),
                                                      C.byref(po)))
        poffs = np.ctypeslib.as_array(po, shape=(n + 1,)).astype(np.uintp)
        nl = int(poffs[n])
        loffs = np.ctypeslib.as_array(lo, shape=(nl + 1,)).astype(np.uintp)
        rects = np.ctypeslib.as_array(lr, shape=(max(len(allw), 1) * 6,))[: len(allw) * 6].reshape(-1, 6).copy()
        for p in (lr, lo, po):
            lib().ocrs_buffer_free(p)
        return rects, loffs, poffs

    def recognize_text_batch_raw(self, inputs, rects, line_offsets, page_line_offsets):
        """Packed form of recognize_text_batch: returns (chars, char_offsets) where chars is a
        structured array (ch, top, left, bottom, right) and line i owns chars[char_offsets[i]:char_offsets[i+1]]."""
        n = len(inputs)
        pages = (C.c_void_p * n)(*[i._h for i in inputs])
        rects = np.ascontiguousarray(rects, np.float32)
        lo = np.ascontiguousarray(line_offsets, np.uintp)
        po = np.ascontiguousarray(page_line_offsets, np.uintp)
        nl = len(lo) - 1
        chars = C.POINTER(_lib.TextCharC)()
        coffs = C.POINTER(C.c_size_t)()
        check(lib().ocrs_engine_recognize_text_batch(
            self._h, pages, C.c_size_t(n), po.ctypes.data_as(C.POINTER(C.c_size_t)),
            rects.ctypes.data_as(C.POINTER(C.c_float)), lo.ctypes.data_as(C.POINTER(C.c_size_t)), C.c_size_t(nl),
            C.byref(chars), C.byref(coffs)))
        co = np.ctypeslib.as_array(coffs, shape=(nl + 1,)).astype(np.uintp)
        total = int(co[nl])
        dt = np.dtype([("ch", np.uint32), ("top", np.int32), ("left", np.int32), ("bottom", np.int32), ("right", np.int32)])
        if total:
            buf = (C.c_char * (total * dt.itemsize)).from_address(C.addressof(chars.contents))
            arr = np.frombuffer(buf, dtype=dt, count=total).copy()
        else:
            arr = np.zeros(0, dt)
        lib().ocrs_buffer_free(chars)
        lib().ocrs_buffer_free(coffs)
        return arr, co

    # ---- lib.rs:237-256
    def recognize_text(self, inp, lines):
        """OcrEngine::recognize_text (lib.rs:237-256) through the single-page entry point a Rust binding uses."""
        rects, offs = _pack_lines(lines)
        chars = C.POINTER(_lib.TextCharC)()
        coffs = C.POINTER(C.c_size_t)()
        check(lib().ocrs_engine_recognize_text(
            self._h, inp._h, rects.ctypes.data_as(C.POINTER(C.c_float)), offs.ctypes.data_as(C.POINTER(C.c_size_t)),
            C.c_size_t(len(lines)), C.byref(chars), C.byref(coffs)))
        out = []
        for li in range(len(lines)):
            a, b = coffs[li], coffs[li + 1]
            out.append(TextLine([TextChar(chr(chars[k].ch), (chars[k].top, chars[k].left, chars[k].bottom, chars[k].right))
                                 for k in range(a, b)]) if b > a else None)
        lib().ocrs_buffer_free(chars)
        lib().ocrs_buffer_free(coffs)
        return out

    def recognize_text_batch(self, inputs, lines_per_page):
        n = len(inputs)
        pages = (C.c_void_p * n)(*[i._h for i in inputs])
        all_lines = [l for lines in lines_per_page for l in lines]
        plo = [0]
        for lines in lines_per_page:
            plo.append(plo[-1] + len(lines))
        rects, offs = _pack_lines(all_lines)
        plo_a = np.array(plo, dtype=np.uintp)
        chars = C.POINTER(_lib.TextCharC)()
        coffs = C.POINTER(C.c_size_t)()
        check(lib().ocrs_engine_recognize_text_batch(
            self._h, pages, C.c_size_t(n), plo_a.ctypes.data_as(C.POINTER(C.c_size_t)),
            rects.ctypes.data_as(C.POINTER(C.c_float)), offs.ctypes.data_as(C.POINTER(C.c_size_t)),
            C.c_size_t(len(all_lines)), C.byref(chars), C.byref(coffs)))
        result = []
        li = 0
        for lines in lines_per_page:
            page_out = []
            for _ in lines:
                a, b = coffs[li], coffs[li + 1]
                if b > a:
                    page_out.append(TextLine([TextChar(chr(chars[k].ch), (chars[k].top, chars[k].left, chars[k].bottom,
                                                                           chars[k].right)) for k in range(a, b)]))
                else:
                    page_out.append(None)
                li += 1
            result.append(page_out)
        lib().ocrs_buffer_free(chars)
        lib().ocrs_buffer_free(coffs)
        return result

    def recognize_tokens(self, inp, lines):
        """Raw greedy-CTC output per line: list of (label, pos) — CtcHypothesis::steps()."""
        rects, offs = _pack_lines(lines)
        lab = C.POINTER(C.c_uint32)()
        pos = C.POINTER(C.c_uint32)()
        toff = C.POINTER(C.c_size_t)()
        check(lib().ocrs_engine_recognize_tokens(self._h, inp._h, rects.ctypes.data_as(C.POINTER(C.c_float)),
                                                 offs.ctypes.data_as(C.POINTER(C.c_size_t)), C.c_size_t(len(lines)),
                                                 C.byref(lab), C.byref(pos), C.byref(toff)))
        out = []
        for i in range(len(lines)):
            out.append([(int(lab[k]), int(pos[k])) for k in range(toff[i], toff[i + 1])])
        for p in (lab, pos, toff):
            lib().ocrs_buffer_free(p)
        return out

    def recognize_logits(self, inp, lines):
        """The recognition model's log-probabilities per line, [T_i, classes] each (ocrs_engine_recognize_logits)."""
        rects, offs = _pack_lines(lines)
        lp = C.POINTER(C.c_float)()
        roff = C.POINTER(C.c_size_t)()
        ncls = C.c_int(0)
        check(lib().ocrs_engine_recognize_logits(self._h, inp._h, rects.ctypes.data_as(C.POINTER(C.c_float)),
                                                 offs.ctypes.data_as(C.POINTER(C.c_size_t)), C.c_size_t(len(lines)),
                                                 C.byref(lp), C.byref(roff), C.byref(ncls)))
        c = ncls.value
        out = []
        for i in range(len(lines)):
            a, b = roff[i], roff[i + 1]
            out.append(np.ctypeslib.as_array(lp, shape=(roff[len(lines)] * c,))[a * c:b * c].reshape(b - a, c).copy() if b > a
                       else np.zeros((0, c), np.float32))
        lib().ocrs_buffer_free(lp)
        lib().ocrs_buffer_free(roff)
        return out

    # ---- lib.rs:268-278
    def prepare_recognition_input(self, inp, line):
        a = _rects_to_array(line)
        out = C.POINTER(C.c_float)()
        h, w = C.c_int(0), C.c_int(0)
        check(lib().ocrs_engine_prepare_recognition_input(self._h, inp._h, a.ctypes.data_as(C.POINTER(C.c_float)),
                                                          C.c_size_t(len(a)), C.byref(out), C.byref(h), C.byref(w)))
        img = np.ctypeslib.as_array(out, shape=(max(h.value * w.value, 1),))[: h.value * w.value].reshape(h.value, w.value).copy()
        lib().ocrs_buffer_free(out)
        return img

    # ---- lib.rs:282-287
    def detection_threshold(self):
        return float(lib().ocrs_engine_detection_threshold(self._h))

    # ---- lib.rs:290-300
    def get_text(self, inp):
        txt = C.c_char_p()
        check(lib().ocrs_engine_get_text(self._h, inp._h, C.byref(txt)))
        s = txt.value.decode("utf-8")
        lib().ocrs_buffer_free(txt)
        return s

    # ---- measurement hooks
    def enable_timing(self, level=1):
        """0 off, 1 per-stage HIP-event timers, 2 also per-launch kernel-class timers."""
        check(lib().ocrs_engine_enable_timing(self._h, int(level)))

    def set_kernel_timing_classes(self, names=None):
        """Restrict per-launch kernel timing to the named classes (None = all)."""
        n = lib().ocrs_kernel_class_count()
        all_names = [lib().ocrs_kernel_class_name(i).decode() for i in range(n)]
        mask = 0xFFFFFFFF if names is None else sum(1 << all_names.index(x) for x in names)
        check(lib().ocrs_engine_set_kernel_timing_mask(self._h, C.c_uint32(mask)))

    def kernel_stats(self, reset=True):
        n = lib().ocrs_kernel_class_count()
        ms = (C.c_double * n)()
        cnt = (C.c_uint64 * n)()
        fl = (C.c_double * n)()
        by = (C.c_double * n)()
        mf = (C.c_double * n)()
        check(lib().ocrs_engine_kernel_mfma_flops(self._h, mf))
        check(lib().ocrs_engine_kernel_stats(self._h, ms, cnt, fl, by, 1 if reset else 0))
        return {lib().ocrs_kernel_class_name(i).decode(): dict(ms=ms[i], launches=int(cnt[i]), flops=fl[i], bytes=by[i],
                                                               mfma_flops=mf[i])
                for i in range(n)}

    def stage_times(self, reset=True):
        n = lib().ocrs_stage_count()
        ms = (C.c_double * n)()
        cnt = (C.c_uint64 * n)()
        check(lib().ocrs_engine_stage_times(self._h, ms, cnt, 1 if reset else 0))
        return {lib().ocrs_stage_name(i).decode(): (ms[i], int(cnt[i])) for i in range(n)}

    def __del__(self):
        try:
            if self._h and getattr(self, "_owned", True):
                lib().ocrs_engine_free(self._h)
            self._h = None
        except Exception:
            pass


def _chars_from_c(chars, coffs, nl):
    co = np.ctypeslib.as_array(coffs, shape=(nl + 1,)).astype(np.uintp)
    total = int(co[nl])
    dt = np.dtype([("ch", np.uint32), ("top", np.int32), ("left", np.int32), ("bottom", np.int32), ("right", np.int32)])
    if total:
        buf = (C.c_char * (total * dt.itemsize)).from_address(C.addressof(chars.contents))
        arr = np.frombuffer(buf, dtype=dt, count=total).copy()
    else:
        arr = np.zeros(0, dt)
    lib().ocrs_buffer_free(chars)
    lib().ocrs_buffer_free(coffs)
    return arr, co


class EngineGroup:
    """Several GPUs behind one handle in one process (include/ocrs_amd.h "engine group"): a page is processed by a
    member of the device it lives on; pages the group places itself go out in contiguous blocks.  `devices` may repeat
    a device (members then share it; RCCL refuses such a communicator and the gathers use the host transport)."""

    GATHER = {"auto": 0, "host": 1, "rccl": 2}

    def __init__(self, devices, detection_bytes=None, recognition_bytes=None, debug=False, decode_method=DecodeMethod.Greedy,
                 alphabet=None, allowed_chars=None, gather="auto", numerics="exact", coalesce=0, coalesce_pages=0,
                 coalesce_window_us=0, layout_threads=0, rec_max_pixels=0, min_block=0, shared_block=0):
        p = _lib.GroupParams()
        p.numerics = NUMERICS[numerics]
        p.coalesce, p.coalesce_pages, p.coalesce_window_us = int(coalesce), int(coalesce_pages), int(coalesce_window_us)
        p.layout_threads, p.rec_max_pixels = int(layout_threads), int(rec_max_pixels)
        p.min_block, p.shared_block = int(min_block), int(shared_block)
        self._det = bytes(detection_bytes) if detection_bytes is not None else None
        self._rec = bytes(recognition_bytes) if recognition_bytes is not None else None
        p.detection_model = C.cast(C.c_char_p(self._det), C.c_void_p) if self._det else None
        p.detection_model_len = len(self._det) if self._det else 0
        p.recognition_model = C.cast(C.c_char_p(self._rec), C.c_void_p) if self._rec else None
        p.recognition_model_len = len(self._rec) if self._rec else 0
        self._devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        p.devices = self._devs
        p.n_devices = len(devices)
        p.debug = 1 if debug else 0
        p.decode_method = 0 if decode_method[0] == "greedy" else 1
        p.beam_width = decode_method[1]
        p.alphabet = alphabet.encode("utf-8") if alphabet is not None else None
        p.allowed_chars = allowed_chars.encode("utf-8") if allowed_chars is not None else None
        p.gather = self.GATHER[gather]
        self._h = C.c_void_p()
        check(lib().ocrs_engine_group_new(C.byref(p), C.byref(self._h)))
        self.devices = [int(d) for d in devices]

    def __len__(self):
        n = C.c_size_t(0)
        check(lib().ocrs_engine_group_size(self._h, C.byref(n)))
        return n.value

    def member(self, i):
        """(OcrEngine view of member i, its device)"""
        e, d = C.c_void_p(), C.c_int(-1)
        check(lib().ocrs_engine_group_member(self._h, C.c_size_t(i), C.byref(e), C.byref(d)))
        return OcrEngine._borrowed(e, self), d.value

    def member_stats(self, i):
        """ocrs_group_member_stats as a dict."""
        v = (C.c_uint64 * 8)()
        check(lib().ocrs_group_member_stats(self._h, C.c_size_t(i), v))
        return {"device": int(v[7]), "shares": int(v[0]), "pages": int(v[1]), "host_cpu_s": v[2] / 1e9, "busy_wall_s": v[3] / 1e9,
                "numa_node": int(v[4]) - 1 if v[4] else None, "node_cpus": int(v[5]), "bound_shares": int(v[6])}

    def set_option(self, name, value):
        """ocrs_engine_set_option on every member."""
        for i in range(len(self)):
            self.member(i)[0].set_option(name, value)

    def last_gather(self):
        t, b, why = C.c_int(0), C.c_size_t(0), C.c_char_p()
        check(lib().ocrs_group_last_gather(self._h, C.byref(t), C.byref(b), C.byref(why)))
        return {"transport": {0: None, 1: "host", 2: "rccl"}[t.value], "bytes": b.value,
                "why_host": (why.value or b"").decode()}

    def prepare_input_batch(self, images, order=DimOrder.Hwc):
        """images: equally shaped numpy arrays (host)."""
        arrs = [np.ascontiguousarray(a) for a in images]
        a0 = arrs[0]
        if order == DimOrder.Hwc:
            h, w, c = a0.shape
        else:
            c, h, w = a0.shape
        n = len(arrs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        out = (C.c_void_p * n)()
        check(lib().ocrs_group_prepare_input_batch(self._h, ptrs, C.c_size_t(n), 0 if a0.dtype == np.uint8 else 1, order, h, w, c, out))
        return [OcrInput(C.c_void_p(out[i])) for i in range(n)]

    def prepare_input_device_batch(self, d_ptrs, dtype, order, h, w, c):
        n = len(d_ptrs)
        ptrs = (C.c_void_p * n)(*d_ptrs)
        out = (C.c_void_p * n)()
        check(lib().ocrs_group_prepare_input_device_batch(self._h, ptrs, C.c_size_t(n), 0 if dtype == np.uint8 else 1, order,
                                                          h, w, c, out))
        return [OcrInput(C.c_void_p(out[i])) for i in range(n)]

    def detect_words_batch(self, inputs):
        n = len(inputs)
        pages = (C.c_void_p * n)(*[i._h for i in inputs])
        rects = C.POINTER(C.c_float)()
        offs = (C.c_size_t * (n + 1))()
        check(lib().ocrs_group_detect_words_batch(self._h, pages, C.c_size_t(n), C.byref(rects), offs))
        total = offs[n]
        flat = np.ctypeslib.as_array(rects, shape=(max(total, 1) * 6,))[: total * 6].reshape(-1, 6).copy()
        lib().ocrs_buffer_free(rects)
        return [flat[offs[i]:offs[i + 1]] for i in range(n)]

    def find_text_lines_batch_raw(self, words_per_page):
        return self.member(0)[0].find_text_lines_batch_raw(words_per_page)   # host work: any engine handle serves

    def recognize_text_batch_raw(self, inputs, rects, line_offsets, page_line_offsets):
        n = len(inputs)
        pages = (C.c_void_p * n)(*[i._h for i in inputs])
        rects = np.ascontiguousarray(rects, np.float32)
        lo = np.ascontiguousarray(line_offsets, np.uintp)
        po = np.ascontiguousarray(page_line_offsets, np.uintp)
        nl = len(lo) - 1
        chars = C.POINTER(_lib.TextCharC)()
        coffs = C.POINTER(C.c_size_t)()
        check(lib().ocrs_group_recognize_text_batch(
            self._h, pages, C.c_size_t(n), po.ctypes.data_as(C.POINTER(C.c_size_t)),
            rects.ctypes.data_as(C.POINTER(C.c_float)), lo.ctypes.data_as(C.POINTER(C.c_size_t)), C.c_size_t(nl),
            C.byref(chars), C.byref(coffs)))
        return _chars_from_c(chars, coffs, nl)

    def gather(self, payloads):
        """payloads: one bytes object per member -> (concatenation through the group's transport, offsets)."""
        g = len(self)
        assert len(payloads) == g
        bufs = [C.create_string_buffer(bytes(p), max(len(p), 1)) for p in payloads]
        ptrs = (C.c_void_p * g)(*[C.addressof(b) for b in bufs])
        sizes = (C.c_size_t * g)(*[len(p) for p in payloads])
        out = C.c_void_p()
        offs = (C.c_size_t * (g + 1))()
        check(lib().ocrs_group_gather(self._h, ptrs, sizes, C.byref(out), offs))
        data = C.string_at(out, offs[g])
        lib().ocrs_buffer_free(out)
        return data, [int(offs[i]) for i in range(g + 1)]

    def final_gather(self, payloads, mode="auto"):
        """The end-of-stream result gather: like gather(), the transport named per call (auto = RCCL when it can be had)."""
        g = len(self)
        assert len(payloads) == g
        bufs = [C.create_string_buffer(bytes(p), max(len(p), 1)) for p in payloads]
        ptrs = (C.c_void_p * g)(*[C.addressof(b) for b in bufs])
        sizes = (C.c_size_t * g)(*[len(p) for p in payloads])
        out = C.c_void_p()
        offs = (C.c_size_t * (g + 1))()
        check(lib().ocrs_group_final_gather(self._h, C.c_int(self.GATHER[mode]), ptrs, sizes, C.byref(out), offs))
        data = C.string_at(out, offs[g])
        lib().ocrs_buffer_free(out)
        return data, [int(offs[i]) for i in range(g + 1)]

    def set_replay(self, mode, seconds=(0.0, 0.0, 0.0)):
        """ocrs_group_set_replay (test hook): 0 off, 1 record per-page results, 2 replay them after sleeping seconds[stage]."""
        arr = (C.c_double * 3)(*[float(x) for x in seconds])
        check(lib().ocrs_group_set_replay(self._h, int(mode), arr))

    def worker_threads(self):
        n = C.c_size_t(0)
        check(lib().ocrs_group_worker_threads(self._h, C.byref(n)))
        return n.value

    def __del__(self):
        try:
            if self._h:
                lib().ocrs_engine_group_free(self._h)
                self._h = None
        except Exception:
            pass
