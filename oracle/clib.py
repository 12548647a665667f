"""ctypes loader for the C oracle (oracle/csrc/ocrs_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never from ocrs_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libocrs_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "csrc", "ocrs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_find_contours_external.restype = C.c_int
        _lib.orc_component_rects.restype = C.c_int
        _lib.orc_simplify_polygon.restype = C.c_int
        _lib.orc_min_area_rect.restype = C.c_int
        _lib.orc_convex_hull.restype = C.c_int
        _lib.orc_ctc_greedy.restype = C.c_int
        _lib.orc_prepare_image.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- image ops
def prepare_image(pixels, chans_last):
    """preprocess.rs:149-248.  pixels: u8 or f32 [H,W,C] (chans_last) or [C,H,W]."""
    a = np.ascontiguousarray(pixels)
    if a.dtype not in (np.uint8, np.float32):
        raise TypeError("pixels must be uint8 or float32")
    if chans_last:
        h, w, c = a.shape
    else:
        c, h, w = a.shape
    out = np.empty((1, h, w), np.float32)
    rc = lib().orc_prepare_image(_p(a), C.c_int(a.dtype == np.uint8), C.c_int(bool(chans_last)),
                                 C.c_int(h), C.c_int(w), C.c_int(c), _p(out))
    if rc != 0:
        raise ValueError("expected greyscale, RGB or RGBA input image")
    return out


def resize_bilinear(src, out_h, out_w, virt_h=None, virt_w=None, fill=-0.5):
    src = _f32(src)
    sh, sw = src.shape
    vh = sh if virt_h is None else virt_h
    vw = sw if virt_w is None else virt_w
    dst = np.empty((out_h, out_w), np.float32)
    lib().orc_resize_bilinear(_p(src), C.c_int(sh), C.c_int(sw), C.c_int(sw), C.c_int(vh), C.c_int(vw),
                              C.c_float(fill), _p(dst), C.c_int(out_h), C.c_int(out_w))
    return dst


def threshold(prob, thr):
    prob = _f32(prob)
    mask = np.empty(prob.shape, np.uint8)
    lib().orc_threshold(_p(prob), C.c_float(thr), _p(mask), C.c_int64(prob.size))
    return mask


# ---------------------------------------------------------------- contours / rects
def find_contours_external(mask):
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    h, w = mask.shape
    cap = h * w * 4 + 16
    pts = np.empty((cap, 2), np.int32)
    maxc = h * w // 2 + 16
    offs = np.empty(maxc + 1, np.int64)
    n = lib().orc_find_contours_external(_p(mask), C.c_int(h), C.c_int(w), _p(pts), C.c_int64(cap),
                                         _p(offs), C.c_int(maxc))
    if n < 0:
        raise RuntimeError("contour capacity exceeded")
    return [pts[offs[i]:offs[i + 1]].copy() for i in range(n)]  # each [k,2] (y,x)


def simplify_polygon(xy, eps):
    xy = _f32(xy).reshape(-1, 2)
    out = np.empty_like(xy)
    m = lib().orc_simplify_polygon(_p(xy), C.c_int(len(xy)), C.c_float(eps), _p(out))
    return out[:m].copy()


def convex_hull(xy):
    xy = _f32(xy).reshape(-1, 2)
    out = np.empty((2 * len(xy) + 2, 2), np.float32)
    m = lib().orc_convex_hull(_p(xy), C.c_int(len(xy)), _p(out))
    return out[:m].copy()


def min_area_rect(xy):
    """Returns 6 floats (cx, cy, upx, upy, w, h) or None."""
    xy = _f32(xy).reshape(-1, 2)
    rr = np.empty(6, np.float32)
    ok = lib().orc_min_area_rect(_p(xy), C.c_int(len(xy)), _p(rr))
    return rr if ok else None


def component_rects(mask, expand, min_area):
    """detection.rs:41-62 -> [n,6] float32 (cx, cy, upx, upy, w, h)."""
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    h, w = mask.shape
    maxr = h * w // 2 + 16
    rects = np.empty((maxr, 6), np.float32)
    n = lib().orc_component_rects(_p(mask), C.c_int(h), C.c_int(w), C.c_float(expand),
                                  C.c_float(min_area), _p(rects), C.c_int(maxr))
    if n < 0:
        raise RuntimeError("component capacity exceeded")
    return rects[:n].copy()


# ---------------------------------------------------------------- line crops
def polygon_fill_mask(poly_yx, top, left, bh, bw):
    poly = np.ascontiguousarray(poly_yx, dtype=np.int32).reshape(-1, 2)
    out = np.empty((bh, bw), np.uint8)
    lib().orc_polygon_fill_mask(_p(poly), C.c_int(len(poly)), C.c_int(top), C.c_int(left), C.c_int(bh),
                                C.c_int(bw), _p(out))
    return out


def prepare_text_line_into(page_hw, poly_yx, resized_w, out_h, dst2d):
    """recognition.rs:91-126 + :152-154; dst2d is a writable [out_h, >=resized_w] view
    with contiguous rows."""
    page = _f32(page_hw)
    poly = np.ascontiguousarray(poly_yx, dtype=np.int32).reshape(-1, 2)
    assert dst2d.dtype == np.float32 and dst2d.strides[1] == 4
    lib().orc_prepare_text_line(_p(page), C.c_int(page.shape[0]), C.c_int(page.shape[1]), _p(poly),
                                C.c_int(len(poly)), C.c_int(resized_w), C.c_int(out_h),
                                C.c_void_p(dst2d.ctypes.data), C.c_int(dst2d.strides[0] // 4))


# ---------------------------------------------------------------- nn ops (NHWC)
def conv2d(x, wt, b, relu):
    x = _f32(x); wt = _f32(wt); b = _f32(b)
    n, h, w, cin = x.shape
    kh, kw, cin2, cout = wt.shape
    assert cin == cin2
    y = np.empty((n, h, w, cout), np.float32)
    lib().orc_conv2d(_p(x), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(cin), _p(wt), _p(b), C.c_int(kh),
                     C.c_int(kw), C.c_int(cout), C.c_int(relu), _p(y))
    return y


def dwconv3x3(x, wt, b, relu):
    x = _f32(x); wt = _f32(wt); b = _f32(b)
    n, h, w, c = x.shape
    assert wt.shape == (3, 3, c)
    y = np.empty_like(x)
    lib().orc_dwconv3x3(_p(x), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(c), _p(wt), _p(b), C.c_int(relu), _p(y))
    return y


def maxpool(x, kh, kw):
    x = _f32(x)
    n, h, w, c = x.shape
    y = np.empty((n, h // kh, w // kw, c), np.float32)
    lib().orc_maxpool(_p(x), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(c), C.c_int(kh), C.c_int(kw), _p(y))
    return y


def avgpool(x, kh, kw):
    x = _f32(x)
    n, h, w, c = x.shape
    y = np.empty((n, h // kh, w // kw, c), np.float32)
    lib().orc_avgpool(_p(x), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(c), C.c_int(kh), C.c_int(kw), _p(y))
    return y


def convt2x2(x, wt, b):
    x = _f32(x); wt = _f32(wt); b = _f32(b)
    n, h, w, cin = x.shape
    assert wt.shape[:3] == (2, 2, cin)
    cout = wt.shape[3]
    y = np.empty((n, 2 * h, 2 * w, cout), np.float32)
    lib().orc_convt2x2(_p(x), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(cin), _p(wt), _p(b), C.c_int(cout), _p(y))
    return y


def padcat(skip, x):
    skip = _f32(skip); x = _f32(x)
    n, sh, sw, cs = skip.shape
    _, h, w, cx = x.shape
    y = np.empty((n, sh, sw, cs + cx), np.float32)
    lib().orc_padcat(_p(skip), C.c_int(n), C.c_int(sh), C.c_int(sw), C.c_int(cs), _p(x), C.c_int(h), C.c_int(w),
                     C.c_int(cx), _p(y))
    return y


def sigmoid(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().orc_sigmoid(_p(x), _p(y), C.c_int64(x.size))
    return y


def linear(x, wt, b):
    x = _f32(x); wt = _f32(wt); b = _f32(b)
    k, o = wt.shape
    rows = x.size // k
    y = np.empty(x.shape[:-1] + (o,), np.float32)
    lib().orc_linear(_p(x), C.c_int64(rows), C.c_int(k), _p(wt), _p(b), C.c_int(o), _p(y))
    return y


def gru_bidir(x, wts):
    """x: [T,N,I]; wts: (wi_f, bi_f, wh_f, bh_f, wi_b, bi_b, wh_b, bh_b) -> [T,N,2H]."""
    x = _f32(x)
    t, n, i = x.shape
    wts = [_f32(a) for a in wts]
    hdim = wts[2].shape[0]
    y = np.empty((t, n, 2 * hdim), np.float32)
    for d in range(2):
        wi, bi, wh, bh = wts[4 * d:4 * d + 4]
        lib().orc_gru_dir(_p(x), C.c_int(t), C.c_int(n), C.c_int(i), _p(wi), _p(bi), _p(wh), _p(bh),
                          C.c_int(hdim), C.c_int(d), _p(y), C.c_int(2 * hdim), C.c_int(d * hdim))
    return y


def log_softmax(x):
    x = _f32(x)
    c = x.shape[-1]
    y = np.empty_like(x)
    lib().orc_log_softmax(_p(x), C.c_int64(x.size // c), C.c_int(c), _p(y))
    return y


def spec_exp(x):
    x = _f32(x); y = np.empty_like(x)
    lib().orc_exp(_p(x), _p(y), C.c_int64(x.size)); return y


def spec_log(x):
    x = _f32(x); y = np.empty_like(x)
    lib().orc_log(_p(x), _p(y), C.c_int64(x.size)); return y


def spec_tanh(x):
    x = _f32(x); y = np.empty_like(x)
    lib().orc_tanh(_p(x), _p(y), C.c_int64(x.size)); return y


# ---------------------------------------------------------------- CTC
def ctc_greedy(seq_tc):
    """seq_tc: [T,C] log-probs -> list of (label, pos)."""
    seq = _f32(seq_tc)
    t, c = seq.shape
    labels = np.empty(t, np.uint32)
    pos = np.empty(t, np.uint32)
    n = lib().orc_ctc_greedy(_p(seq), C.c_int(t), C.c_int(c), C.c_int(c), _p(labels), _p(pos))
    return [(int(labels[i]), int(pos[i])) for i in range(n)]
