"""Oracle restatement of the rten-imageproc 0.24.0 geometry types that
ocrs/src/{geom_util,layout_analysis,recognition,text_items}.rs call.

TEST INFRASTRUCTURE ONLY.  rten-imageproc is not vendored in /root/reference
(Cargo.lock:735-786), so the method semantics below are restated from the
crate's documented behaviour and from how the reference's own tests use them
(text_items.rs:156-166 pins `corners()` order; lib.rs:437-445 pins
`bounding_rect()`); anything not pinned there is "parity unpinned".

All float arithmetic is IEEE fp32, one rounding per operation (numpy float32
scalars), matching Rust `f32` without FMA contraction.
"""
import math

import numpy as np

f32 = np.float32
_ZERO = f32(0.0)
_TWO = f32(2.0)


def rround(x):
    """Rust f32::round — half away from zero."""
    x = float(x)
    return math.copysign(math.floor(abs(x) + 0.5), x)


def as_i32(x):
    """Rust `as i32` from f32: truncate toward zero, saturate, NaN -> 0."""
    x = float(x)
    if x != x:
        return 0
    if x >= 2147483647.0:
        return 2147483647
    if x <= -2147483648.0:
        return -2147483648
    return int(x)


class Line:
    """Line<f32> / Line<i32>; start/end are (x, y) tuples."""

    __slots__ = ("start", "end")

    def __init__(self, start, end):
        self.start = start
        self.end = end

    def center(self):
        if isinstance(self.start[0], (int, np.integer)):
            return ((self.start[0] + self.end[0]) // 2, (self.start[1] + self.end[1]) // 2)
        return ((self.start[0] + self.end[0]) / _TWO, (self.start[1] + self.end[1]) / _TWO)

    def downwards(self):
        return self if self.start[1] <= self.end[1] else Line(self.end, self.start)

    def rightwards(self):
        return self if self.start[0] <= self.end[0] else Line(self.end, self.start)

    def to_f32(self):
        return Line((f32(self.start[0]), f32(self.start[1])), (f32(self.end[0]), f32(self.end[1])))

    def vertical_overlap(self, other):
        a, b = self.downwards(), other.downwards()
        return _overlap(a.start[1], a.end[1], b.start[1], b.end[1])

    def horizontal_overlap(self, other):
        a, b = self.rightwards(), other.rightwards()
        return _overlap(a.start[0], a.end[0], b.start[0], b.end[0])

    def intersects(self, other):
        """Segment/segment intersection, both parameters in [0,1]; parallel or
        coincident segments do not intersect."""
        x1, y1 = self.start
        x2, y2 = self.end
        x3, y3 = other.start
        x4, y4 = other.end
        a = x2 - x1
        b = -(x4 - x3)
        c = y2 - y1
        d = -(y4 - y3)
        b0 = x3 - x1
        b1 = y3 - y1
        det = a * d - b * c
        if det == 0:
            return False
        s = (d * b0 - b * b1) / det
        t = (a * b1 - c * b0) / det
        return 0 <= s <= 1 and 0 <= t <= 1

    def y_for_x(self, x):
        lo, hi = (self.start[0], self.end[0]) if self.start[0] <= self.end[0] else (self.end[0], self.start[0])
        if x < lo or x > hi:
            return None
        dx = self.end[0] - self.start[0]
        if dx == 0:
            return None
        slope = (self.end[1] - self.start[1]) / dx
        intercept = self.start[1] - slope * self.start[0]
        return slope * x + intercept

    def bounding_rect_i(self):
        return Rect(min(self.start[1], self.end[1]), min(self.start[0], self.end[0]),
                    max(self.start[1], self.end[1]), max(self.start[0], self.end[0]))


def _overlap(a, b, c, d):
    """Length of the overlap of [a,b] and [c,d], >= 0."""
    lo = a if a >= c else c
    hi = b if b <= d else d
    v = hi - lo
    return v if v > 0 else type(v)(0)


class Rect:
    """Rect<i32> (top, left, bottom, right); bottom/right exclusive for area."""

    __slots__ = ("top", "left", "bottom", "right")

    def __init__(self, top, left, bottom, right):
        self.top, self.left, self.bottom, self.right = int(top), int(left), int(bottom), int(right)

    @staticmethod
    def from_tlbr(t, l, b, r):
        return Rect(t, l, b, r)

    @staticmethod
    def from_tlhw(t, l, h, w):
        return Rect(t, l, t + h, l + w)

    @staticmethod
    def from_hw(h, w):
        return Rect(0, 0, h, w)

    def width(self):
        return self.right - self.left

    def height(self):
        return self.bottom - self.top

    def area(self):
        return self.width() * self.height()

    def is_empty(self):
        return self.right <= self.left or self.bottom <= self.top

    def center(self):
        return ((self.left + self.right) // 2, (self.top + self.bottom) // 2)  # (x, y)

    def tlbr(self):
        return (self.top, self.left, self.bottom, self.right)

    def adjust_tlbr(self, t, l, b, r):
        return Rect(self.top + t, self.left + l, self.bottom + b, self.right + r)

    def contains_point(self, p):
        x, y = p
        return self.top <= y <= self.bottom and self.left <= x <= self.right

    def contains(self, o):
        return self.top <= o.top and self.left <= o.left and self.bottom >= o.bottom and self.right >= o.right

    def intersects(self, o):
        return self.left < o.right and self.right > o.left and self.top < o.bottom and self.bottom > o.top

    def union(self, o):
        return Rect(min(self.top, o.top), min(self.left, o.left), max(self.bottom, o.bottom), max(self.right, o.right))

    def iou(self, o):
        it, il = max(self.top, o.top), max(self.left, o.left)
        ib, ir = min(self.bottom, o.bottom), min(self.right, o.right)
        inter = max(ib - it, 0) * max(ir - il, 0)
        union = self.area() + o.area() - inter
        return f32(inter) / f32(union)

    def corners(self):
        # top-left, top-right, bottom-right, bottom-left as (x, y)
        return [(self.left, self.top), (self.right, self.top), (self.right, self.bottom), (self.left, self.bottom)]

    def to_f32(self):
        return RectF(f32(self.top), f32(self.left), f32(self.bottom), f32(self.right))

    def __eq__(self, o):
        return isinstance(o, Rect) and self.tlbr() == o.tlbr()

    def __hash__(self):
        return hash(self.tlbr())

    def __repr__(self):
        return "Rect(t=%d,l=%d,b=%d,r=%d)" % self.tlbr()


class RectF:
    __slots__ = ("top", "left", "bottom", "right")

    def __init__(self, top, left, bottom, right):
        self.top, self.left, self.bottom, self.right = f32(top), f32(left), f32(bottom), f32(right)

    def width(self):
        return self.right - self.left

    def height(self):
        return self.bottom - self.top

    def center(self):
        return ((self.left + self.right) / _TWO, (self.top + self.bottom) / _TWO)

    def union(self, o):
        return RectF(min(self.top, o.top), min(self.left, o.left), max(self.bottom, o.bottom), max(self.right, o.right))

    def integral_bounding_rect(self):
        return Rect(math.floor(self.top), math.floor(self.left), math.ceil(self.bottom), math.ceil(self.right))

    def left_edge(self):
        return Line((self.left, self.top), (self.left, self.bottom))

    def right_edge(self):
        return Line((self.right, self.top), (self.right, self.bottom))

    def tlbr(self):
        return (float(self.top), float(self.left), float(self.bottom), float(self.right))

    def tlhw(self):
        return (float(self.top), float(self.left), float(self.height()), float(self.width()))

    def __eq__(self, o):
        return isinstance(o, RectF) and self.tlbr() == o.tlbr()

    def __repr__(self):
        return "RectF(t=%r,l=%r,b=%r,r=%r)" % self.tlbr()


class RotatedRect:
    """center (x,y), unit `up` axis (x,y), width (extent perpendicular to up),
    height (extent along up)."""

    __slots__ = ("cx", "cy", "upx", "upy", "w", "h")

    def __init__(self, cx, cy, upx, upy, w, h):
        """Raw constructor: takes the six stored floats verbatim."""
        self.cx, self.cy, self.upx, self.upy, self.w, self.h = f32(cx), f32(cy), f32(upx), f32(upy), f32(w), f32(h)

    @staticmethod
    def new(center, up, width, height):
        """RotatedRect::new — normalises `up`."""
        ux, uy = f32(up[0]), f32(up[1])
        ln = np.sqrt(ux * ux + uy * uy)
        return RotatedRect(center[0], center[1], ux / ln, uy / ln, width, height)

    @staticmethod
    def from_rect(r):
        rf = r.to_f32() if isinstance(r, Rect) else r
        c = rf.center()
        return RotatedRect.new(c, (f32(0.0), f32(1.0)), rf.width(), rf.height())

    @staticmethod
    def from_array(a):
        return RotatedRect(a[0], a[1], a[2], a[3], a[4], a[5])

    def to_array(self):
        return np.array([self.cx, self.cy, self.upx, self.upy, self.w, self.h], np.float32)

    def center(self):
        return (self.cx, self.cy)

    def width(self):
        return self.w

    def height(self):
        return self.h

    def area(self):
        return self.w * self.h

    def resize(self, w, h):
        self.w, self.h = f32(w), f32(h)

    def corners(self):
        """text_items.rs:156-166 pins the order for up=(y:-1,x:0)."""
        half_w = self.w / _TWO
        half_h = self.h / _TWO
        # perpendicular(v) = (x: v.y, y: -v.x)
        parx, pary = self.upy * half_w, (-self.upx) * half_w
        perx, pery = self.upx * half_h, self.upy * half_h
        cx, cy = self.cx, self.cy
        return [
            (cx - perx - parx, cy - pery - pary),
            (cx - perx + parx, cy - pery + pary),
            (cx + perx + parx, cy + pery + pary),
            (cx + perx - parx, cy + pery - pary),
        ]

    def bounding_rect(self):
        cs = self.corners()
        xs = [c[0] for c in cs]
        ys = [c[1] for c in cs]
        return RectF(min(ys), min(xs), max(ys), max(xs))

    def __repr__(self):
        return "RotatedRect(c=(%g,%g) up=(%g,%g) w=%g h=%g)" % (self.cx, self.cy, self.upx, self.upy, self.w, self.h)


def bounding_rect_of(rects_f):
    """rten_imageproc::bounding_rect(iter) -> union of RectF or None."""
    out = None
    for r in rects_f:
        out = r if out is None else out.union(r)
    return out


# ---- ocrs/src/geom_util.rs:6-26 -------------------------------------------
def _sorted_corners_by_x(r):
    cs = r.corners()
    # stable sort by x (f32 total order; no NaNs here)
    return sorted(cs, key=lambda c: float(c[0]))


def rightmost_edge(r):
    cs = _sorted_corners_by_x(r)
    return Line(cs[2], cs[3])


def leftmost_edge(r):
    cs = _sorted_corners_by_x(r)
    return Line(cs[0], cs[1])


def downwards_line(l):
    return l if l.start[1] <= l.end[1] else Line(l.end, l.start)
