// Host-side geometry used by layout analysis and recognition pre/post-processing.
// Mirrors the rten-imageproc 0.24.0 types that ocrs/src/{geom_util,
// layout_analysis,recognition}.rs call (Point/Line/Rect/RotatedRect); fp32
// arithmetic with one rounding per operation (this translation unit is built
// with -ffp-contract=off), integer casts with Rust `as` semantics.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <optional>
#include <vector>

namespace ocrs {
namespace geom {

struct PointF { float x, y; };
struct PointI { int32_t x, y; };

// Rust `f32 as i32`: truncate toward zero, saturate, NaN -> 0.
inline int32_t as_i32(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return std::numeric_limits<int32_t>::max();
    if (v <= -2147483648.0f) return std::numeric_limits<int32_t>::min();
    return (int32_t)v;
}
// Rust f32::round: half away from zero.
inline float rround(float v) { return std::round(v); }

template <class T>
inline T overlap(T a, T b, T c, T d) {  // length of [a,b] ∩ [c,d], >= 0
    T lo = a >= c ? a : c;
    T hi = b <= d ? b : d;
    T v = hi - lo;
    return v > T(0) ? v : T(0);
}

struct LineF {
    PointF start, end;
    PointF center() const { return PointF{(start.x + end.x) / 2.0f, (start.y + end.y) / 2.0f}; }
    LineF downwards() const { return start.y <= end.y ? *this : LineF{end, start}; }
    LineF rightwards() const { return start.x <= end.x ? *this : LineF{end, start}; }
    float vertical_overlap(const LineF& o) const {
        LineF a = downwards(), b = o.downwards();
        return overlap(a.start.y, a.end.y, b.start.y, b.end.y);
    }
    float horizontal_overlap(const LineF& o) const {
        LineF a = rightwards(), b = o.rightwards();
        return overlap(a.start.x, a.end.x, b.start.x, b.end.x);
    }
    // Segment intersection (both parameters within [0,1]); parallel -> false.
    bool intersects(const LineF& o) const {
        float a = end.x - start.x, b = -(o.end.x - o.start.x);
        float c = end.y - start.y, d = -(o.end.y - o.start.y);
        float b0 = o.start.x - start.x, b1 = o.start.y - start.y;
        float det = a * d - b * c;
        if (det == 0.0f) return false;
        float s = (d * b0 - b * b1) / det;
        float t = (a * b1 - c * b0) / det;
        return s >= 0.0f && s <= 1.0f && t >= 0.0f && t <= 1.0f;
    }
    std::optional<float> y_for_x(float x) const {
        float lo = start.x <= end.x ? start.x : end.x;
        float hi = start.x <= end.x ? end.x : start.x;
        if (x < lo || x > hi) return std::nullopt;
        float dx = end.x - start.x;
        if (dx == 0.0f) return std::nullopt;
        float slope = (end.y - start.y) / dx;
        float intercept = start.y - slope * start.x;
        return slope * x + intercept;
    }
};

struct Rect {  // Rect<i32>
    int32_t top, left, bottom, right;
    static Rect from_tlbr(int32_t t, int32_t l, int32_t b, int32_t r) { return Rect{t, l, b, r}; }
    int32_t width() const { return right - left; }
    int32_t height() const { return bottom - top; }
    int64_t area() const { return (int64_t)width() * height(); }
    bool is_empty() const { return right <= left || bottom <= top; }
    PointI center() const { return PointI{(left + right) / 2, (top + bottom) / 2}; }
    bool intersects(const Rect& o) const { return left < o.right && right > o.left && top < o.bottom && bottom > o.top; }
    bool contains_point(int32_t x, int32_t y) const { return top <= y && y <= bottom && left <= x && x <= right; }
    Rect unite(const Rect& o) const {
        return Rect{std::min(top, o.top), std::min(left, o.left), std::max(bottom, o.bottom), std::max(right, o.right)};
    }
    float iou(const Rect& o) const {
        int32_t it = std::max(top, o.top), il = std::max(left, o.left);
        int32_t ib = std::min(bottom, o.bottom), ir = std::min(right, o.right);
        int64_t inter = (int64_t)std::max(ib - it, 0) * std::max(ir - il, 0);
        int64_t uni = area() + o.area() - inter;
        return (float)inter / (float)uni;
    }
    bool operator==(const Rect& o) const { return top == o.top && left == o.left && bottom == o.bottom && right == o.right; }
};

struct RectF {
    float top, left, bottom, right;
    float width() const { return right - left; }
    float height() const { return bottom - top; }
    RectF unite(const RectF& o) const {
        return RectF{std::min(top, o.top), std::min(left, o.left), std::max(bottom, o.bottom), std::max(right, o.right)};
    }
    Rect integral_bounding_rect() const {
        return Rect{(int32_t)std::floor(top), (int32_t)std::floor(left), (int32_t)std::ceil(bottom), (int32_t)std::ceil(right)};
    }
    LineF left_edge() const { return LineF{PointF{left, top}, PointF{left, bottom}}; }
    LineF right_edge() const { return LineF{PointF{right, top}, PointF{right, bottom}}; }
};

// RotatedRect crossing the ABI as (cx, cy, upx, upy, w, h).
struct RotatedRect {
    float cx, cy, upx, upy, w, h;
    static RotatedRect from_array(const float* a) { return RotatedRect{a[0], a[1], a[2], a[3], a[4], a[5]}; }
    void to_array(float* a) const { a[0] = cx; a[1] = cy; a[2] = upx; a[3] = upy; a[4] = w; a[5] = h; }
    PointF center() const { return PointF{cx, cy}; }
    // order pinned by text_items.rs:156-166
    std::array<PointF, 4> corners() const {
        float half_w = w / 2.0f, half_h = h / 2.0f;
        float parx = upy * half_w, pary = (-upx) * half_w;  // perpendicular(up) = (up.y, -up.x)
        float perx = upx * half_h, pery = upy * half_h;
        return {PointF{cx - perx - parx, cy - pery - pary}, PointF{cx - perx + parx, cy - pery + pary},
                PointF{cx + perx + parx, cy + pery + pary}, PointF{cx + perx - parx, cy + pery - pary}};
    }
    RectF bounding_rect() const {
        auto c = corners();
        float x0 = c[0].x, x1 = c[0].x, y0 = c[0].y, y1 = c[0].y;
        for (int i = 1; i < 4; i++) {
            x0 = std::min(x0, c[i].x); x1 = std::max(x1, c[i].x);
            y0 = std::min(y0, c[i].y); y1 = std::max(y1, c[i].y);
        }
        return RectF{y0, x0, y1, x1};
    }
};

// geom_util.rs:6-26
inline std::array<PointF, 4> corners_sorted_by_x(const RotatedRect& r) {
    auto c = r.corners();
    std::stable_sort(c.begin(), c.end(), [](const PointF& a, const PointF& b) { return a.x < b.x; });
    return c;
}
inline LineF rightmost_edge(const RotatedRect& r) { auto c = corners_sorted_by_x(r); return LineF{c[2], c[3]}; }
inline LineF leftmost_edge(const RotatedRect& r) { auto c = corners_sorted_by_x(r); return LineF{c[0], c[1]}; }
inline LineF downwards_line(const LineF& l) { return l.start.y <= l.end.y ? l : LineF{l.end, l.start}; }

}  // namespace geom

// text_items.cpp — text_items.rs:18-30
bool text_item_rotated_rect(const int32_t* tlbr, size_t n_chars, geom::RotatedRect* out);

// layout.cpp — layout_analysis.rs:158-233
std::vector<std::vector<geom::RotatedRect>> find_text_lines(const std::vector<geom::RotatedRect>& words);
std::vector<geom::Rect> find_block_separators(const std::vector<geom::RotatedRect>& words);

}  // namespace ocrs
