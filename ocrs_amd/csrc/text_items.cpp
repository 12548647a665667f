// TextItem::rotated_rect (ocrs/src/text_items.rs:18-30): minimum-area rectangle of the
// characters' box corners, oriented "up".  Host side; the same min-area-rectangle
// restatement as the device code in kernels_ccl.hip (monotone-chain hull, exhaustive
// search over hull edges, strict '<' keeps the first minimal edge), plus
// RotatedRect::orient_towards.  Pinned by ocrs-cli/test-data/format-json-expected.json.
#include <algorithm>
#include <cmath>
#include <vector>

#include "geometry.hpp"

namespace ocrs {
using namespace geom;

namespace {
inline float cross3(PointF o, PointF a, PointF b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

std::vector<PointF> convex_hull(std::vector<PointF> s) {
    std::sort(s.begin(), s.end(), [](const PointF& a, const PointF& b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });
    s.erase(std::unique(s.begin(), s.end(), [](const PointF& a, const PointF& b) { return a.x == b.x && a.y == b.y; }), s.end());
    const int n = (int)s.size();
    if (n <= 2) return s;
    std::vector<PointF> hull(2 * n + 2);
    int k = 0;
    for (int i = 0; i < n; i++) {
        while (k >= 2 && cross3(hull[k - 2], hull[k - 1], s[i]) <= 0.0f) k--;
        hull[k++] = s[i];
    }
    const int lower = k + 1;
    for (int i = n - 2; i >= 0; i--) {
        while (k >= lower && cross3(hull[k - 2], hull[k - 1], s[i]) <= 0.0f) k--;
        hull[k++] = s[i];
    }
    hull.resize(k - 1);
    return hull;
}
}  // namespace

bool min_area_rect(const std::vector<PointF>& pts, RotatedRect* out) {
    const std::vector<PointF> hull = convex_hull(pts);
    const int hn = (int)hull.size();
    bool found = false;
    float best = 3.40282347e+38f;
    for (int e = 0; e < hn; e++) {
        const PointF a = hull[e], b = hull[(e + 1) % hn];
        const float ex = b.x - a.x, ey = b.y - a.y;
        const float len = std::sqrt(ex * ex + ey * ey);
        const float parx = ex / len, pary = ey / len;
        const float perx = -pary, pery = parx;
        float min_par = 3.40282347e+38f, max_par = -3.40282347e+38f, max_perp = -3.40282347e+38f;
        for (const PointF& c : hull) {
            const float dx = c.x - a.x, dy = c.y - a.y;
            const float pp = parx * dx + pary * dy, qq = perx * dx + pery * dy;
            min_par = pp < min_par ? pp : min_par;
            max_par = pp > max_par ? pp : max_par;
            max_perp = qq > max_perp ? qq : max_perp;
        }
        const float height = max_perp, width = max_par - min_par;
        const float area = height * width;
        if (area < best) {
            best = area;
            const float along = min_par + width / 2.0f, half_h = height / 2.0f;
            const float ul = std::sqrt(perx * perx + pery * pery);
            *out = RotatedRect{a.x + along * parx + half_h * perx, a.y + along * pary + half_h * pery, perx / ul, pery / ul,
                               width, height};
            found = true;
        }
    }
    return found;
}

// RotatedRect::orient_towards: of the four (up, width, height) descriptions of the same
// rectangle, the one whose up axis is closest to `target` (last maximum on ties).
RotatedRect orient_towards(const RotatedRect& r, float tx, float ty) {
    const float tl = std::sqrt(tx * tx + ty * ty);
    tx = tx / tl; ty = ty / tl;
    const float rx = r.upy, ry = -r.upx;  // up rotated by 90 degrees
    const float ups[4][2] = {{r.upx, r.upy}, {rx, ry}, {-r.upx, -r.upy}, {-rx, -ry}};
    const float ws[4] = {r.w, r.h, r.w, r.h}, hs[4] = {r.h, r.w, r.h, r.w};
    int best = 0;
    float bd = ups[0][0] * tx + ups[0][1] * ty;
    for (int i = 1; i < 4; i++) {
        const float d = ups[i][0] * tx + ups[i][1] * ty;
        if (d >= bd) { bd = d; best = i; }
    }
    const float ux = ups[best][0], uy = ups[best][1];
    const float ul = std::sqrt(ux * ux + uy * uy);
    return RotatedRect{r.cx, r.cy, ux / ul, uy / ul, ws[best], hs[best]};
}

// text_items.rs:18-30
bool text_item_rotated_rect(const int32_t* tlbr, size_t n_chars, RotatedRect* out) {
    std::vector<PointF> pts;
    pts.reserve(n_chars * 4);
    for (size_t i = 0; i < n_chars; i++) {
        const float t = (float)tlbr[4 * i], l = (float)tlbr[4 * i + 1], b = (float)tlbr[4 * i + 2], r = (float)tlbr[4 * i + 3];
        pts.push_back(PointF{l, t}); pts.push_back(PointF{r, t}); pts.push_back(PointF{r, b}); pts.push_back(PointF{l, b});
    }
    RotatedRect rr;
    if (!min_area_rect(pts, &rr)) return false;
    *out = orient_towards(rr, 0.0f, -1.0f);
    return true;
}

}  // namespace ocrs
