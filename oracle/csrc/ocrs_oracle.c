/*
 * ocrs_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the arithmetic on the ocrs hot path
 * (prepare_input -> detect_words -> recognize_text), used ONLY as the checker
 * in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing
 * under ocrs_amd/ may include, link or call this file.
 *
 * PARITY STATUS: the reference (robertknight/ocrs 0.12.2) cannot be built in
 * this environment (no Rust toolchain) and the arithmetic it calls lives in
 * the un-vendored crates rten / rten-imageproc / rten-tensor 0.24.0
 * (Cargo.lock:682-786).  This file therefore restates
 *   (1) the in-tree Rust of ocrs/src/{preprocess,detection,recognition}.rs, and
 *   (2) the published algorithms behind the rten calls made there (ONNX
 *       Resize/Conv/GRU/... operator semantics; Suzuki-Abe border following;
 *       Ramer-Douglas-Peucker; exhaustive-search minimum-area rectangle; CTC
 *       greedy decoding),
 * and is pinned against every known-answer test the reference holds for the
 * path (tests/test_oracle_kat.py: preprocess.rs:379-594, detection.rs:213-246,
 * lib.rs:466-488, lib.rs:527-577).  Everything else is "parity unpinned":
 * see DESIGN.md §3.
 *
 * Floating point: fp32 throughout, like the reference.  Built with
 * -ffp-contract=off; every fused multiply-add is an explicit fmaf().  The
 * NUMERIC SPEC (accumulation order, exp/log/sigmoid/tanh polynomials) is
 * written down in DESIGN.md §4; this file and the HIP kernels each restate it
 * independently, and because gfx950's fp32 MFMA is bit-for-bit a k-ordered
 * fmaf chain the two agree bitwise.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* Numeric spec: transcendental functions             */
/* ------------------------------------------------------------------ */

typedef union { float f; int32_t i; uint32_t u; } f32bits;

static inline float spec_expf(float x) {
    if (x != x) return x;
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    float k = rintf(x * 1.44269504088896341f);
    float r = fmaf(k, -0.693145751953125f, x);
    r = fmaf(k, -1.42860682030941723212e-6f, r);
    float p = 1.98412698412698413e-4f;            /* 1/5040 */
    p = fmaf(p, r, 1.38888888888888894e-3f);      /* 1/720  */
    p = fmaf(p, r, 8.33333333333333322e-3f);      /* 1/120  */
    p = fmaf(p, r, 4.16666666666666644e-2f);      /* 1/24   */
    p = fmaf(p, r, 1.66666666666666657e-1f);      /* 1/6    */
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    f32bits u; u.f = p;
    u.i += ((int32_t)k) << 23;
    return u.f;
}

static inline float spec_logf(float s) {
    f32bits u; u.f = s;
    int e = (int)((u.u >> 23) & 0xffu) - 127;
    u.u = (u.u & 0x007fffffu) | 0x3f800000u;
    float m = u.f;
    if (m > 1.41421356237309515f) { m = m * 0.5f; e += 1; }
    float t = (m - 1.0f) / (m + 1.0f);
    float t2 = t * t;
    float p = 1.11111111111111105e-1f;            /* 1/9 */
    p = fmaf(p, t2, 1.42857142857142849e-1f);     /* 1/7 */
    p = fmaf(p, t2, 0.2f);
    p = fmaf(p, t2, 3.33333333333333315e-1f);     /* 1/3 */
    p = fmaf(p, t2, 1.0f);
    float lm = (2.0f * t) * p;
    return fmaf((float)e, 0.693147180559945286f, lm);
}

static inline float spec_sigmoidf(float x) { return 1.0f / (1.0f + spec_expf(-x)); }

static inline float spec_tanhf(float x) {
    float t = spec_expf(2.0f * x);
    return (t - 1.0f) / (t + 1.0f);
}

ORC_API void orc_exp(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = spec_expf(x[i]); }
ORC_API void orc_log(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = spec_logf(x[i]); }
ORC_API void orc_sigmoid(const float* x, float* y, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) y[i] = spec_sigmoidf(x[i]);
}
ORC_API void orc_tanh(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = spec_tanhf(x[i]); }

/* ------------------------------------------------------------------ */
/* Stage 0: prepare_image (ocrs/src/preprocess.rs:149-248)             */
/* ------------------------------------------------------------------ */

/* out[y,x] = -0.5 + sum_{c<min(chans,3)} px[c] * w[c], accumulated in channel
 * order starting from BLACK_VALUE (preprocess.rs:229-233 / :238-241).
 * is_u8: weights are ITU/255 (preprocess.rs:182), grey weight 1/255 (:184).
 * chans_last: HWC (1) or CHW (0). */
ORC_API int orc_prepare_image(const void* src, int is_u8, int chans_last, int h, int w, int chans,
                              float* out) {
    if (!(chans == 1 || chans == 3 || chans == 4)) return -1;
    const float itu[3] = {0.299f, 0.587f, 0.114f};
    float wts[3];
    int nw = chans == 1 ? 1 : 3;
    for (int c = 0; c < 3; c++) {
        if (chans == 1) wts[c] = is_u8 ? (1.0f / 255.0f) : 1.0f;
        else wts[c] = is_u8 ? (itu[c] / 255.0f) : itu[c];
    }
    const uint8_t* s8 = (const uint8_t*)src;
    const float* sf = (const float*)src;
    const int64_t plane = (int64_t)h * w;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < plane; p++) {
        float px = -0.5f;
        for (int c = 0; c < nw; c++) {
            int64_t idx = chans_last ? p * chans + c : (int64_t)c * plane + p;
            float v = is_u8 ? (float)s8[idx] : sf[idx];
            px += v * wts[c];
        }
        out[p] = px;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Bilinear resize — rten `resize_image` (detection.rs:168,194;        */
/* recognition.rs:121).  ONNX Resize, mode=linear,                     */
/* coordinate_transformation_mode=half_pixel, no antialias.            */
/* [UNVERIFIED-RECALL of rten internals; pinned by lib.rs:437-488]     */
/* ------------------------------------------------------------------ */

static inline void resize_axis(int o, int in_len, int out_len, int* i0, int* i1, float* wgt) {
    float scale = (float)in_len / (float)out_len;
    float c = ((float)o + 0.5f) * scale - 0.5f;
    float hi = (float)(in_len - 1);
    if (c < 0.0f) c = 0.0f;
    if (c > hi) c = hi;
    int a = (int)c;
    int b = a + 1 < in_len ? a + 1 : in_len - 1;
    *i0 = a; *i1 = b; *wgt = c - (float)a;
}

/* Source is a virtual [vh, vw] image: pixels with y < sh && x < sw come from
 * `src` (row stride `sstride`), everything else reads `fill` (this is the
 * constant pad of detection.rs:155-164 folded into the resize). */
ORC_API void orc_resize_bilinear(const float* src, int sh, int sw, int sstride, int vh, int vw,
                                 float fill, float* dst, int dh, int dw) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y++) {
        int y0, y1; float wy;
        resize_axis(y, vh, dh, &y0, &y1, &wy);
        for (int x = 0; x < dw; x++) {
            int x0, x1; float wx;
            resize_axis(x, vw, dw, &x0, &x1, &wx);
            float tl = (y0 < sh && x0 < sw) ? src[(int64_t)y0 * sstride + x0] : fill;
            float tr = (y0 < sh && x1 < sw) ? src[(int64_t)y0 * sstride + x1] : fill;
            float bl = (y1 < sh && x0 < sw) ? src[(int64_t)y1 * sstride + x0] : fill;
            float br = (y1 < sh && x1 < sw) ? src[(int64_t)y1 * sstride + x1] : fill;
            float top = (1.0f - wx) * tl + wx * tr;
            float bot = (1.0f - wx) * bl + wx * br;
            dst[(int64_t)y * dw + x] = (1.0f - wy) * top + wy * bot;
        }
    }
}

/* detection.rs:110 — strict `>` */
ORC_API void orc_threshold(const float* p, float thr, uint8_t* mask, int64_t n) {
    for (int64_t i = 0; i < n; i++) mask[i] = p[i] > thr ? 1 : 0;
}

/* ------------------------------------------------------------------ */
/* find_contours(mask, RetrievalMode::External) — rten-imageproc,      */
/* called at detection.rs:46.  Suzuki & Abe 1985, Appendix I border    */
/* following with the Algorithm 2 (outermost borders only) changes.    */
/* ------------------------------------------------------------------ */

/* Neighbour offsets in CLOCKWISE order as seen on screen (y down):
 * W, NW, N, NE, E, SE, S, SW. */
static const int NB_DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};
static const int NB_DX[8] = {-1, -1, 0, 1, 1, 1, 0, -1};

static inline int nb_index(int dy, int dx) {
    for (int i = 0; i < 8; i++) if (NB_DY[i] == dy && NB_DX[i] == dx) return i;
    return -1;
}

/* Returns number of contours.  `points` receives (y,x) int32 pairs for all
 * contours back-to-back; `offsets[i]..offsets[i+1]` delimit contour i.
 * Returns -1 if a capacity is exceeded. */
ORC_API int orc_find_contours_external(const uint8_t* mask, int h, int w, int32_t* points,
                                       int64_t points_cap, int64_t* offsets, int max_contours) {
    const int ph = h + 2, pw = w + 2;
    int8_t* f = (int8_t*)calloc((size_t)ph * pw, 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) f[(y + 1) * pw + x + 1] = mask[(int64_t)y * w + x] ? 1 : 0;
    int n_contours = 0;
    int64_t n_points = 0;
    offsets[0] = 0;
    for (int i = 1; i <= h; i++) {
        int lnbd = 0;
        for (int j = 1; j <= w; j++) {
            int8_t fij = f[i * pw + j];
            if (fij == 0) continue;
            if (fij == 1 && f[i * pw + j - 1] == 0 && lnbd <= 0) {
                /* outer border start, (i2,j2) = (i, j-1) */
                if (n_contours >= max_contours) { free(f); return -1; }
                int start_dir = 0; /* W */
                int first = -1;
                /* 3.1: clockwise from (i2,j2), first non-zero neighbour */
                for (int s = 0; s < 8; s++) {
                    int d = (start_dir + s) & 7;
                    if (f[(i + NB_DY[d]) * pw + j + NB_DX[d]] != 0) { first = d; break; }
                }
                if (first < 0) {
                    f[i * pw + j] = -2;
                    if (n_points + 1 > points_cap) { free(f); return -1; }
                    points[2 * n_points] = i - 1; points[2 * n_points + 1] = j - 1; n_points++;
                } else {
                    const int i1 = i + NB_DY[first], j1 = j + NB_DX[first];
                    int i2 = i1, j2 = j1, i3 = i, j3 = j;
                    for (;;) {
                        /* 3.3: counter-clockwise from the element after (i2,j2) */
                        int d0 = nb_index(i2 - i3, j2 - j3);
                        int east_zero_examined = 0;
                        int i4 = 0, j4 = 0;
                        for (int s = 1; s <= 8; s++) {
                            int d = (d0 - s) & 7; /* counter-clockwise = decreasing index */
                            int yy = i3 + NB_DY[d], xx = j3 + NB_DX[d];
                            if (f[yy * pw + xx] != 0) { i4 = yy; j4 = xx; break; }
                            if (d == 4) east_zero_examined = 1;
                        }
                        /* 3.4 */
                        if (east_zero_examined) f[i3 * pw + j3] = -2;
                        else if (f[i3 * pw + j3] == 1) f[i3 * pw + j3] = 2;
                        if (n_points + 1 > points_cap) { free(f); return -1; }
                        points[2 * n_points] = i3 - 1; points[2 * n_points + 1] = j3 - 1; n_points++;
                        /* 3.5 */
                        if (i4 == i && j4 == j && i3 == i1 && j3 == j1) break;
                        i2 = i3; j2 = j3; i3 = i4; j3 = j4;
                    }
                }
                n_contours++;
                offsets[n_contours] = n_points;
            }
            /* step 4 (Algorithm 2: signed) */
            fij = f[i * pw + j];
            if (fij != 1) lnbd = fij;
        }
    }
    free(f);
    return n_contours;
}

/* ------------------------------------------------------------------ */
/* simplify_polygon (RDP, eps) and min_area_rect — rten-imageproc,     */
/* called at detection.rs:50,52.                                       */
/* ------------------------------------------------------------------ */

typedef struct { float x, y; } ptf;

/* Distance from p to the segment a-b (clamped projection); a==b -> |p-a|. */
static inline float seg_distance(ptf a, ptf b, ptf p) {
    float abx = b.x - a.x, aby = b.y - a.y;
    float apx = p.x - a.x, apy = p.y - a.y;
    float len2 = abx * abx + aby * aby;
    if (len2 == 0.0f) return sqrtf(apx * apx + apy * apy);
    float t = (apx * abx + apy * aby) / len2;
    if (t < 0.0f) t = 0.0f;
    if (t > 1.0f) t = 1.0f;
    float qx = a.x + t * abx, qy = a.y + t * aby;
    float dx = p.x - qx, dy = p.y - qy;
    return sqrtf(dx * dx + dy * dy);
}

/* Iterative RDP over polyline pts[0..n-1]; keep[] marks survivors.
 * Pivot = FIRST point attaining the maximum distance; split iff max > eps. */
static void rdp_mark(const ptf* pts, int n, float eps, uint8_t* keep, int* stack) {
    memset(keep, 0, (size_t)n);
    keep[0] = 1; keep[n - 1] = 1;
    int sp = 0;
    stack[sp++] = 0; stack[sp++] = n - 1;
    while (sp > 0) {
        int hi = stack[--sp], lo = stack[--sp];
        if (hi - lo < 2) continue;
        float maxd = 0.0f; int maxi = -1;
        for (int k = lo + 1; k < hi; k++) {
            float d = seg_distance(pts[lo], pts[hi], pts[k]);
            if (d > maxd) { maxd = d; maxi = k; }
        }
        if (maxi >= 0 && maxd > eps) {
            keep[maxi] = 1;
            stack[sp++] = lo; stack[sp++] = maxi;
            stack[sp++] = maxi; stack[sp++] = hi;
        }
    }
}

/* Closed polygon -> close it into a polyline, simplify, drop the duplicated
 * end point.  Returns number of output points. */
static int simplify_polygon_impl(const ptf* poly, int n, float eps, ptf* out) {
    if (n == 0) return 0;
    ptf* line = (ptf*)malloc(sizeof(ptf) * (size_t)(n + 1));
    memcpy(line, poly, sizeof(ptf) * (size_t)n);
    line[n] = poly[0];
    uint8_t* keep = (uint8_t*)malloc((size_t)n + 1);
    int* stack = (int*)malloc(sizeof(int) * (size_t)(2 * (n + 2)));
    rdp_mark(line, n + 1, eps, keep, stack);
    int m = 0;
    for (int k = 0; k < n; k++) if (keep[k]) out[m++] = line[k]; /* index n (dup) dropped */
    free(line); free(keep); free(stack);
    return m;
}

ORC_API int orc_simplify_polygon(const float* xy, int n, float eps, float* out_xy) {
    return simplify_polygon_impl((const ptf*)xy, n, eps, (ptf*)out_xy);
}

static int cmp_ptf(const void* a, const void* b) {
    const ptf* p = (const ptf*)a; const ptf* q = (const ptf*)b;
    if (p->x < q->x) return -1;
    if (p->x > q->x) return 1;
    if (p->y < q->y) return -1;
    if (p->y > q->y) return 1;
    return 0;
}

static inline float cross3(ptf o, ptf a, ptf b) {
    return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x);
}

/* Andrew monotone chain.  Output order: start at (min x, min y), walk the
 * chain with minimal y first — clockwise as seen on screen (y down).
 * Collinear points are dropped.  Returns hull size (1 or 2 for degenerate). */
static int convex_hull_impl(const ptf* pts, int n, ptf* hull) {
    if (n == 0) return 0;
    ptf* s = (ptf*)malloc(sizeof(ptf) * (size_t)n);
    memcpy(s, pts, sizeof(ptf) * (size_t)n);
    qsort(s, (size_t)n, sizeof(ptf), cmp_ptf);
    int m = 0;
    for (int i = 0; i < n; i++) /* dedupe */
        if (m == 0 || s[i].x != s[m - 1].x || s[i].y != s[m - 1].y) s[m++] = s[i];
    n = m;
    if (n <= 2) { memcpy(hull, s, sizeof(ptf) * (size_t)n); free(s); return n; }
    int k = 0;
    for (int i = 0; i < n; i++) {
        while (k >= 2 && cross3(hull[k - 2], hull[k - 1], s[i]) <= 0.0f) k--;
        hull[k++] = s[i];
    }
    int lower = k + 1;
    for (int i = n - 2; i >= 0; i--) {
        while (k >= lower && cross3(hull[k - 2], hull[k - 1], s[i]) <= 0.0f) k--;
        hull[k++] = s[i];
    }
    free(s);
    return k - 1;
}

/* RotatedRect as 6 floats: cx, cy, upx, upy, w, h. */
static int min_area_rect_impl(const ptf* pts, int n, float* rr) {
    ptf* hull = (ptf*)malloc(sizeof(ptf) * (size_t)(2 * n + 2));
    int hn = convex_hull_impl(pts, n, hull);
    int found = 0;
    float best_area = 3.40282347e+38f;
    for (int e = 0; e < hn; e++) {
        ptf a = hull[e], b = hull[(e + 1) % hn];
        float ex = b.x - a.x, ey = b.y - a.y;
        float len = sqrtf(ex * ex + ey * ey);
        float parx = ex / len, pary = ey / len;
        /* perpendicular(v) = (v.y, -v.x); inward axis = -perpendicular */
        float perx = -pary, pery = parx;
        float min_par = 3.40282347e+38f, max_par = -3.40282347e+38f, max_perp = -3.40282347e+38f;
        for (int k = 0; k < hn; k++) {
            float dx = hull[k].x - a.x, dy = hull[k].y - a.y;
            float pp = parx * dx + pary * dy;
            float qq = perx * dx + pery * dy;
            if (pp < min_par) min_par = pp;
            if (pp > max_par) max_par = pp;
            if (qq > max_perp) max_perp = qq;
        }
        float height = max_perp;
        float width = max_par - min_par;
        float area = height * width;
        if (area < best_area) {
            best_area = area;
            float along = min_par + width / 2.0f;
            float half_h = height / 2.0f;
            rr[0] = a.x + along * parx + half_h * perx;
            rr[1] = a.y + along * pary + half_h * pery;
            /* RotatedRect::new normalises the up axis (x / length). */
            float ul = sqrtf(perx * perx + pery * pery);
            rr[2] = perx / ul; rr[3] = pery / ul;
            rr[4] = width; rr[5] = height;
            found = 1;
        }
    }
    free(hull);
    return found;
}

ORC_API int orc_min_area_rect(const float* xy, int n, float* rr) {
    return min_area_rect_impl((const ptf*)xy, n, rr);
}

ORC_API int orc_convex_hull(const float* xy, int n, float* out_xy) {
    return convex_hull_impl((const ptf*)xy, n, (ptf*)out_xy);
}

/* find_connected_component_rects (detection.rs:41-62).  rects: n x 6 floats.
 * Returns number of rects kept (in contour discovery order) or -1. */
ORC_API int orc_component_rects(const uint8_t* mask, int h, int w, float expand, float min_area,
                                float* rects, int max_rects) {
    int64_t cap = (int64_t)h * w * 4 + 16;
    int32_t* pts = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)cap);
    int max_c = h * w / 2 + 16;
    int64_t* offs = (int64_t*)malloc(sizeof(int64_t) * (size_t)(max_c + 1));
    int nc = orc_find_contours_external(mask, h, w, pts, cap, offs, max_c);
    if (nc < 0) { free(pts); free(offs); return -1; }
    int nr = 0;
    for (int c = 0; c < nc; c++) {
        int n = (int)(offs[c + 1] - offs[c]);
        ptf* poly = (ptf*)malloc(sizeof(ptf) * (size_t)(n + 1));
        ptf* simp = (ptf*)malloc(sizeof(ptf) * (size_t)(n + 1));
        for (int k = 0; k < n; k++) {
            poly[k].y = (float)pts[2 * (offs[c] + k)];
            poly[k].x = (float)pts[2 * (offs[c] + k) + 1];
        }
        int m = simplify_polygon_impl(poly, n, 2.0f, simp);
        float rr[6];
        if (min_area_rect_impl(simp, m, rr)) {
            rr[4] = rr[4] + 2.0f * expand;
            rr[5] = rr[5] + 2.0f * expand;
            if (rr[4] * rr[5] >= min_area) {
                if (nr >= max_rects) { free(poly); free(simp); free(pts); free(offs); return -1; }
                memcpy(rects + 6 * nr, rr, sizeof(rr));
                nr++;
            }
        }
        free(poly); free(simp);
    }
    free(pts); free(offs);
    return nr;
}

/* ------------------------------------------------------------------ */
/* Polygon::fill_iter + prepare_text_line (recognition.rs:91-126)      */
/* ------------------------------------------------------------------ */

/* x of a downward edge (y0<y1) at scanline y: x0 + round_half_away((y-y0)*dx/dy). */
static inline int edge_x_at(int x0, int y0, int x1, int y1, int y) {
    float t = (float)(y - y0) * ((float)(x1 - x0) / (float)(y1 - y0));
    return x0 + (int)roundf(t);
}

/* Even-odd scanline rule: pixel (y,x), with top<=y<bottom, left<=x<right of
 * the polygon's bounding rect, is inside iff the number of non-horizontal
 * edges spanning y (ytop <= y < ybot) whose x at y is <= x is odd.
 * poly: n (y,x) int pairs.  Writes 0/1 into inside[bh*bw]. */
ORC_API void orc_polygon_fill_mask(const int32_t* poly, int n, int top, int left, int bh, int bw,
                                   uint8_t* inside) {
    memset(inside, 0, (size_t)bh * (size_t)bw);
    int* xs = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    for (int r = 0; r < bh; r++) {
        int y = top + r;
        int m = 0;
        for (int e = 0; e < n; e++) {
            int ya = poly[2 * e], xa = poly[2 * e + 1];
            int yb = poly[2 * ((e + 1) % n)], xb = poly[2 * ((e + 1) % n) + 1];
            if (ya == yb) continue;
            if (ya > yb) { int t = ya; ya = yb; yb = t; t = xa; xa = xb; xb = t; }
            if (y < ya || y >= yb) continue;
            xs[m++] = edge_x_at(xa, ya, xb, yb, y);
        }
        for (int c = 0; c < bw; c++) {
            int x = left + c, cnt = 0;
            for (int k = 0; k < m; k++) if (xs[k] <= x) cnt++;
            inside[(size_t)r * bw + c] = (uint8_t)(cnt & 1);
        }
    }
    free(xs);
}

/* recognition.rs:91-126: gather polygon pixels of the page into a line image
 * pre-filled with -0.5, then bilinear-resize to [out_h, resized_w], written
 * into `dst` with row stride `dst_stride` (recognition.rs:152-154). */
ORC_API void orc_prepare_text_line(const float* page, int ph, int pw, const int32_t* poly, int n,
                                   int resized_w, int out_h, float* dst, int dst_stride) {
    int top = poly[0], bot = poly[0], left = poly[1], right = poly[1];
    for (int k = 1; k < n; k++) {
        if (poly[2 * k] < top) top = poly[2 * k];
        if (poly[2 * k] > bot) bot = poly[2 * k];
        if (poly[2 * k + 1] < left) left = poly[2 * k + 1];
        if (poly[2 * k + 1] > right) right = poly[2 * k + 1];
    }
    int bh = bot - top, bw = right - left;
    if (bh <= 0 || bw <= 0) return;
    uint8_t* inside = (uint8_t*)malloc((size_t)bh * bw);
    float* img = (float*)malloc(sizeof(float) * (size_t)bh * bw);
    orc_polygon_fill_mask(poly, n, top, left, bh, bw, inside);
    for (int r = 0; r < bh; r++)
        for (int c = 0; c < bw; c++) {
            int y = top + r, x = left + c;
            float v = -0.5f;
            /* page_index_rect.contains_point(in_p) && contains_point(out_p)
             * (recognition.rs:100,112), both inclusive on [0,h-1]x[0,w-1]. */
            if (inside[(size_t)r * bw + c] && y >= 0 && y <= ph - 1 && x >= 0 && x <= pw - 1 &&
                r <= ph - 1 && c <= pw - 1)
                v = page[(int64_t)y * pw + x];
            img[(size_t)r * bw + c] = v;
        }
    float* tmp = (float*)malloc(sizeof(float) * (size_t)out_h * resized_w);
    orc_resize_bilinear(img, bh, bw, bw, bh, bw, -0.5f, tmp, out_h, resized_w);
    for (int r = 0; r < out_h; r++)
        memcpy(dst + (int64_t)r * dst_stride, tmp + (int64_t)r * resized_w, sizeof(float) * (size_t)resized_w);
    free(tmp); free(img); free(inside);
}

/* ------------------------------------------------------------------ */
/* Neural-network ops (the rten `Model::run` side of model.rs:33-40).  */
/* NHWC activations.  DESIGN.md §4.1 gives the accumulation orders.    */
/* ------------------------------------------------------------------ */

/* Dense conv, stride 1, zero pad (kh/2, kw/2).  w: [KH][KW][Cin][Cout].
 * acc = b[co]; for ky, kx, ci ascending: acc = fmaf(x, w, acc).  Out-of-range
 * taps contribute fmaf(0, w, acc). */
ORC_API void orc_conv2d(const float* x, int n, int h, int w, int cin, const float* wt, const float* b,
                        int kh, int kw, int cout, int relu, float* y) {
    const int ph = kh / 2, pw = kw / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int in = 0; in < n; in++)
        for (int oy = 0; oy < h; oy++) {
            float* acc = (float*)malloc(sizeof(float) * (size_t)cout);
            for (int ox = 0; ox < w; ox++) {
                for (int co = 0; co < cout; co++) acc[co] = b[co];
                for (int ky = 0; ky < kh; ky++)
                    for (int kx = 0; kx < kw; kx++) {
                        int iy = oy + ky - ph, ix = ox + kx - pw;
                        int inb = iy >= 0 && iy < h && ix >= 0 && ix < w;
                        const float* xp = inb ? x + (((int64_t)in * h + iy) * w + ix) * cin : NULL;
                        const float* wp = wt + (int64_t)(ky * kw + kx) * cin * cout;
                        for (int ci = 0; ci < cin; ci++) {
                            float xv = inb ? xp[ci] : 0.0f;
                            const float* wr = wp + (int64_t)ci * cout;
                            for (int co = 0; co < cout; co++) acc[co] = fmaf(xv, wr[co], acc[co]);
                        }
                    }
                float* yp = y + (((int64_t)in * h + oy) * w + ox) * cout;
                for (int co = 0; co < cout; co++) {
                    float v = acc[co];
                    yp[co] = relu ? (v > 0.0f ? v : 0.0f) : v;
                }
            }
            free(acc);
        }
}

/* Depthwise 3x3, pad 1.  w: [3][3][C].  acc = b[c]; ky,kx ascending. */
ORC_API void orc_dwconv3x3(const float* x, int n, int h, int w, int c, const float* wt, const float* b,
                           int relu, float* y) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int in = 0; in < n; in++)
        for (int oy = 0; oy < h; oy++)
            for (int ox = 0; ox < w; ox++) {
                float* yp = y + (((int64_t)in * h + oy) * w + ox) * c;
                for (int ch = 0; ch < c; ch++) yp[ch] = b[ch];
                for (int ky = 0; ky < 3; ky++)
                    for (int kx = 0; kx < 3; kx++) {
                        int iy = oy + ky - 1, ix = ox + kx - 1;
                        int inb = iy >= 0 && iy < h && ix >= 0 && ix < w;
                        const float* xp = inb ? x + (((int64_t)in * h + iy) * w + ix) * c : NULL;
                        const float* wp = wt + (int64_t)(ky * 3 + kx) * c;
                        for (int ch = 0; ch < c; ch++)
                            yp[ch] = fmaf(inb ? xp[ch] : 0.0f, wp[ch], yp[ch]);
                    }
                if (relu) for (int ch = 0; ch < c; ch++) yp[ch] = yp[ch] > 0.0f ? yp[ch] : 0.0f;
            }
}

/* MaxPool kernel=stride=(kh,kw), floor.  m = v0; m = v > m ? v : m. */
ORC_API void orc_maxpool(const float* x, int n, int h, int w, int c, int kh, int kw, float* y) {
    int oh = h / kh, ow = w / kw;
#pragma omp parallel for collapse(2) schedule(static)
    for (int in = 0; in < n; in++)
        for (int oy = 0; oy < oh; oy++)
            for (int ox = 0; ox < ow; ox++)
                for (int ch = 0; ch < c; ch++) {
                    float m = x[(((int64_t)in * h + oy * kh) * w + ox * kw) * c + ch];
                    for (int ky = 0; ky < kh; ky++)
                        for (int kx = 0; kx < kw; kx++) {
                            float v = x[(((int64_t)in * h + oy * kh + ky) * w + ox * kw + kx) * c + ch];
                            m = v > m ? v : m;
                        }
                    y[(((int64_t)in * oh + oy) * ow + ox) * c + ch] = m;
                }
}

/* AveragePool kernel=stride=(kh,kw): sequential sum (ky,kx ascending) * (1/(kh*kw)). */
ORC_API void orc_avgpool(const float* x, int n, int h, int w, int c, int kh, int kw, float* y) {
    int oh = h / kh, ow = w / kw;
    float inv = 1.0f / (float)(kh * kw);
    for (int in = 0; in < n; in++)
        for (int oy = 0; oy < oh; oy++)
            for (int ox = 0; ox < ow; ox++)
                for (int ch = 0; ch < c; ch++) {
                    float s = 0.0f;
                    for (int ky = 0; ky < kh; ky++)
                        for (int kx = 0; kx < kw; kx++)
                            s = s + x[(((int64_t)in * h + oy * kh + ky) * w + ox * kw + kx) * c + ch];
                    y[(((int64_t)in * oh + oy) * ow + ox) * c + ch] = s * inv;
                }
}

/* ConvTranspose 2x2 stride 2.  w: [2][2][Cin][Cout]. acc = b; ci ascending. */
ORC_API void orc_convt2x2(const float* x, int n, int h, int w, int cin, const float* wt, const float* b,
                          int cout, float* y) {
    int oh = 2 * h, ow = 2 * w;
#pragma omp parallel for collapse(2) schedule(static)
    for (int in = 0; in < n; in++)
        for (int oy = 0; oy < oh; oy++)
            for (int ox = 0; ox < ow; ox++) {
                int iy = oy >> 1, ix = ox >> 1, dy = oy & 1, dx = ox & 1;
                const float* xp = x + (((int64_t)in * h + iy) * w + ix) * cin;
                const float* wp = wt + (int64_t)(dy * 2 + dx) * cin * cout;
                float* yp = y + (((int64_t)in * oh + oy) * ow + ox) * cout;
                for (int co = 0; co < cout; co++) yp[co] = b[co];
                for (int ci = 0; ci < cin; ci++)
                    for (int co = 0; co < cout; co++)
                        yp[co] = fmaf(xp[ci], wp[(int64_t)ci * cout + co], yp[co]);
            }
}

/* Zero-pad `x` [n,h,w,cx] to the spatial size of `skip` [n,sh,sw,cs]
 * (before = diff/2, after = diff - diff/2) and concat channels [skip, x]. */
ORC_API void orc_padcat(const float* skip, int n, int sh, int sw, int cs, const float* x, int h, int w,
                        int cx, float* y) {
    int py = (sh - h) / 2, px = (sw - w) / 2;
    int ct = cs + cx;
    for (int in = 0; in < n; in++)
        for (int oy = 0; oy < sh; oy++)
            for (int ox = 0; ox < sw; ox++) {
                float* yp = y + (((int64_t)in * sh + oy) * sw + ox) * ct;
                memcpy(yp, skip + (((int64_t)in * sh + oy) * sw + ox) * cs, sizeof(float) * (size_t)cs);
                int iy = oy - py, ix = ox - px;
                if (iy >= 0 && iy < h && ix >= 0 && ix < w)
                    memcpy(yp + cs, x + (((int64_t)in * h + iy) * w + ix) * cx, sizeof(float) * (size_t)cx);
                else
                    memset(yp + cs, 0, sizeof(float) * (size_t)cx);
            }
}

/* Linear: y[r][o] = b[o] + chain_k fmaf(x[r][k], w[k][o]).  w: [K][O]. */
ORC_API void orc_linear(const float* x, int64_t rows, int k, const float* wt, const float* b, int o,
                        float* y) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++) {
        float* yp = y + r * o;
        for (int j = 0; j < o; j++) yp[j] = b[j];
        for (int kk = 0; kk < k; kk++) {
            float xv = x[r * k + kk];
            const float* wr = wt + (int64_t)kk * o;
            for (int j = 0; j < o; j++) yp[j] = fmaf(xv, wr[j], yp[j]);
        }
    }
}

/* One GRU direction (PyTorch / ONNX linear_before_reset=1 semantics).
 * x: [T][N][I]; wi: [I][3H]; bi: [3H]; wh: [H][3H]; bh: [3H]; gate order r,z,n.
 * Writes y[t][n][yoff + j], row stride ystride.  h0 = 0. */
ORC_API void orc_gru_dir(const float* x, int T, int N, int I, const float* wi, const float* bi,
                         const float* wh, const float* bh, int H, int reverse, float* y, int ystride,
                         int yoff) {
    float* gx = (float*)malloc(sizeof(float) * (size_t)T * N * 3 * H);
    orc_linear(x, (int64_t)T * N, I, wi, bi, 3 * H, gx);
    float* hcur = (float*)calloc((size_t)N * H, sizeof(float));
    float* gh = (float*)malloc(sizeof(float) * (size_t)N * 3 * H);
    for (int s = 0; s < T; s++) {
        int t = reverse ? T - 1 - s : s;
        orc_linear(hcur, N, H, wh, bh, 3 * H, gh);
#pragma omp parallel for schedule(static)
        for (int n = 0; n < N; n++) {
            const float* gxp = gx + ((int64_t)t * N + n) * 3 * H;
            const float* ghp = gh + (int64_t)n * 3 * H;
            float* hp = hcur + (int64_t)n * H;
            float* yp = y + ((int64_t)t * N + n) * ystride + yoff;
            for (int j = 0; j < H; j++) {
                float r = spec_sigmoidf(gxp[j] + ghp[j]);
                float z = spec_sigmoidf(gxp[H + j] + ghp[H + j]);
                float nn = spec_tanhf(fmaf(r, ghp[2 * H + j], gxp[2 * H + j]));
                float hn = fmaf(z, hp[j] - nn, nn);
                hp[j] = hn;
                yp[j] = hn;
            }
        }
    }
    free(gx); free(hcur); free(gh);
}

/* LogSoftmax over the last axis: m = max; s = sum_c exp(v-m) (c ascending,
 * s starts at 0); out = v - (m + log(s)). */
ORC_API void orc_log_softmax(const float* x, int64_t rows, int c, float* y) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++) {
        const float* xp = x + r * c;
        float m = xp[0];
        for (int j = 1; j < c; j++) m = xp[j] > m ? xp[j] : m;
        float s = 0.0f;
        for (int j = 0; j < c; j++) s = s + spec_expf(xp[j] - m);
        float lse = m + spec_logf(s);
        for (int j = 0; j < c; j++) y[r * c + j] = xp[j] - lse;
    }
}

/* ------------------------------------------------------------------ */
/* CTC greedy decode — rten::ctc::CtcDecoder::decode_greedy            */
/* (recognition.rs:511).  seq: [T][C] log-probs.  argmax = first max;  */
/* collapse repeats, then drop blank (0).  Returns number of steps.    */
/* ------------------------------------------------------------------ */
ORC_API int orc_ctc_greedy(const float* seq, int T, int C, int row_stride, uint32_t* labels,
                           uint32_t* pos) {
    int n = 0;
    uint32_t last = 0;
    for (int t = 0; t < T; t++) {
        const float* p = seq + (int64_t)t * row_stride;
        int best = 0; float bv = p[0];
        for (int c = 1; c < C; c++) if (p[c] > bv) { bv = p[c]; best = c; }
        if ((uint32_t)best == last) continue;
        last = (uint32_t)best;
        if (best > 0) { labels[n] = (uint32_t)best; pos[n] = (uint32_t)t; n++; }
    }
    return n;
}
