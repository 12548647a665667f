#include "engine.hpp"

#include <algorithm>
#include <limits>
#include <map>
#include <thread>

#include "kernels.hpp"

using namespace ocrs;
using namespace ocrs::geom;

// ===========================================================================
// Detection — detection.rs:104-200
// ===========================================================================
void ocrs_engine::detect_now(const ocrs_page* const* pages, size_t n, std::vector<std::vector<RotatedRect>>* rects_out,
                             float* host_map) const {
    if (!detection) fail(OCRS_ERR_MODEL_NOT_LOADED, "Detection model not loaded");
    if (n == 0) {
        if (rects_out) rects_out->clear();
        return;
    }
    const int64_t in_h64 = detection->input_shape[2], in_w64 = detection->input_shape[3];
    if (in_h64 <= 0 || in_w64 <= 0) fail(OCRS_ERR_MODEL_DIMS, "failed to get model dims");  // detection.rs:141-144
    const int in_h = (int)in_h64, in_w = (int)in_w64;
    const int N = (int)n;
    // The reference takes any image per call (detection.rs:131-171) and the model always runs at its own fixed size, so a
    // batch may hold pages of SEVERAL sizes (r6; the coalescer merges whatever waits): the model runs once over all of them,
    // the size-dependent kernels before and after it (resize in; resize back + threshold, components, contours) run once per
    // size, on that size's pages — the same launches with the same arguments as a batch of that size alone, so nobody's bits
    // change.  Internally the pages are ordered by size group (`order`); results go back in the caller's order.
    struct SizeGroup {
        int h = 0, w = 0, first = 0, count = 0;          // pages [first, first + count) of the grouped order
        int pad_bottom = 0, pad_right = 0, max_comp = 0;
        int64_t px = 0, arena = 0;
        uint8_t* d_mask = nullptr;
        k::CclBuffers b{};
        std::vector<int32_t> counts, ovf;
        std::vector<float> hr_all;
        std::vector<uint8_t> hv_all;
    };
    std::vector<SizeGroup> groups;
    std::vector<int> order(n), group_of(n);
    {
        std::vector<int> gi(n);
        for (size_t i = 0; i < n; i++) {
            const int h = pages[i]->h, w = pages[i]->w;
            if (h <= 0 || w <= 0 || h > 65535 || w > 65535) fail(OCRS_ERR_INVALID_ARGUMENT, "unsupported page size %dx%d", h, w);
            size_t g = 0;
            while (g < groups.size() && (groups[g].h != h || groups[g].w != w)) g++;
            if (g == groups.size()) { groups.emplace_back(); groups[g].h = h; groups[g].w = w; }
            groups[g].count++;
            gi[i] = (int)g;
        }
        if (host_map && groups.size() > 1)
            fail(OCRS_ERR_INVALID_ARGUMENT, "pages whose probability maps are returned in one [n, h, w] array must share a size");
        int at = 0;
        for (SizeGroup& g : groups) { g.first = at; at += g.count; g.count = 0; }
        for (size_t i = 0; i < n; i++) {
            SizeGroup& g = groups[gi[i]];
            order[g.first + g.count] = (int)i;
            group_of[g.first + g.count] = gi[i];
            g.count++;
        }
    }

    Workspace ws;
    hipStream_t st = ws.s();
    StageTimers* T = tm();
    hipStream_t ex = st;   // (round 3 could route a small request's kernels through the device's conv-stack stream — option det_heavy,
                           // off since round 4: 180 vs 194 pages/s for one-page calls from 12 threads — removed in round 5)

    // page pointer table, grouped order
    std::vector<const float*> hp(n);
    for (size_t i = 0; i < n; i++) hp[i] = pages[order[i]]->grey.as<float>();
    const float** d_ptrs = ws.alloc_n<const float*>(n);
    ws.upload(d_ptrs, hp.data(), n * sizeof(float*));

    float* d_in = ws.alloc_n<float>((size_t)N * in_h * in_w);
    {
        StageScope sc(T, ST_RESIZE_IN, ex, groups.size());
        for (SizeGroup& g : groups) {
            g.pad_bottom = std::max(in_h - g.h, 0);   // detection.rs:155-156
            g.pad_right = std::max(in_w - g.w, 0);
            k::resize_pages_to_model(d_ptrs + g.first, g.count, g.h, g.w, g.h + g.pad_bottom, g.w + g.pad_right,
                                     d_in + (size_t)g.first * in_h * in_w, in_h, in_w, ex);
        }
    }

    const float* d_prob = nullptr;
    if (detection->is_callback()) {
        // `trait Model` implemented by the caller: one run per page, host tensors (detection.rs:184).
        const auto* cb = static_cast<const CallbackModel*>(detection);
        std::vector<float> hin((size_t)in_h * in_w), hout;
        float* d_out = ws.alloc_n<float>((size_t)N * in_h * in_w);
        for (int oi = 0; oi < N; oi++) {
            int i = 0;
            while (order[i] != oi) i++;      // runs in the CALLER's page order (a caller's model may count its runs)
            ws.download(hin.data(), d_in + (size_t)i * in_h * in_w, hin.size() * sizeof(float));
            ws.sync();
            const int64_t ishape[4] = {1, 1, in_h, in_w};
            int64_t oshape[4];
            int ond = 0;
            cb->run(hin.data(), ishape, hout, oshape, &ond);
            if (ond != 4 || oshape[2] != in_h || oshape[3] != in_w || oshape[0] * oshape[1] != 1)
                fail(OCRS_ERR_WRONG_OUTPUT, "model output had unexpected type or shape: detection output is not [1,1,%d,%d]",
                     in_h, in_w);
            OCRS_HIP(hipMemcpyAsync(d_out + (size_t)i * in_h * in_w, hout.data(), hout.size() * sizeof(float),
                                    hipMemcpyHostToDevice, st));
            ws.sync();
        }
        d_prob = d_out;
    } else {
        const auto* hm = static_cast<const HipModel*>(detection);
        TensorShape os;
        d_prob = hm->run_device(ws, d_in, N, in_h, in_w, &os, T, nullptr, nullptr, true, debug, -1, nullptr);
        if (os.n != N || os.h != in_h || os.w != in_w || os.c != 1)
            fail(OCRS_ERR_WRONG_OUTPUT, "model output had unexpected type or shape: detection output [%d,%d,%d,%d]", os.n,
                 os.c, os.h, os.w);
    }

    // connected components -> rects (detection.rs:41-62)
    // Scratch is sized for pages of text: up to 65 536 components and 2 border points per pixel.  The reference
    // takes ANY mask (detection.rs:41-62), so a page that does not fit (salt noise, dense halftone) gets its
    // component stage re-run on its own with buffers for the worst case — below.
    auto alloc_ccl = [&](int np, int h, int64_t px, int mc, int64_t ar, hipStream_t cs, bool zero) {
        k::CclBuffers b{};
        b.labels = ws.alloc_n<int32_t>((size_t)np * px);
        b.row_counts = ws.alloc_n<int32_t>((size_t)np * h);
        b.row_offsets = ws.alloc_n<int32_t>((size_t)np * h);
        b.n_roots = ws.alloc_n<int32_t>(np);
        b.roots = ws.alloc_n<int32_t>((size_t)np * mc);
        b.lengths = ws.alloc_n<int32_t>((size_t)np * mc);
        b.offsets = ws.alloc_n<int32_t>((size_t)np * mc);
        b.overflow = ws.alloc_n<int32_t>(np);
        b.pts = ws.alloc_n<uint32_t>((size_t)np * ar);
        b.tmp = ws.alloc_n<uint32_t>((size_t)np * ar * 4);
        b.keep = ws.alloc_n<uint8_t>((size_t)np * ar);
        b.rects = ws.alloc_n<float>((size_t)np * mc * 6);
        b.valid = ws.alloc_n<uint8_t>((size_t)np * mc);
        if (zero) OCRS_HIP(hipMemsetAsync(b.overflow, 0, np * sizeof(int32_t), cs));
        return b;
    };
    auto run_ccl = [&](const uint8_t* mask, int np, int h, int w, const k::CclBuffers& b, int mc, int64_t ar, hipStream_t cs, bool prepared) {
        {
            StageScope sc(T, ST_CCL, cs, prepared ? 3 : 4);
            k::ccl_label(mask, np, h, w, b, mc, cs, prepared);
        }
        {
            StageScope sc(T, ST_CONTOUR_RECTS, cs, prepared ? 1 : 2);
            k::contour_rects(mask, np, h, w, b, mc, ar, /*expand*/ 3.0f, min_area, /*eps*/ 2.0f, cs, prepared);
        }
    };

    // slice off the padded region, resize back, threshold (detection.rs:187-194,110); r6: where the page width allows, the same
    // launch writes the component stage's initial labels and zeroes its counters (one launch and two fills fewer per size)
    float* d_map = host_map ? ws.alloc_n<float>((size_t)N * groups[0].h * groups[0].w) : nullptr;
    std::vector<char> prepared(groups.size(), 0);
    {
        StageScope sc(T, ST_RESIZE_THRESH, ex, groups.size());
        for (size_t gi = 0; gi < groups.size(); gi++) {
            SizeGroup& g = groups[gi];
            g.px = (int64_t)g.h * g.w;
            g.d_mask = ws.alloc_n<uint8_t>((size_t)g.count * g.px);
            if (rects_out) {
                g.max_comp = (int)std::min<int64_t>(65536, g.px / 2 + 16);
                g.arena = 2 * g.px + 64;
                g.b = alloc_ccl(g.count, g.h, g.px, g.max_comp, g.arena, ex, false);
            }
            prepared[gi] = k::resize_threshold(d_prob + (size_t)g.first * in_h * in_w, g.count, in_h, in_w, in_h - g.pad_bottom, in_w - g.pad_right,
                                               text_threshold, g.d_mask, d_map, g.h, g.w, ex, g.b.labels, g.b.overflow, g.b.offsets);
            if (rects_out && !prepared[gi]) OCRS_HIP(hipMemsetAsync(g.b.overflow, 0, g.count * sizeof(int32_t), ex));
        }
    }
    if (host_map)   // (one size group: grouped order = the caller's order)
        ws.download(host_map, d_map, (size_t)N * groups[0].px * sizeof(float));
    if (!rects_out) {
        ws.sync();
        if (T) T->collect();
        return;
    }

    // One round trip in the common case: the counts travel together with the first kSpec candidate rects of every
    // page (a page of text has a few hundred to ~1 500 components); only a page with more needs a second one.
    constexpr int kSpec = 2048;
    for (size_t gi = 0; gi < groups.size(); gi++) {
        SizeGroup& g = groups[gi];
        run_ccl(g.d_mask, g.count, g.h, g.w, g.b, g.max_comp, g.arena, ex, prepared[gi] != 0);
        const int spec = std::min(g.max_comp, kSpec);
        g.counts.resize(g.count); g.ovf.resize(g.count);
        g.hr_all.resize((size_t)g.count * spec * 6); g.hv_all.resize((size_t)g.count * spec);
        ws.download(g.counts.data(), g.b.n_roots, g.count * sizeof(int32_t));
        ws.download(g.ovf.data(), g.b.overflow, g.count * sizeof(int32_t));
        // the first `spec` candidates of every page in ONE strided copy per array (r2: two copies per page — each a blit
        // kernel that waits for CU slots like any other)
        ws.download_2d(g.hr_all.data(), g.b.rects, (size_t)g.max_comp * 6 * sizeof(float), (size_t)spec * 6 * sizeof(float), g.count);
        ws.download_2d(g.hv_all.data(), g.b.valid, (size_t)g.max_comp, (size_t)spec, g.count);
    }
    ws.sync();   // one wait for all sizes
    std::vector<int32_t> counts(N);
    std::vector<std::vector<float>> hr(N);
    std::vector<std::vector<uint8_t>> hv(N);
    rects_out->assign(n, {});
    bool more = false;
    std::vector<int> big;   // pages (grouped order) whose component stage did not fit
    for (const SizeGroup& g : groups) {
        const int spec = std::min(g.max_comp, kSpec);
        for (int j = 0; j < g.count; j++) {
            const int i = g.first + j, cnt = g.counts[j];
            counts[i] = cnt;
            if (g.ovf[j] || cnt > g.max_comp) { big.push_back(i); continue; }
            if (cnt <= spec) {
                hr[i].assign(g.hr_all.begin() + (size_t)j * spec * 6, g.hr_all.begin() + (size_t)(j + 1) * spec * 6);
                hv[i].assign(g.hv_all.begin() + (size_t)j * spec, g.hv_all.begin() + (size_t)(j + 1) * spec);
                continue;
            }
            more = true;
            hr[i].resize((size_t)cnt * 6);
            hv[i].resize(cnt);
            ws.download(hr[i].data(), g.b.rects + (size_t)j * g.max_comp * 6, hr[i].size() * sizeof(float));
            ws.download(hv[i].data(), g.b.valid + (size_t)j * g.max_comp, cnt);
        }
    }
    if (more) ws.sync();
    for (int i : big) {
        const SizeGroup& g = groups[group_of[i]];
        // Worst case of an h x w mask: no more than px / 4 + O(h + w) 8-connected components can be pairwise
        // separated, and a border walk enters a pixel at most once per direction (8 px points in total).
        const int64_t mc64 = g.px / 4 + (int64_t)g.h + g.w + 16, ar_big = 8 * g.px + 64;
        if (ar_big >= (int64_t)0x7fffffff)
            fail(OCRS_ERR_CAPACITY, "text mask of page %d: %lld pixels exceed the 32-bit contour arena", order[i], (long long)g.px);
        const int mc = (int)mc64;
        const k::CclBuffers bb = alloc_ccl(1, g.h, g.px, mc, ar_big, st, true);
        run_ccl(g.d_mask + (size_t)(i - g.first) * g.px, 1, g.h, g.w, bb, mc, ar_big, st, false);
        int32_t cnt = 0, o = 0;
        ws.download(&cnt, bb.n_roots, sizeof cnt);
        ws.download(&o, bb.overflow, sizeof o);
        ws.sync();
        if (o || cnt > mc)
            fail(OCRS_ERR_DEVICE, "internal: component stage of page %d overflowed its worst-case buffers (%d components)", order[i], cnt);
        counts[i] = cnt;
        hr[i].resize((size_t)cnt * 6);
        hv[i].resize(cnt);
        ws.download(hr[i].data(), bb.rects, hr[i].size() * sizeof(float));
        ws.download(hv[i].data(), bb.valid, cnt);
        ws.sync();
    }
    for (int i = 0; i < N; i++) {
        auto& out = (*rects_out)[order[i]];
        for (int c = 0; c < counts[i]; c++)
            if (hv[i][c]) out.push_back(RotatedRect::from_array(&hr[i][(size_t)c * 6]));
    }
    if (T) T->collect();
}

// ===========================================================================
// Coalescing front ends (coalesce.hpp): concurrent small requests share one launch sequence
// ===========================================================================
namespace {

// Runs `merged` for the whole batch; if that fails and the batch has several requests, every request is re-run
// on its own so that an error reaches only the caller whose input caused it.
template <class Req, class Merged, class Single>
void run_batch(std::vector<Req*>& batch, Merged&& merged, Single&& single) {
    try {
        merged();
        return;
    } catch (...) {
        if (batch.size() == 1) {
            batch[0]->error = std::current_exception();
            return;
        }
    }
    for (Req* r : batch) {
        try {
            single(*r);
        } catch (...) {
            r->error = std::current_exception();
        }
    }
}

}  // namespace

void ocrs_engine::init_coalescers() {
    det_queue = std::make_unique<Coalescer<DetRequest>>(
        [this](std::vector<DetRequest*>& batch) {
            run_batch(
                batch,
                [&] {
                    if (batch.size() == 1) { detect_now(batch[0]->pages, batch[0]->n, batch[0]->rects, nullptr); return; }
                    std::vector<const ocrs_page*> pages;
                    for (DetRequest* r : batch) pages.insert(pages.end(), r->pages, r->pages + r->n);
                    std::vector<std::vector<RotatedRect>> rects;
                    detect_now(pages.data(), pages.size(), &rects, nullptr);
                    size_t at = 0;
                    for (DetRequest* r : batch) {
                        r->rects->assign(std::make_move_iterator(rects.begin() + at), std::make_move_iterator(rects.begin() + at + r->n));
                        at += r->n;
                    }
                },
                [&](DetRequest& r) { detect_now(r.pages, r.n, r.rects, nullptr); });
        },
        [](const DetRequest&, const DetRequest&) { return true; });   // pages of any sizes share a batch (detect_now groups them by size)
    rec_queue = std::make_unique<Coalescer<RecRequest>>(
        [this](std::vector<RecRequest*>& batch) {
            run_batch(
                batch,
                [&] {
                    if (batch.size() == 1) {
                        RecRequest& r = *batch[0];
                        recognize_now(r.pages, r.n_pages, *r.lines_per_page, r.steps, r.rec_lines, r.ctc_len);
                        return;
                    }
                    std::vector<const ocrs_page*> pages;
                    std::vector<std::vector<std::vector<RotatedRect>>> lpp;
                    for (RecRequest* r : batch) {
                        pages.insert(pages.end(), r->pages, r->pages + r->n_pages);
                        lpp.insert(lpp.end(), r->lines_per_page->begin(), r->lines_per_page->end());
                    }
                    std::vector<std::vector<CtcStep>> steps;
                    std::vector<RecLine> rl;
                    std::vector<uint32_t> cl;
                    recognize_now(pages.data(), pages.size(), lpp, &steps, &rl, &cl);
                    size_t line0 = 0, page0 = 0;
                    for (RecRequest* r : batch) {
                        size_t nl = 0;
                        for (const auto& pg : *r->lines_per_page) nl += pg.size();
                        r->steps->assign(std::make_move_iterator(steps.begin() + line0), std::make_move_iterator(steps.begin() + line0 + nl));
                        r->ctc_len->assign(cl.begin() + line0, cl.begin() + line0 + nl);
                        r->rec_lines->assign(std::make_move_iterator(rl.begin() + line0), std::make_move_iterator(rl.begin() + line0 + nl));
                        for (RecLine& l : *r->rec_lines) {   // back to the caller's numbering
                            l.page -= page0;
                            l.index -= line0;
                        }
                        line0 += nl;
                        page0 += r->n_pages;
                    }
                },
                [&](RecRequest& r) { recognize_now(r.pages, r.n_pages, *r.lines_per_page, r.steps, r.rec_lines, r.ctc_len); });
        },
        [](const RecRequest&, const RecRequest&) { return true; });   // lines of any pages share a ragged batch
}

void ocrs_engine::detect(const ocrs_page* const* pages, size_t n, std::vector<std::vector<RotatedRect>>* rects_out,
                         float* host_map) const {
    const int max_active = option(OPT_COALESCE);   // (r4: 4 / 6 / 12 detection batches in flight instead of 2: 188-193 pages/s from 12 threads either way)
    const size_t max_pages = (size_t)std::max(1, option(OPT_COALESCE_PAGES));
    // merged only where it cannot be observed: HIP executor (a caller's `trait Model` sees every run), rects only
    if (max_active <= 0 || !det_queue || !rects_out || host_map || n == 0 || 2 * n >= max_pages || !detection ||
        detection->is_callback() || debug) {
        detect_now(pages, n, rects_out, host_map);
        return;
    }
    DetRequest r;
    r.pages = pages; r.n = n; r.rects = rects_out; r.weight = n;
    det_queue->submit(r, max_active, max_pages, option_long(OPT_COALESCE_WINDOW_US));
}

void ocrs_engine::recognize(const ocrs_page* const* pages, size_t n_pages,
                            const std::vector<std::vector<std::vector<RotatedRect>>>& lines_per_page,
                            std::vector<std::vector<CtcStep>>* steps_out, std::vector<RecLine>* rec_lines_out,
                            std::vector<uint32_t>* ctc_len_out) const {
    const int max_active = option(OPT_COALESCE);
    const size_t max_pages = (size_t)std::max(1, option(OPT_COALESCE_PAGES));
    if (max_active <= 0 || !rec_queue || n_pages == 0 || 2 * n_pages >= max_pages || !recognition || recognition->is_callback()) {
        recognize_now(pages, n_pages, lines_per_page, steps_out, rec_lines_out, ctc_len_out);
        return;
    }
    RecRequest r;
    r.pages = pages; r.n_pages = n_pages; r.lines_per_page = &lines_per_page;
    r.steps = steps_out; r.rec_lines = rec_lines_out; r.ctc_len = ctc_len_out; r.weight = n_pages;
    rec_queue->submit(r, max_active, max_pages, option_long(OPT_COALESCE_WINDOW_US));
}

// ===========================================================================
// Recognition — recognition.rs
// ===========================================================================
uint32_t ocrs_engine::rec_input_height() const {  // recognition.rs:332-337
    const int64_t hgt = recognition->input_shape[2];
    return hgt > 0 ? (uint32_t)hgt : 50u;
}

namespace {

// recognition.rs:58-75
uint32_t resized_line_width(int32_t orig_width, int32_t orig_height, int32_t height) {
    const float aspect = (float)orig_width / (float)orig_height;
    float v = (float)height * aspect;
    if (v != v) return 0;  // clamp keeps NaN; `as u32` maps it to 0
    v = v < 10.0f ? 10.0f : v;
    v = v > 2400.0f ? 2400.0f : v;
    return (uint32_t)v;
}

// recognition.rs:29-55
std::vector<PointI> line_polygon(const std::vector<RotatedRect>& words) {
    std::vector<PointI> poly;
    poly.reserve(words.size() * 4);
    auto floor_point = [](PointF p) { return PointI{as_i32(p.x), as_i32(p.y)}; };
    for (const RotatedRect& w : words) {
        LineF left = downwards_line(leftmost_edge(w)), right = downwards_line(rightmost_edge(w));
        poly.push_back(floor_point(left.start));
        poly.push_back(floor_point(right.start));
    }
    for (auto it = words.rbegin(); it != words.rend(); ++it) {
        LineF left = downwards_line(leftmost_edge(*it)), right = downwards_line(rightmost_edge(*it));
        poly.push_back(floor_point(right.end));
        poly.push_back(floor_point(left.end));
    }
    return poly;
}

// recognition.rs:162-193
bool polygon_slice_bounding_rect(const std::vector<PointI>& poly, int32_t min_x, int32_t max_x, Rect* out) {
    bool have = false;
    Rect acc{0, 0, 0, 0};
    const size_t n = poly.size();
    for (size_t k = 0; k < n; k++) {
        PointI s = poly[k], e = poly[(k + 1) % n];
        if (s.x > e.x) std::swap(s, e);  // rightwards()
        if ((s.x < min_x && e.x < min_x) || (s.x > max_x && e.x > max_x)) continue;
        LineF ef{PointF{(float)s.x, (float)s.y}, PointF{(float)e.x, (float)e.y}};
        PointI ts = s, te = e;
        if (auto y = ef.y_for_x((float)min_x)) ts = PointI{min_x, (int32_t)rround(*y)};
        if (auto y = ef.y_for_x((float)max_x)) te = PointI{max_x, (int32_t)rround(*y)};
        Rect br{std::min(ts.y, te.y), std::min(ts.x, te.x), std::max(ts.y, te.y), std::max(ts.x, te.x)};
        acc = have ? acc.unite(br) : br;
        have = true;
    }
    if (have) *out = acc;
    return have;
}

}  // namespace

RecLine ocrs_engine::make_rec_line(const std::vector<RotatedRect>& words, size_t page, size_t index) const {
    if (words.empty()) fail(OCRS_ERR_INVALID_ARGUMENT, "line has no words");  // recognition.rs:433
    RectF br = words[0].bounding_rect();
    for (size_t i = 1; i < words.size(); i++) br = br.unite(words[i].bounding_rect());
    const Rect line_rect = br.integral_bounding_rect();
    RecLine l;
    l.page = page;
    l.index = index;
    l.resized_width = resized_line_width(line_rect.width(), line_rect.height(), (int32_t)rec_input_height());
    l.group_width = (l.resized_width + 49) / 50 * 50;  // next_multiple_of(50), recognition.rs:437
    l.polygon = line_polygon(words);
    int32_t t = l.polygon[0].y, bt = t, lf = l.polygon[0].x, rt = lf;
    for (const PointI& p : l.polygon) {
        t = std::min(t, p.y); bt = std::max(bt, p.y);
        lf = std::min(lf, p.x); rt = std::max(rt, p.x);
    }
    l.bounds = Rect{t, lf, bt, rt};
    return l;
}

// Activations of the recognition conv stack scale with the input (~100 B per input pixel of the padded line
// batch).  A request beyond the budget is run as consecutive sub-requests over contiguous runs of its lines
// (lines are independent: recognition.rs:448-503 itself works in chunks of 20), so the caller never has to
// know the limit.  Option "rec_max_pixels" overrides the budget (tests).
static double rec_pixel_budget() {
    const long v = option_long(OPT_REC_MAX_PIXELS);
    return v > 0 ? (double)v : 2.0e9;
}

void ocrs_engine::recognize_now(const ocrs_page* const* pages, size_t n_pages,
                                const std::vector<std::vector<std::vector<RotatedRect>>>& lines_per_page,
                                std::vector<std::vector<CtcStep>>* steps_out, std::vector<RecLine>* rec_lines_out,
                                std::vector<uint32_t>* ctc_len_out) const {
    if (!recognition) fail(OCRS_ERR_MODEL_NOT_LOADED, "Recognition model not loaded");
    const uint32_t rec_h = rec_input_height();
    std::vector<RecLine> lines;
    for (size_t p = 0; p < n_pages; p++)
        for (const auto& words : lines_per_page[p]) lines.push_back(make_rec_line(words, p, lines.size()));
    const size_t L = lines.size();
    const double budget = rec_pixel_budget();
    double total = 0.0;
    for (const RecLine& l : lines) total += (double)rec_h * l.group_width;
    if (total <= budget) {
        recognize_lines(pages, n_pages, lines, steps_out, ctc_len_out);
        *rec_lines_out = std::move(lines);
        return;
    }
    steps_out->assign(L, {});
    ctc_len_out->assign(L, 0);
    for (size_t b = 0; b < L;) {
        size_t e = b;
        double px = 0.0;
        while (e < L && (e == b || px + (double)rec_h * lines[e].group_width <= budget)) px += (double)rec_h * lines[e++].group_width;
        std::vector<RecLine> part(lines.begin() + b, lines.begin() + e);
        std::vector<std::vector<CtcStep>> st;
        std::vector<uint32_t> cl;
        recognize_lines(pages, n_pages, part, &st, &cl);
        for (size_t i = b; i < e; i++) {
            (*steps_out)[i] = std::move(st[i - b]);
            (*ctc_len_out)[i] = cl[i - b];
        }
        b = e;
    }
    *rec_lines_out = std::move(lines);
}

void ocrs_engine::recognize_logits(const ocrs_page* page, const std::vector<std::vector<RotatedRect>>& lines_in,
                                   std::vector<std::vector<float>>* logp, int* classes) const {
    if (!recognition) fail(OCRS_ERR_MODEL_NOT_LOADED, "Recognition model not loaded");
    if (recognition->is_callback()) fail(OCRS_ERR_INVALID_ARGUMENT, "recognize_logits needs a model of the fixed-graph executor");
    std::vector<RecLine> lines;
    for (const auto& words : lines_in) lines.push_back(make_rec_line(words, 0, lines.size()));
    std::vector<std::vector<CtcStep>> steps;
    std::vector<uint32_t> ctc_len;
    recognize_lines(&page, 1, lines, &steps, &ctc_len, logp);
    *classes = (int)alphabet.size() + 1;
}

void ocrs_engine::recognize_lines(const ocrs_page* const* pages, size_t n_pages, const std::vector<RecLine>& lines,
                                  std::vector<std::vector<CtcStep>>* steps_out, std::vector<uint32_t>* ctc_len_out,
                                  std::vector<std::vector<float>>* logp_out) const {
    const bool beam = decode_method == OCRS_DECODE_BEAM_SEARCH;
    if (logp_out) logp_out->assign(lines.size(), {});
    const uint32_t rec_h = rec_input_height();
    const size_t alphabet_len = alphabet.size();

    const size_t L = lines.size();
    steps_out->assign(L, {});
    ctc_len_out->assign(L, 0);
    if (L == 0) return;

    // group by padded width (recognition.rs:430-446); std::map gives a deterministic order
    std::map<uint32_t, std::vector<size_t>> groups;
    for (size_t i = 0; i < L; i++) groups[lines[i].group_width].push_back(i);

    Workspace ws;
    hipStream_t st = ws.s();
    StageTimers* T = tm();

    std::vector<const float*> hp(n_pages);
    std::vector<int32_t> hhw(2 * n_pages);
    for (size_t i = 0; i < n_pages; i++) {
        hp[i] = pages[i]->grey.as<float>();
        hhw[2 * i] = pages[i]->h;
        hhw[2 * i + 1] = pages[i]->w;
    }
    const float** d_pages = ws.alloc_n<const float*>(n_pages);
    int32_t* d_hw = ws.alloc_n<int32_t>(2 * n_pages);
    ws.upload(d_pages, hp.data(), n_pages * sizeof(float*));
    ws.upload(d_hw, hhw.data(), hhw.size() * sizeof(int32_t));

    const bool callback = recognition->is_callback();
    const uint8_t* d_excl = has_excluded ? d_excluded.as<uint8_t>() : nullptr;

    // ---- crop + resize + pad every line into its width group's batch (recognition.rs:135-158),
    // one launch for all lines of all groups
    struct Chunk { uint32_t gw; std::vector<size_t> members; int64_t off; float* ptr = nullptr; };
    std::vector<Chunk> chunks;
    {
        int64_t off = 0;
        for (auto& kv : groups) {
            const uint32_t gw = kv.first;
            if (gw == 0) continue;  // zero-width lines produce no input and no text
            // reference: chunks of 20 (recognition.rs:450); rows are independent, so the HIP
            // executor takes bigger chunks, bounded by activation memory (~8 KB per input pixel column).
            const size_t max_chunk = callback ? 20 : std::max<size_t>(1, 2457600 / gw);
            for (size_t c0 = 0; c0 < kv.second.size(); c0 += max_chunk) {
                Chunk ch;
                ch.gw = gw;
                ch.members.assign(kv.second.begin() + c0, kv.second.begin() + std::min(kv.second.size(), c0 + max_chunk));
                ch.off = off;
                off += (int64_t)ch.members.size() * rec_h * gw;
                chunks.push_back(std::move(ch));
            }
        }
        // (ocrs_engine::recognize keeps a request within the activation budget; a single line beyond it is refused)
        if ((double)off > std::max(rec_pixel_budget(), 2.0e9))
            fail(OCRS_ERR_CAPACITY, "text line too large for the recognition model (%lld input pixels)", (long long)off);
        std::vector<k::LineDesc> descs;
        std::vector<int32_t> poly;
        for (const Chunk& ch : chunks)
            for (size_t j = 0; j < ch.members.size(); j++) {
                const RecLine& ln = lines[ch.members[j]];
                k::LineDesc d{};
                d.page = (int32_t)ln.page;
                d.poly_off = (int32_t)(poly.size() / 2);
                d.poly_n = (int32_t)ln.polygon.size();
                d.top = ln.bounds.top; d.left = ln.bounds.left;
                d.bh = ln.bounds.height(); d.bw = ln.bounds.width();
                d.resized_w = (int32_t)ln.resized_width;
                d.out_w = (int32_t)ch.gw;
                d.out_off = ch.off + (int64_t)j * rec_h * ch.gw;
                descs.push_back(d);
                for (const PointI& p : ln.polygon) { poly.push_back(p.y); poly.push_back(p.x); }
            }
        if (descs.empty()) return;
        k::LineDesc* d_descs = ws.alloc_n<k::LineDesc>(descs.size());
        int32_t* d_poly = ws.alloc_n<int32_t>(poly.size());
        // host temporaries travel through the workspace's pinned staging: real asynchronous copies, no host wait here
        ws.upload(d_descs, descs.data(), descs.size() * sizeof(k::LineDesc));
        ws.upload(d_poly, poly.data(), poly.size() * sizeof(int32_t));
        float* d_all = ws.alloc_n<float>((size_t)off);
        {
            StageScope sc(T, ST_LINE_CROP, st);
            k::crop_lines(d_pages, d_hw, d_descs, d_poly, (int)descs.size(), (int)rec_h, d_all, st);
        }
        if (callback) ws.sync();   // the callback path reads the crops back right away
        for (Chunk& ch : chunks) ch.ptr = d_all + ch.off;
    }
    auto chunk_ptr = [](const Chunk& ch) { return ch.ptr; };

    if (callback) {
        // ---- `trait Model` implemented by the caller: one run per <=20-line chunk (recognition.rs:485)
        const auto* cb = static_cast<const CallbackModel*>(recognition);
        for (const Chunk& ch : chunks) {
            const size_t nb = ch.members.size();
            const uint32_t gw = ch.gw;
            std::vector<float> hin(nb * rec_h * gw), hout;
            ws.download(hin.data(), chunk_ptr(ch), hin.size() * sizeof(float));
            ws.sync();
            const int64_t ishape[4] = {(int64_t)nb, 1, rec_h, gw};
            int64_t oshape[4];
            int ond = 0;
            cb->run(hin.data(), ishape, hout, oshape, &ond);
            if (ond != 3)
                fail(OCRS_ERR_WRONG_OUTPUT,
                     "model output had unexpected type or shape: expected recognition output to have 3 dims but it has %d", ond);
            if ((size_t)oshape[1] != nb)
                fail(OCRS_ERR_WRONG_OUTPUT, "model output had unexpected type or shape: batch size %lld != %zu",
                     (long long)oshape[1], nb);
            const int Tn = (int)oshape[0], C = (int)oshape[2];
            if (alphabet_len + 1 != (size_t)C)
                fail(OCRS_ERR_WRONG_OUTPUT,
                     "model output had unexpected type or shape: output column count (%d) does not match alphabet size (%zu)",
                     C, alphabet_len + 1);
            if (beam) {  // decode_beam on the model output, masked as recognition.rs:547-561 does
                std::vector<float> seq((size_t)Tn * C);
                for (size_t j = 0; j < nb; j++) {
                    for (int t = 0; t < Tn; t++)
                        for (int c = 0; c < C; c++) {
                            float v = hout[((size_t)t * nb + j) * C + c];
                            if (has_excluded && excluded[c]) v = -std::numeric_limits<float>::infinity();
                            seq[(size_t)t * C + c] = v;
                        }
                    const size_t li = ch.members[j];
                    (*steps_out)[li] = ctc_beam_search(seq.data(), Tn, C, C, beam_width);
                    (*ctc_len_out)[li] = (uint32_t)Tn;
                }
                continue;
            }
            float* d_logp = ws.alloc_n<float>(hout.size());
            OCRS_HIP(hipMemcpyAsync(d_logp, hout.data(), hout.size() * sizeof(float), hipMemcpyHostToDevice, st));
            int32_t* d_labels = ws.alloc_n<int32_t>((size_t)Tn * nb);
            uint32_t* d_ol = ws.alloc_n<uint32_t>((size_t)nb * Tn);
            uint32_t* d_op = ws.alloc_n<uint32_t>((size_t)nb * Tn);
            int32_t* d_cnt = ws.alloc_n<int32_t>(nb);
            {
                StageScope sc(T, ST_CTC, st, 2);
                k::argmax_rows(d_logp, (int64_t)Tn * nb, C, d_excl, d_labels, st);
                k::ctc_collapse(d_labels, Tn, (int)nb, d_ol, d_op, d_cnt, st);  // recognition.rs:511
            }
            std::vector<uint32_t> hl((size_t)nb * Tn), hpz((size_t)nb * Tn);
            std::vector<int32_t> hc(nb);
            ws.download(hl.data(), d_ol, hl.size() * 4);
            ws.download(hpz.data(), d_op, hpz.size() * 4);
            ws.download(hc.data(), d_cnt, nb * 4);
            ws.sync();
            for (size_t j = 0; j < nb; j++) {
                const size_t li = ch.members[j];
                auto& sv = (*steps_out)[li];
                sv.resize(hc[j]);
                for (int q = 0; q < hc[j]; q++) sv[q] = CtcStep{hl[j * Tn + q], hpz[j * Tn + q]};
                (*ctc_len_out)[li] = (uint32_t)Tn;
            }
        }
    } else {
        // ---- fixed-graph HIP executor: all width groups in ONE ragged batch.  Each line keeps the
        // padded width (hence sequence length) the reference gives it (recognition.rs:437), lines are
        // sorted by length so that step t of the recurrence works on a dense prefix of rows.
        const auto* hm = static_cast<const HipModel*>(recognition);
        if (hm->packed_split() < 0)
            fail(OCRS_ERR_WRONG_OUTPUT,
                 "model output had unexpected type or shape: expected recognition output to have 3 dims but it has 4");
        std::vector<int> chunk_T(chunks.size());
        int C = 0;
        for (size_t c = 0; c < chunks.size(); c++) {
            TensorShape os = hm->infer(1, (int)rec_h, (int)chunks[c].gw);
            if (!os.seq)
                fail(OCRS_ERR_WRONG_OUTPUT,
                     "model output had unexpected type or shape: expected recognition output to have 3 dims but it has 4");
            chunk_T[c] = os.n;
            C = os.c;
        }
        if (alphabet_len + 1 != (size_t)C)  // recognition.rs:487-493
            fail(OCRS_ERR_WRONG_OUTPUT,
                 "model output had unexpected type or shape: output column count (%d) does not match alphabet size (%zu)", C,
                 alphabet_len + 1);
        // Two ragged batches when the request mixes long and short lines: the recurrence is a chain
        // of up to 600 dependent, latency-bound steps whose tail only involves the few longest lines.
        // The long groups run first on a high-priority stream; their recurrence then overlaps the
        // conv stack of the short groups (the bulk of the FLOPs) on this call's main stream.
        struct Slot { int T; size_t line; size_t chunk, j; };
        struct Sub {
            std::vector<Slot> slots;
            std::vector<uint32_t> hl, hp;
            std::vector<int32_t> hc;
            int Tmax = 0;
            uint32_t gru_status[8] = {0};  // time-out words of the persistent GRU kernels (0 = fine)
            bool gpu_beam = false;         // beam search already done on the GPU: hl/hp/hc hold its steps
            std::vector<float> logp;      // beam search on the host / logits wanted: packed [R][C]
            std::vector<int32_t> off;     // with logp
        };
        const int T_SPLIT = 160;
        size_t first_long = chunks.size();
        for (size_t c = 0; c < chunks.size(); c++)
            if (chunk_T[c] > T_SPLIT) { first_long = c; break; }
        size_t n_long_lines = 0, n_short_lines = 0;
        for (size_t c = 0; c < chunks.size(); c++) (c >= first_long ? n_long_lines : n_short_lines) += chunks[c].members.size();
        const bool split = n_long_lines > 0 && n_short_lines >= 64;
        Workspace ws_long(true);  // high-priority stream (idle if unused)

        auto launch = [&](size_t c0, size_t c1, Workspace& w, Sub& sub) {
            hipStream_t sst = w.s();
            for (size_t c = c0; c < c1; c++)
                for (size_t j = 0; j < chunks[c].members.size(); j++)
                    if (chunk_T[c] > 0) sub.slots.push_back(Slot{chunk_T[c], chunks[c].members[j], c, j});
            std::stable_sort(sub.slots.begin(), sub.slots.end(), [](const Slot& x, const Slot& y) { return x.T > y.T; });
            const int M = (int)sub.slots.size();
            if (M == 0) return;
            HipModel::PackedPlan plan;
            plan.h_status = sub.gru_status;
            plan.M = M;
            plan.Tmax = sub.slots[0].T;
            plan.active.assign(plan.Tmax, 0);
            std::vector<int32_t> hTm(M), hoff(plan.Tmax + 1, 0);
            std::vector<std::vector<int32_t>> hpos(c1 - c0);
            for (size_t c = c0; c < c1; c++) hpos[c - c0].assign(chunks[c].members.size(), 0);
            for (int m = 0; m < M; m++) {
                hTm[m] = sub.slots[m].T;
                hpos[sub.slots[m].chunk - c0][sub.slots[m].j] = m;
                for (int t = 0; t < sub.slots[m].T; t++) plan.active[t]++;
            }
            for (int t = 0; t < plan.Tmax; t++) hoff[t + 1] = hoff[t] + plan.active[t];
            plan.R = hoff[plan.Tmax];
            std::vector<int32_t> meta;  // Tm | off | pos of every chunk (contiguous, chunk order)
            meta.insert(meta.end(), hTm.begin(), hTm.end());
            meta.insert(meta.end(), hoff.begin(), hoff.end());
            std::vector<size_t> pos_at(c1 - c0);
            for (size_t c = c0; c < c1; c++) {
                pos_at[c - c0] = meta.size();
                meta.insert(meta.end(), hpos[c - c0].begin(), hpos[c - c0].end());
            }
            int32_t* d_meta = w.alloc_n<int32_t>(meta.size());
            w.upload(d_meta, meta.data(), meta.size() * sizeof(int32_t));
            plan.d_Tm = d_meta;
            plan.h_Tm = hTm;
            plan.d_off = d_meta + M;
            std::vector<HipModel::PackedGroup> pg;
            for (size_t c = c0; c < c1; c++)
                if (chunk_T[c] > 0)
                    pg.push_back(HipModel::PackedGroup{chunk_ptr(chunks[c]), (int)chunks[c].members.size(), (int)chunks[c].gw,
                                                       d_meta + pos_at[c - c0]});
            int32_t* d_labels = w.alloc_n<int32_t>((size_t)plan.R);
            float* d_logp = nullptr;
            hm->run_recognition_packed(w, pg, plan, (int)rec_h, T, d_excl, d_labels, (beam || logp_out) ? &d_logp : nullptr);
            if (logp_out) {
                sub.logp.resize((size_t)plan.R * C);
                sub.off = hoff;
                w.download(sub.logp.data(), d_logp, sub.logp.size() * sizeof(float), sst);
            }
            if (beam && option(OPT_BEAM_GPU) && k::ctc_beam_supported(C, (int)beam_width)) {
                // rten decode_beam (recognition.rs:512-514) on the GPU, one workgroup per line (kernels_beam.hip)
                const int Tmax = plan.Tmax;
                sub.Tmax = Tmax;
                sub.gpu_beam = true;
                const size_t arena = k::ctc_beam_arena_entries(Tmax, (int)beam_width);
                int2* d_nodes = w.alloc_n<int2>((size_t)M * arena);
                int2* d_posn = w.alloc_n<int2>((size_t)M * arena);
                uint32_t* d_ol = w.alloc_n<uint32_t>((size_t)M * Tmax);
                uint32_t* d_op = w.alloc_n<uint32_t>((size_t)M * Tmax);
                int32_t* d_cnt = w.alloc_n<int32_t>(M);
                {
                    StageScope sc(T, ST_CTC, sst);
                    k::ctc_beam_packed(d_logp, plan.d_Tm, plan.d_off, M, Tmax, C, (int)beam_width, d_excl, d_nodes, d_posn, d_ol,
                                       d_op, d_cnt, sst);
                }
                sub.hl.resize((size_t)M * Tmax);
                sub.hp.resize((size_t)M * Tmax);
                sub.hc.resize(M);
                w.download(sub.hl.data(), d_ol, sub.hl.size() * 4, sst);
                w.download(sub.hp.data(), d_op, sub.hp.size() * 4, sst);
                w.download(sub.hc.data(), d_cnt, (size_t)M * 4, sst);
                return;
            }
            if (beam) {
                sub.Tmax = plan.Tmax;
                if (!logp_out) {
                    sub.logp.resize((size_t)plan.R * C);
                    sub.off = hoff;
                    w.download(sub.logp.data(), d_logp, sub.logp.size() * sizeof(float), sst);
                }
                return;
            }
            // greedy CTC (recognition.rs:511)
            const int Tmax = plan.Tmax;
            sub.Tmax = Tmax;
            uint32_t* d_ol = w.alloc_n<uint32_t>((size_t)M * Tmax);
            uint32_t* d_op = w.alloc_n<uint32_t>((size_t)M * Tmax);
            int32_t* d_cnt = w.alloc_n<int32_t>(M);
            {
                StageScope sc(T, ST_CTC, sst);
                k::ctc_collapse_packed(d_labels, plan.d_Tm, plan.d_off, M, Tmax, d_ol, d_op, d_cnt, sst);
            }
            sub.hl.resize((size_t)M * Tmax);
            sub.hp.resize((size_t)M * Tmax);
            sub.hc.resize(M);
            w.download(sub.hl.data(), d_ol, sub.hl.size() * 4, sst);
            w.download(sub.hp.data(), d_op, sub.hp.size() * 4, sst);
            w.download(sub.hc.data(), d_cnt, (size_t)M * 4, sst);
        };
        auto unpack = [&](const Sub& sub) {
            for (uint32_t st8 : sub.gru_status)
                if (st8) fail(OCRS_ERR_DEVICE, "GRU recurrence kernel timed out waiting for a peer workgroup (status 0x%x)", st8);
            if (logp_out)
                for (size_t m = 0; m < sub.slots.size(); m++) {
                    auto& dst = (*logp_out)[sub.slots[m].line];
                    dst.resize((size_t)sub.slots[m].T * C);
                    for (int t = 0; t < sub.slots[m].T; t++)
                        memcpy(&dst[(size_t)t * C], &sub.logp[((size_t)sub.off[t] + m) * C], (size_t)C * sizeof(float));
                }
            if (beam && !sub.gpu_beam) {  // rten decode_beam (recognition.rs:512-514), host side, one thread per slice of lines
                const size_t M = sub.slots.size();
                const unsigned nth = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), (size_t)32, M}));
                std::vector<std::thread> th;
                for (unsigned w0 = 0; w0 < nth; w0++)
                    th.emplace_back([&, w0] {
                        std::vector<float> seq;
                        for (size_t m = w0; m < M; m += nth) {
                            const int Tm = sub.slots[m].T;
                            seq.resize((size_t)Tm * C);
                            for (int t = 0; t < Tm; t++) {
                                const float* src = &sub.logp[((size_t)sub.off[t] + m) * C];
                                for (int c = 0; c < C; c++)
                                    seq[(size_t)t * C + c] = (has_excluded && excluded[c]) ? -std::numeric_limits<float>::infinity() : src[c];
                            }
                            const size_t li = sub.slots[m].line;
                            (*steps_out)[li] = ctc_beam_search(seq.data(), Tm, C, C, beam_width);
                            (*ctc_len_out)[li] = (uint32_t)Tm;
                        }
                    });
                for (auto& t : th) t.join();
                return;
            }
            for (size_t m = 0; m < sub.slots.size(); m++) {
                const size_t li = sub.slots[m].line;
                auto& sv = (*steps_out)[li];
                sv.resize(sub.hc[m]);
                for (int q = 0; q < sub.hc[m]; q++)
                    sv[q] = CtcStep{sub.hl[m * sub.Tmax + q], sub.hp[m * sub.Tmax + q]};
                (*ctc_len_out)[li] = (uint32_t)sub.slots[m].T;
            }
        };
        Sub sub_long, sub_short;
        if (split) {
            {   // the crops were produced on `st`: the long lines' stream starts after them
                hipEvent_t crops = ws.make_event();
                OCRS_HIP(hipEventRecord(crops, st));
                OCRS_HIP(hipStreamWaitEvent(ws_long.s(), crops, 0));
            }
            launch(first_long, chunks.size(), ws_long, sub_long);
            launch(0, first_long, ws, sub_short);
            ws_long.sync();
            ws.sync();
            unpack(sub_long);
            unpack(sub_short);
        } else {
            launch(0, chunks.size(), ws, sub_short);
            ws.sync();
            unpack(sub_short);
        }
    }
    if (T) T->collect();
}

// recognition.rs:241-311 for one line
std::vector<TextChar> ocrs_engine::text_line_from_result(const RecLine& line, uint32_t ctc_input_len,
                                                         const std::vector<CtcStep>& steps) const {
    std::vector<TextChar> out;
    if (steps.empty() || ctc_input_len == 0) return out;
    const Rect line_rect = line.bounds;
    const float x_scale = (float)line_rect.width() / (float)line.resized_width;
    const uint32_t downsample = (uint32_t)rround((float)line.group_width / (float)ctc_input_len);
    for (size_t i = 0; i < steps.size(); i++) {
        const uint32_t start_x = steps[i].pos * downsample;
        const uint32_t end_x = i + 1 < steps.size() ? steps[i + 1].pos * downsample : line.resized_width;
        const int32_t sx = line_rect.left + as_i32((float)start_x * x_scale);
        const int32_t ex = line_rect.left + as_i32((float)end_x * x_scale);
        if (sx >= line_rect.right) continue;  // character starts in the padding
        const uint32_t idx = steps[i].label - 1;
        const uint32_t ch = idx < alphabet.size() ? (uint32_t)alphabet[idx] : (uint32_t)'?';
        Rect r;
        if (!polygon_slice_bounding_rect(line.polygon, sx, ex, &r))
            fail(OCRS_ERR_RUN_FAILED, "invalid X coords");  // recognition.rs:299
        out.push_back(TextChar{ch, r});
    }
    return out;
}
