// Layout analysis: words -> lines in reading order.  Host side by design
// (O(n^2) on ~600 rects, branchy, sub-millisecond) — mirrors
// ocrs/src/layout_analysis.rs:19-233 and layout_analysis/empty_rects.rs:47-229.
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <functional>
#include <memory>

#include "geometry.hpp"

namespace ocrs {
using namespace geom;

namespace {

// std::collections::BinaryHeap<Partition> with Rust's exact sift order, so that
// partitions with equal scores pop in the same order as in the reference
// (empty_rects.rs:20-24 compares scores only; ties are decided by heap mechanics).
//
// A partition's obstacle list (empty_rects.rs:118-122) is a pure function of its
// boundary and its parent's list, and only the score takes part in the ordering,
// so the list is materialised lazily when the partition is popped: most pushed
// partitions are never popped once the 80 separators are found.
//
// The search pops ~48 000 partitions for a 700-word page, so its inner loops are laid out for the host's
// caches (21 -> 14 ms per 700-word page on the build host in round 2, 14 -> 8.2 in round 3; results identical — the sequence of
// pushes and pops is unchanged):
// heap entries are 8 bytes (score, id) with a branch-free child choice; obstacle lists are indices
// appended to one arena by a branch-free filter over structure-of-arrays obstacle coordinates; payloads are
// written once and read once.
struct Partition {
    Rect boundary;
    uint32_t obs_off, obs_len;  // the PARENT's obstacle list: a slice of the index arena
    Partition(const Rect& r, uint32_t off, uint32_t len) : boundary(r), obs_off(off), obs_len(len) {}
    Rect rect() const { return boundary; }
};
// The same in 16 bytes, for pages whose coordinates fit 16 bits and that have fewer than 65 536 words (every page the
// engine accepts up to 32 767 pixels a side): payload store 1.1 MB instead of 1.7, index arena halved.
struct PartitionC {
    int16_t top, left, bottom, right;
    uint32_t obs_off;
    uint16_t obs_len, pad;
    PartitionC(const Rect& r, uint32_t off, uint32_t len)
        : top((int16_t)r.top), left((int16_t)r.left), bottom((int16_t)r.bottom), right((int16_t)r.right), obs_off(off),
          obs_len((uint16_t)len), pad(0) {}
    Rect rect() const { return Rect{top, left, bottom, right}; }
};
static_assert(sizeof(PartitionC) == 16, "compact payload");

struct HeapEntry {
    float score;
    uint32_t id;
};

class RustBinaryHeap {
  public:
    // Entry i lives at slot i + 1 of a 64-byte aligned buffer: the eight great-grandchildren of any node (entries
    // 8 p + 7 .. 8 p + 14) are then exactly one cache line, which sift_down_to_bottom prefetches two levels ahead.
    explicit RustBinaryHeap(size_t reserve) { grow(std::max<size_t>(reserve, 64)); }
    ~RustBinaryHeap() { std::free(base_); }
    RustBinaryHeap(const RustBinaryHeap&) = delete;
    RustBinaryHeap& operator=(const RustBinaryHeap&) = delete;
    void push(HeapEntry e) {
        if (size_ == cap_) grow(cap_ * 2);
        d_[size_++] = e;
        sift_up(0, size_ - 1);
    }
    bool empty() const { return size_ == 0; }
    uint32_t top_id() const { return d_[0].id; }   // the entry the next pop returns unless a higher score is pushed first
    size_t size() const { return size_; }
    uint32_t id_at(size_t pos) const { return d_[pos].id; }   // pos < size(): entries 1 and 2 are the candidates for the pop after next
    bool pop(HeapEntry& out) {
        if (size_ == 0) return false;
        HeapEntry item = d_[--size_];
        if (size_ != 0) {
            std::swap(item, d_[0]);
            sift_down_to_bottom(0);
        }
        out = item;
        return true;
    }

  private:
    void grow(size_t cap) {
        const size_t bytes = ((cap + 1) * sizeof(HeapEntry) + 63) / 64 * 64;
        HeapEntry* nb = static_cast<HeapEntry*>(std::aligned_alloc(64, bytes));
        if (!nb) throw std::bad_alloc();
        if (base_) {
            std::memcpy(nb + 1, d_, size_ * sizeof(HeapEntry));
            std::free(base_);
        }
        base_ = nb;
        d_ = nb + 1;
        cap_ = cap;
    }
    // f32::total_cmp on scores that are never NaN here
    size_t sift_up(size_t start, size_t pos) {
        HeapEntry* d = d_;
        const HeapEntry elt = d[pos];
        while (pos > start) {
            const size_t parent = (pos - 1) / 2;
            if (elt.score <= d[parent].score) break;
            d[pos] = d[parent];
            pos = parent;
        }
        d[pos] = elt;
        return pos;
    }
    void sift_down_to_bottom(size_t pos) {
        HeapEntry* d = d_;
        const size_t end = size_;
        const size_t start = pos;
        const HeapEntry elt = d[pos];
        size_t child = 2 * pos + 1;
        while (end >= 2 && child <= end - 2) {
            // on its way to L1 two levels before the walk needs it (the heap is ~0.5 MB; without this every level
            // below the ninth is an L2 round trip)
            if (8 * pos + 7 < end) __builtin_prefetch(d + 8 * pos + 7);
            child += (size_t)(d[child].score <= d[child + 1].score);  // the greater child; the right one on a tie
            d[pos] = d[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) {
            d[pos] = d[child];
            pos = child;
        }
        d[pos] = elt;
        sift_up(start, pos);
    }
    HeapEntry* base_ = nullptr;
    HeapEntry* d_ = nullptr;
    size_t size_ = 0, cap_ = 0;
};

// empty_rects.rs:80-138 + FilterRectIter (:184-221) + take(n)
template <class Part, class Idx, class Score>
std::vector<Rect> max_empty_rects_search(const std::vector<Rect>& obstacles, Rect boundary, Score&& score,
                                         uint32_t min_width, uint32_t min_height, float iou_threshold, size_t take) {
    const size_t n_obs = obstacles.size();
    // obstacle coordinates as one 16-byte record each (left, top, -right, -bottom): the filter below tests a record
    // against a partition with ONE vector compare — (left, top, -right, -bottom) < (b.right, b.bottom, -b.left, -b.top) in
    // all four lanes <=> the rects intersect — and the 11 KB of records stay in L1
    struct alignas(16) Quad { int32_t v[4]; };
    std::vector<Quad> oq(n_obs);
    for (size_t i = 0; i < n_obs; i++)
        oq[i] = Quad{{obstacles[i].left, obstacles[i].top, -obstacles[i].right, -obstacles[i].bottom}};
    // index arena: every materialised obstacle list is appended here; partitions refer to slices.
    // Grown geometrically by hand so that the filter loop can store without a capacity check.
    std::vector<Idx> arena(std::max<size_t>(n_obs * 96, 1024));
    size_t arena_n = n_obs;
    for (size_t i = 0; i < n_obs; i++) arena[i] = (Idx)i;
    RustBinaryHeap queue(n_obs * 128 + 64);
    std::vector<Part> store;  // payloads; ids stay unique
    store.reserve(n_obs * 128 + 64);
    auto push = [&](const Rect& r, uint32_t off, uint32_t len) {
        store.emplace_back(r, off, len);
        queue.push(HeapEntry{score(r), (uint32_t)(store.size() - 1)});
    };
    const bool have_root = !boundary.is_empty();
    if (have_root) push(boundary, 0, (uint32_t)n_obs);
    std::vector<Rect> found;
    HeapEntry he;
    while (found.size() < take && queue.pop(he)) {
        const Part part = store[he.id];
        const Rect b = part.rect();
        // the payload store (~1.7 MB) and the arena (~2 MB) are visited in score order, i.e. at random: start
        // fetching the likely next partition's payload now and its obstacle slice at the end of this iteration
        const bool have_top = !queue.empty();
        const uint32_t top = have_top ? queue.top_id() : 0;
        if (have_top) __builtin_prefetch(&store[top]);
        if (queue.size() > 2) {   // one of these two is the top after the next pop: its payload is then already here
            __builtin_prefetch(&store[queue.id_at(1)]);
            __builtin_prefetch(&store[queue.id_at(2)]);
        }
        if (have_top) {           // ... so the likely next partition's obstacle slice can be requested a whole iteration ahead
            const Idx* nx = arena.data() + store[top].obs_off;
            __builtin_prefetch(nx);
            __builtin_prefetch(nx + 16);
        }
        // materialise this partition's obstacle list (the root keeps every obstacle, as in the reference)
        uint32_t my_off, my_len;
        if (he.id == 0 && have_root) {
            my_off = 0;
            my_len = part.obs_len;
        } else {
            if (arena_n + part.obs_len > arena.size()) arena.resize(std::max(arena.size() * 2, arena_n + part.obs_len));
            const Idx* src = arena.data() + part.obs_off;
            Idx* dst = arena.data() + arena_n;
            size_t k = 0;
#if defined(__SSE2__)
            const __m128i bound = _mm_set_epi32(-b.top, -b.left, b.bottom, b.right);
            for (uint32_t q = 0; q < part.obs_len; q++) {   // branch-free filter: store always, advance if it intersects
                const Idx idx = src[q];
                dst[k] = idx;
                const __m128i lt = _mm_cmplt_epi32(_mm_load_si128(reinterpret_cast<const __m128i*>(oq[idx].v)), bound);
                k += (size_t)(_mm_movemask_ps(_mm_castsi128_ps(lt)) == 0xF);
            }
#else
            const int32_t bound[4] = {b.right, b.bottom, -b.left, -b.top};   // the same four strict comparisons, scalar
            for (uint32_t q = 0; q < part.obs_len; q++) {
                const Idx idx = src[q];
                dst[k] = idx;
                const int32_t* v = oq[idx].v;
                k += (size_t)((v[0] < bound[0]) & (v[1] < bound[1]) & (v[2] < bound[2]) & (v[3] < bound[3]));
            }
#endif
            my_off = (uint32_t)arena_n;
            my_len = (uint32_t)k;
            arena_n += k;
        }
        if (my_len == 0) {
            bool overlaps = false;
            for (const Rect& f : found) {
                // disjoint rects have iou == 0 < threshold: skip the float division
                if (!(f.left < b.right && f.right > b.left && f.top < b.bottom && f.bottom > b.top) && iou_threshold > 0.0f)
                    continue;
                if (f.iou(b) >= iou_threshold) { overlaps = true; break; }
            }
            if (!overlaps) found.push_back(b);
            continue;
        }
        const Rect pivot = obstacles[arena[my_off + my_len / 2]];
        const Rect right_rect = Rect::from_tlbr(b.top, pivot.right, b.bottom, b.right);
        const Rect left_rect = Rect::from_tlbr(b.top, b.left, b.bottom, pivot.left);
        const Rect top_rect = Rect::from_tlbr(b.top, b.left, pivot.top, b.right);
        const Rect bottom_rect = Rect::from_tlbr(pivot.bottom, b.left, b.bottom, b.right);
        const Rect subs[4] = {top_rect, left_rect, bottom_rect, right_rect};
        for (const Rect& sr : subs) {
            if ((uint32_t)std::max(sr.width(), 0) < min_width || (uint32_t)std::max(sr.height(), 0) < min_height ||
                sr.is_empty())
                continue;
            push(sr, my_off, my_len);
        }
    }
    return found;
}

template <class Score>
std::vector<Rect> max_empty_rects_filtered(const std::vector<Rect>& obstacles_in, Rect boundary, Score&& score,
                                           uint32_t min_width, uint32_t min_height, float iou_threshold, size_t take) {
    std::vector<Rect> obstacles = obstacles_in;
    std::stable_sort(obstacles.begin(), obstacles.end(), [](const Rect& a, const Rect& b) {
        PointI ca = a.center(), cb = b.center();
        return ca.x != cb.x ? ca.x < cb.x : ca.y < cb.y;
    });
    auto fits16 = [](const Rect& r) {
        return r.top >= INT16_MIN && r.left >= INT16_MIN && r.bottom <= INT16_MAX && r.right <= INT16_MAX &&
               r.top <= INT16_MAX && r.left <= INT16_MAX && r.bottom >= INT16_MIN && r.right >= INT16_MIN;
    };
    bool compact = obstacles.size() < 65536 && fits16(boundary);
    for (size_t i = 0; compact && i < obstacles.size(); i++) compact = fits16(obstacles[i]);
    // (every sub-partition's coordinates are coordinates of the boundary or of an obstacle)
    if (compact)
        return max_empty_rects_search<PartitionC, uint16_t>(obstacles, boundary, score, min_width, min_height, iou_threshold, take);
    return max_empty_rects_search<Partition, uint32_t>(obstacles, boundary, score, min_width, min_height, iou_threshold, take);
}

struct WordInfo {  // cached per-word quantities for group_into_lines
    RotatedRect rect;
    int32_t left_i;
    float ledge_cx, redge_cx;
    float ledge_y0, ledge_y1, redge_y0, redge_y1;  // y-extent of the downwards() edges
    int32_t cx_i;
};

// layout_analysis.rs:19-71
std::vector<std::vector<RotatedRect>> group_into_lines(const std::vector<RotatedRect>& rects,
                                                       const std::vector<LineF>& separators) {
    std::vector<WordInfo> ws;
    ws.reserve(rects.size());
    for (const RotatedRect& r : rects) {
        WordInfo w;
        w.rect = r;
        w.left_i = as_i32(r.bounding_rect().left);
        const LineF le = leftmost_edge(r).downwards(), re = rightmost_edge(r).downwards();
        w.ledge_cx = le.center().x;
        w.redge_cx = re.center().x;
        w.ledge_y0 = le.start.y; w.ledge_y1 = le.end.y;
        w.redge_y0 = re.start.y; w.redge_y1 = re.end.y;
        w.cx_i = as_i32(r.cx);
        ws.push_back(w);
    }
    std::stable_sort(ws.begin(), ws.end(), [](const WordInfo& a, const WordInfo& b) { return a.left_i < b.left_i; });
    // Pruning of the candidate scan below (the scan is the reference's: every remaining word, in sorted order; what is
    // skipped provably fails or cannot win): a candidate needs cx > last.cx, and cx < left_i + 1 + max half-width, so the
    // scan may start at the first word with left_i > last.cx - 1 - max half-width; and cx_i >= left_i, so once a best
    // candidate is known no word with left_i > its key can beat it (ties keep the first minimum anyway).
    std::vector<int32_t> lefts(ws.size());
    float max_half = 0.0f;
    for (size_t i = 0; i < ws.size(); i++) {
        lefts[i] = ws[i].left_i;
        max_half = std::max(max_half, ws[i].rect.cx - ws[i].rect.bounding_rect().left);
    }
    const bool prune = std::isfinite(max_half);

    const float overlap_threshold = 5.0f;
    const float max_h_overlap = 5.0f;
    std::vector<char> used(ws.size(), 0);
    std::vector<std::vector<RotatedRect>> lines;
    size_t first_unused = 0;
    while (true) {
        while (first_unused < ws.size() && used[first_unused]) first_unused++;
        if (first_unused >= ws.size()) break;
        std::vector<RotatedRect> line;
        size_t last_i = first_unused;
        used[last_i] = 1;
        line.push_back(ws[last_i].rect);
        while (true) {
            const WordInfo& last = ws[last_i];
            long best = -1;
            int32_t best_key = 0;
            size_t scan_from = first_unused;
            if (prune) {
                const double lo = std::floor((double)last.rect.cx - 2.0 - (double)max_half);
                if (lo > (double)INT32_MIN) {
                    const int32_t lo_i = lo >= (double)INT32_MAX ? INT32_MAX : (int32_t)lo;
                    scan_from = std::max(scan_from, (size_t)(std::lower_bound(lefts.begin(), lefts.end(), lo_i) - lefts.begin()));
                }
            }
            for (size_t i = scan_from; i < ws.size(); i++) {  // remaining rects keep their sorted order
                if (prune && best >= 0 && lefts[i] > best_key) break;
                if (used[i]) continue;
                const WordInfo& w = ws[i];
                if (!(w.rect.cx > last.rect.cx)) continue;
                if (!(w.ledge_cx - last.redge_cx >= -max_h_overlap)) continue;
                // last_edge.vertical_overlap(edge) (Line::vertical_overlap on the downwards() edges)
                if (!(overlap(last.redge_y0, last.redge_y1, w.ledge_y0, w.ledge_y1) >= overlap_threshold)) continue;
                if (!separators.empty()) {
                    LineF a_to_b{last.rect.center(), w.rect.center()};
                    bool sep = false;
                    for (const LineF& s : separators)
                        if (a_to_b.intersects(s)) { sep = true; break; }
                    if (sep) continue;
                }
                if (best < 0 || w.cx_i < best_key) {  // min_by_key keeps the first minimum
                    best = (long)i;
                    best_key = w.cx_i;
                }
            }
            if (best < 0) break;
            used[best] = 1;
            line.push_back(ws[best].rect);
            last_i = (size_t)best;
        }
        lines.push_back(std::move(line));
    }
    return lines;
}

LineF midpoint_line(const std::vector<RotatedRect>& words) {
    return LineF{words.front().bounding_rect().left_edge().center(), words.back().bounding_rect().right_edge().center()};
}

}  // namespace

// layout_analysis.rs:83-155
std::vector<Rect> find_block_separators(const std::vector<RotatedRect>& words) {
    if (words.empty()) return {};
    RectF br = words[0].bounding_rect();
    for (size_t i = 1; i < words.size(); i++) br = br.unite(words[i].bounding_rect());
    const Rect page_rect = br.integral_bounding_rect();

    auto lines = group_into_lines(words, {});
    std::stable_sort(lines.begin(), lines.end(), [](const auto& a, const auto& b) {
        return (int32_t)rround(a.front().bounding_rect().top) < (int32_t)rround(b.front().bounding_rect().top);
    });

    std::vector<int32_t> all_spacings;
    for (const auto& line : lines) {
        if (line.size() > 1) {
            std::vector<int32_t> sp;
            for (size_t i = 0; i + 1 < line.size(); i++) {
                float v = line[i + 1].bounding_rect().left - line[i].bounding_rect().right;
                v = v > 0.0f ? v : 0.0f;
                sp.push_back((int32_t)rround(v));
            }
            std::sort(sp.begin(), sp.end());
            all_spacings.insert(all_spacings.end(), sp.begin(), sp.end());
        }
    }
    std::sort(all_spacings.begin(), all_spacings.end());
    const int32_t median_word_spacing = all_spacings.empty() ? 10 : all_spacings[all_spacings.size() / 2];
    const int32_t median_height = (int32_t)rround(words[words.size() / 2].h);

    // Shafait/Keysers/Breuel score favouring tall rectangles (layout_analysis.rs:127-135).
    // `aspect_ratio.log2().abs()` compared with 3 and 5: for a quotient of two integers below 2^17 the
    // f32 log2 lands on the same side of 3 / 5 as the aspect ratio does of 8 / 32 (and 1/8, 1/32), so the
    // logarithm is only evaluated where its value is used as the weight.
    auto score = [](const Rect& r) -> float {
        const float aspect = (float)r.height() / (float)r.width();
        float wgt;
        if (aspect > 0.125f && aspect < 8.0f) wgt = 0.5f;
        else if (aspect > 0.03125f && aspect < 32.0f) wgt = 1.5f;
        else wgt = std::fabs((float)std::log2((double)aspect));
        return std::sqrt((float)r.area() * wgt);
    };

    std::vector<Rect> boxes;
    boxes.reserve(words.size());
    for (const RotatedRect& w : words) boxes.push_back(w.bounding_rect().integral_bounding_rect());
    const uint32_t min_width = (uint32_t)(median_word_spacing * 3);
    const uint32_t min_height = (uint32_t)(3 * std::max(median_height, 0));
    return max_empty_rects_filtered(boxes, page_rect, score, min_width, min_height, 0.5f, 80);
}

// layout_analysis.rs:158-233
std::vector<std::vector<RotatedRect>> find_text_lines(const std::vector<RotatedRect>& words) {
    const std::vector<Rect> separators = find_block_separators(words);
    std::vector<LineF> vertical, horizontal;
    for (const Rect& r : separators) {
        PointI c = r.center();
        vertical.push_back(LineF{PointF{(float)c.x, (float)r.top}, PointF{(float)c.x, (float)r.bottom}});
        horizontal.push_back(LineF{PointF{(float)r.left, (float)c.y}, PointF{(float)r.right, (float)c.y}});
    }
    auto lines = group_into_lines(words, vertical);
    std::stable_sort(lines.begin(), lines.end(), [](const auto& a, const auto& b) {
        return as_i32(midpoint_line(a).center().y) < as_i32(midpoint_line(b).center().y);
    });

    auto is_separated_by = [&](const LineF& a, const LineF& b) {
        LineF a_to_b{a.center(), b.center()};
        for (const LineF& s : horizontal)
            if (s.intersects(a_to_b)) return true;
        return false;
    };

    std::vector<std::vector<RotatedRect>> out;
    std::vector<char> used(lines.size(), 0);
    for (size_t seed = 0; seed < lines.size(); seed++) {
        if (used[seed]) continue;
        used[seed] = 1;
        out.push_back(lines[seed]);
        LineF prev = midpoint_line(lines[seed]);
        for (size_t i = seed + 1; i < lines.size(); i++) {
            if (used[i]) continue;
            LineF cand = midpoint_line(lines[i]);
            if (prev.horizontal_overlap(cand) > 0.0f && !is_separated_by(prev, cand)) {
                used[i] = 1;
                out.push_back(lines[i]);
                prev = cand;
            }
        }
    }
    return out;
}

}  // namespace ocrs
