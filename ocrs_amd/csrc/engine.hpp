// OcrEngine (ocrs/src/lib.rs:111-301) on MI355X: the grey page stays in HBM
// between prepare_input / detect_words / recognize_text.
#pragma once
#include <string>
#include <vector>

#include <memory>

#include "coalesce.hpp"
#include "common.hpp"
#include "geometry.hpp"
#include "model.hpp"

struct ocrs_page {  // OcrInput (lib.rs:125-128)
    ocrs::DevBuf grey;  // [h, w] fp32 in [-0.5, 0.5]; on the device of the engine that prepared it
    int h = 0, w = 0;
    // the host pixels an engine GROUP prepared this page from (group.cpp: key of the replay mode's recorded results); else null
    const void* source = nullptr;
    int device() const { return grey.device(); }
};

namespace ocrs {

struct CtcStep { uint32_t label, pos; };

// ctc_beam.cpp — rten decode_beam (recognition.rs:512-514)
std::vector<CtcStep> ctc_beam_search(const float* logp, int T, int C, int row_stride, uint32_t width);
// the same function written as the algorithm is usually stated (trie + candidate map); tests compare the two
std::vector<CtcStep> ctc_beam_search_reference(const float* logp, int T, int C, int row_stride, uint32_t width);

struct TextChar {  // text_items.rs:48-54
    uint32_t ch;
    geom::Rect rect;
};

struct RecLine {  // TextRecLine (recognition.rs:80-89) + owning page
    size_t page = 0;
    size_t index = 0;                    // line index in the caller's order
    std::vector<geom::PointI> polygon;   // line_polygon (recognition.rs:29-55)
    geom::Rect bounds{0, 0, 0, 0};       // Polygon::bounding_rect
    uint32_t resized_width = 0;
    uint32_t group_width = 0;
};

// One caller's detection / recognition request while it waits in the engine's coalescer (coalesce.hpp).
struct DetRequest : CoalescedBase {
    const ocrs_page* const* pages = nullptr;
    size_t n = 0;
    std::vector<std::vector<geom::RotatedRect>>* rects = nullptr;
};
struct RecRequest : CoalescedBase {
    const ocrs_page* const* pages = nullptr;
    size_t n_pages = 0;
    const std::vector<std::vector<std::vector<geom::RotatedRect>>>* lines_per_page = nullptr;
    std::vector<std::vector<CtcStep>>* steps = nullptr;
    std::vector<RecLine>* rec_lines = nullptr;
    std::vector<uint32_t>* ctc_len = nullptr;
};

}  // namespace ocrs

struct ocrs_engine {
    int device = 0;   // the HIP device of its models (ocrs_engine_new); every call binds the calling thread to it
    const ocrs::ModelBase* detection = nullptr;
    const ocrs::ModelBase* recognition = nullptr;
    bool debug = false;
    ocrs_decode_method decode_method = OCRS_DECODE_GREEDY;
    uint32_t beam_width = 100;
    std::u32string alphabet;
    bool has_excluded = false;
    std::vector<uint8_t> excluded;  // [alphabet_len + 1] flags by label
    ocrs::DevBuf d_excluded;
    // TextDetectorParams::default() (detection.rs:25-37)
    float min_area = 100.0f;
    float text_threshold = 0.2f;
    mutable ocrs::StageTimers timers;
    // the engine's copy of the tuning options (common.hpp): the process defaults at creation + ocrs_engine_params +
    // ocrs_engine_set_option; installed for the calling thread by every entry point (abi_util.hpp guarded_engine)
    ocrs::Tuning tuning{};
    // numerics != exact: the device context this engine is counted in (DeviceContext::add_relaxed_engine) — remembered, because
    // `device` may be reassigned after creation (a group member without weights) and the count must come off where it went on
    ocrs::DeviceContext* counted_relaxed = nullptr;
    void count_relaxed(ocrs::DeviceContext& c) { uncount_relaxed(); c.add_relaxed_engine(+1); counted_relaxed = &c; }
    void uncount_relaxed() noexcept {
        if (!counted_relaxed) return;
        try { counted_relaxed->add_relaxed_engine(-1); } catch (...) {}
        counted_relaxed = nullptr;
    }
    ~ocrs_engine() { uncount_relaxed(); }

    ocrs::StageTimers* tm() const { return timers.enabled ? &timers : nullptr; }

    // detection.rs:104-200 over a batch of equally sized pages.  Small requests (fewer pages than half of option
    // "coalesce_pages") that only want rects are merged with concurrent ones (coalesce.hpp); results are those of
    // detect_now on the caller's pages alone.
    void detect(const ocrs_page* const* pages, size_t n, std::vector<std::vector<ocrs::geom::RotatedRect>>* rects,
                float* host_map /* [n,h,w] or null */) const;
    void detect_now(const ocrs_page* const* pages, size_t n, std::vector<std::vector<ocrs::geom::RotatedRect>>* rects,
                    float* host_map) const;

    // recognition.rs:404-540 over the lines of several pages; small requests are merged likewise.
    void recognize(const ocrs_page* const* pages, size_t n_pages,
                   const std::vector<std::vector<std::vector<ocrs::geom::RotatedRect>>>& lines_per_page,
                   std::vector<std::vector<ocrs::CtcStep>>* steps, std::vector<ocrs::RecLine>* rec_lines,
                   std::vector<uint32_t>* ctc_input_len) const;
    void recognize_now(const ocrs_page* const* pages, size_t n_pages,
                       const std::vector<std::vector<std::vector<ocrs::geom::RotatedRect>>>& lines_per_page,
                       std::vector<std::vector<ocrs::CtcStep>>* steps, std::vector<ocrs::RecLine>* rec_lines,
                       std::vector<uint32_t>* ctc_input_len) const;
    void init_coalescers();
    mutable std::unique_ptr<ocrs::Coalescer<ocrs::DetRequest>> det_queue;
    mutable std::unique_ptr<ocrs::Coalescer<ocrs::RecRequest>> rec_queue;
    // one sub-request of `recognize` (within the activation budget); outputs indexed like `lines`
    // logp (optional): per line the model's log-probabilities [T][C] (TextRecognizer::run, recognition.rs:341-360), unmasked
    void recognize_lines(const ocrs_page* const* pages, size_t n_pages, const std::vector<ocrs::RecLine>& lines,
                         std::vector<std::vector<ocrs::CtcStep>>* steps, std::vector<uint32_t>* ctc_input_len,
                         std::vector<std::vector<float>>* logp = nullptr) const;
    // the recognition model's output for the lines of one page, one request, no coalescing (parity / tolerance checks)
    void recognize_logits(const ocrs_page* page, const std::vector<std::vector<ocrs::geom::RotatedRect>>& lines,
                          std::vector<std::vector<float>>* logp, int* classes) const;

    std::vector<ocrs::TextChar> text_line_from_result(const ocrs::RecLine& line, uint32_t ctc_input_len,
                                                      const std::vector<ocrs::CtcStep>& steps) const;

    uint32_t rec_input_height() const;
    ocrs::RecLine make_rec_line(const std::vector<ocrs::geom::RotatedRect>& words, size_t page, size_t index) const;
};
