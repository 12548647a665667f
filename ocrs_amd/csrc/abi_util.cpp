// Helpers shared by abi.cpp and group.cpp (abi_util.hpp).
#include "abi_util.hpp"

#include "kernels.hpp"

namespace ocrs {
namespace abi {

using namespace ocrs::geom;

void check_pages_on(const ocrs_engine* e, const ocrs_page* const* pages, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (!pages[i]) fail(OCRS_ERR_INVALID_ARGUMENT, "null page");
        if (pages[i]->device() != e->device)
            fail(OCRS_ERR_INVALID_ARGUMENT, "page %zu lives on device %d, the engine on device %d", i, pages[i]->device(),
                 e->device);
    }
}

std::u32string decode_utf8(const char* s) {
    std::u32string out;
    const unsigned char* p = reinterpret_cast<const unsigned char*>(s);
    while (*p) {
        uint32_t c = *p++;
        int extra = c >= 0xF0 ? 3 : c >= 0xE0 ? 2 : c >= 0xC0 ? 1 : 0;
        if (extra) c &= (0x3F >> extra);
        while (extra-- > 0 && *p) c = (c << 6) | (*p++ & 0x3F);
        out.push_back((char32_t)c);
    }
    return out;
}

void append_utf8(std::string& s, uint32_t c) {
    if (c < 0x80) s.push_back((char)c);
    else if (c < 0x800) { s.push_back((char)(0xC0 | (c >> 6))); s.push_back((char)(0x80 | (c & 0x3F))); }
    else if (c < 0x10000) {
        s.push_back((char)(0xE0 | (c >> 12))); s.push_back((char)(0x80 | ((c >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (c & 0x3F)));
    } else {
        s.push_back((char)(0xF0 | (c >> 18))); s.push_back((char)(0x80 | ((c >> 12) & 0x3F)));
        s.push_back((char)(0x80 | ((c >> 6) & 0x3F))); s.push_back((char)(0x80 | (c & 0x3F)));
    }
}

// lib.rs:34 with the EUR sign restored (lib.rs:33)
static const char kDefaultAlphabet[] =
    " 0123456789!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~\xE2\x82\xAC"
    "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz";

std::vector<std::vector<RotatedRect>> unpack_lines(const float* rects, const size_t* offsets, size_t first, size_t last) {
    std::vector<std::vector<RotatedRect>> lines;
    for (size_t i = first; i < last; i++) {
        std::vector<RotatedRect> words;
        for (size_t k = offsets[i]; k < offsets[i + 1]; k++) words.push_back(RotatedRect::from_array(rects + 6 * k));
        lines.push_back(std::move(words));
    }
    return lines;
}

ocrs_page* make_page(const void* d_pixels, ocrs_pixel_type type, ocrs_dim_order order, int height, int width, int channels,
                     hipStream_t st, StageTimers* T) {
    auto page = std::make_unique<ocrs_page>();
    page->h = height;
    page->w = width;
    page->grey = DevBuf((size_t)height * width * sizeof(float));
    {
        StageScope sc(T, ST_PREPARE, st);
        k::prepare_image(d_pixels, type == OCRS_U8, order == OCRS_HWC, height, width, channels, page->grey.as<float>(), st);
    }
    OCRS_HIP(hipGetLastError());
    return page.release();
}

void check_image_args(const void* pixels, int height, int width, int channels) {
    if (!pixels) fail(OCRS_ERR_INVALID_ARGUMENT, "pixels is null");
    // ImageSource::from_tensor (preprocess.rs:116-122)
    if (!(channels == 1 || channels == 3 || channels == 4)) fail(OCRS_ERR_IMAGE_SOURCE, "channel count is not 1, 3 or 4");
    if (height <= 0 || width <= 0) fail(OCRS_ERR_INVALID_ARGUMENT, "image has no pixels");
}

std::unique_ptr<ocrs_engine> make_engine(const ocrs_engine_params& params) {
    auto e = std::make_unique<ocrs_engine>();
    e->detection = params.detection_model ? params.detection_model->impl.get() : nullptr;
    e->recognition = params.recognition_model ? params.recognition_model->impl.get() : nullptr;
    // the engine lives where its models' weights are; callback models (no device) follow the other model or,
    // failing that, the process default
    const int dd = e->detection ? e->detection->device : -1, rd = e->recognition ? e->recognition->device : -1;
    if (dd >= 0 && rd >= 0 && dd != rd)
        fail(OCRS_ERR_INVALID_ARGUMENT, "detection model is on device %d, recognition model on device %d", dd, rd);
    e->device = dd >= 0 ? dd : rd >= 0 ? rd : default_device();
    e->debug = params.debug != 0;
    e->tuning = default_tuning();
    if (params.numerics != OCRS_NUMERICS_EXACT && params.numerics != OCRS_NUMERICS_RELAXED && params.numerics != OCRS_NUMERICS_REDUCED)
        fail(OCRS_ERR_INVALID_ARGUMENT, "unknown numerics mode %d", (int)params.numerics);
    e->tuning.v[OPT_NUMERICS] = (long)params.numerics;
    if (params.numerics != OCRS_NUMERICS_EXACT) e->count_relaxed(device_context(e->device));   // waits for the device's requests in flight
    // coalescing fields: 0 = default, negative = off / zero
    if (params.coalesce) e->tuning.v[OPT_COALESCE] = params.coalesce < 0 ? 0 : params.coalesce;
    if (params.coalesce_pages) e->tuning.v[OPT_COALESCE_PAGES] = params.coalesce_pages < 0 ? 1 : params.coalesce_pages;
    if (params.coalesce_window_us) e->tuning.v[OPT_COALESCE_WINDOW_US] = params.coalesce_window_us < 0 ? 0 : params.coalesce_window_us;
    if (params.layout_threads > 0) e->tuning.v[OPT_LAYOUT_THREADS] = params.layout_threads;
    if (params.rec_max_pixels > 0) e->tuning.v[OPT_REC_MAX_PIXELS] = (long)params.rec_max_pixels;
    e->decode_method = params.decode_method;
    e->beam_width = params.beam_width ? params.beam_width : 100;
    e->alphabet = decode_utf8(params.alphabet ? params.alphabet : kDefaultAlphabet);
    if (params.allowed_chars) {  // lib.rs:153-170
        const std::u32string allowed = decode_utf8(params.allowed_chars);
        e->excluded.assign(e->alphabet.size() + 1, 0);
        for (size_t i = 0; i < e->alphabet.size(); i++)
            if (allowed.find(e->alphabet[i]) == std::u32string::npos) e->excluded[i + 1] = 1;
        e->has_excluded = true;
        DeviceScope bind(e->device);
        e->d_excluded = DevBuf(e->excluded.size());
        OCRS_HIP(hipMemcpy(e->d_excluded.p, e->excluded.data(), e->excluded.size(), hipMemcpyHostToDevice));
    }
    e->init_coalescers();
    return e;
}

void flatten_chars(const ocrs_engine* e, const std::vector<RecLine>& rl, const std::vector<uint32_t>& ctc_len,
                   const std::vector<std::vector<CtcStep>>& steps, std::vector<ocrs_text_char>* flat, std::vector<size_t>* offs) {
    flat->clear();
    offs->assign(1, 0);
    for (size_t i = 0; i < rl.size(); i++) {
        for (const TextChar& c : e->text_line_from_result(rl[i], ctc_len[i], steps[i]))
            flat->push_back(ocrs_text_char{c.ch, c.rect.top, c.rect.left, c.rect.bottom, c.rect.right});
        offs->push_back(flat->size());
    }
}

}  // namespace abi
}  // namespace ocrs
