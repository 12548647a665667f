// Internals shared by the translation units that export C entry points (abi.cpp, group.cpp): exception -> status
// conversion, device binding, marshalling helpers.  Not part of the public header.
#pragma once
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "engine.hpp"

namespace ocrs {
namespace abi {

template <class F>
ocrs_status guarded(F&& f) {   // host-only entry points: no device is touched
    try {
        f();
        set_last_error("");
        return OCRS_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.status;
    } catch (const std::bad_alloc&) {
        set_last_error("out of host memory");
        return OCRS_ERR_DEVICE;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return OCRS_ERR_RUN_FAILED;
    }
}

// Entry points that do device work: the calling thread is bound to the handle's device for the call
// (device < 0: the process default, ocrs_set_device).
template <class F>
ocrs_status guarded_on(int device, F&& f) {
    return guarded([&] {
        DeviceScope bind(device);
        f();
    });
}

// Entry points of an engine: bound to the engine's device, with the engine's own options installed for the call.
template <class F>
ocrs_status guarded_engine(const ocrs_engine* e, F&& f) {
    return guarded([&] {
        DeviceScope bind(e ? e->device : -1);
        TuningScope tune(e ? &e->tuning : nullptr);
        f();
    });
}

template <class T>
T* dup_buffer(const std::vector<T>& v) {
    T* p = static_cast<T*>(malloc(std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!p) throw std::bad_alloc();
    if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

void check_pages_on(const ocrs_engine* e, const ocrs_page* const* pages, size_t n);
std::u32string decode_utf8(const char* s);
void append_utf8(std::string& s, uint32_t c);
std::vector<std::vector<geom::RotatedRect>> unpack_lines(const float* rects, const size_t* offsets, size_t first, size_t last);
// OcrEngine::prepare_input on pixels that are already on the bound device (launch only, no sync)
ocrs_page* make_page(const void* d_pixels, ocrs_pixel_type type, ocrs_dim_order order, int height, int width, int channels,
                     hipStream_t st, StageTimers* T);
void check_image_args(const void* pixels, int height, int width, int channels);
// OcrEngine::new (lib.rs:132-180)
std::unique_ptr<ocrs_engine> make_engine(const ocrs_engine_params& params);
// flatten per-line characters into the ABI's (chars, char_offsets) form
void flatten_chars(const ocrs_engine* e, const std::vector<RecLine>& rl, const std::vector<uint32_t>& ctc_len,
                   const std::vector<std::vector<CtcStep>>& steps, std::vector<ocrs_text_char>* flat, std::vector<size_t>* offs);

}  // namespace abi
}  // namespace ocrs
