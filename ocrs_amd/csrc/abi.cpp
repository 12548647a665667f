// extern "C" surface of libocrs_amd.so (include/ocrs_amd.h).  Every entry point
// converts C++ exceptions into (status, thread-local message); nothing aborts.
#include <cstdio>
#include <fstream>
#include <thread>

#include <atomic>
#include "abi_util.hpp"
#include "coalesce_selftest.hpp"
#include "engine.hpp"
#include "host_pool.hpp"
#include "jpeg.hpp"
#include "kernels.hpp"

using namespace ocrs;
using namespace ocrs::geom;

using namespace ocrs::abi;

extern "C" {

const char* ocrs_last_error(void) { return last_error().c_str(); }

void ocrs_buffer_free(void* p) { free(p); }

ocrs_status ocrs_device_count(int* n) {
    return guarded([&] {
        int c = 0;
        hipError_t e = hipGetDeviceCount(&c);
        if (e != hipSuccess) c = 0;
        *n = c;
    });
}

ocrs_status ocrs_set_device(int device) {
    return guarded([&] { select_device(device); });
}

ocrs_status ocrs_get_device(int* device) {
    return guarded([&] {
        if (!device) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        *device = default_device();
    });
}

ocrs_status ocrs_gru_tile_plan(const int32_t* lengths_desc, size_t n_lines, int hidden, int32_t* n_clusters, int32_t* waves,
                               int16_t* tiles) {
    return guarded([&] {
        if (!lengths_desc || !n_clusters || !waves || !tiles || n_lines == 0 || n_lines > (size_t)1 << 20)
            fail(OCRS_ERR_INVALID_ARGUMENT, "bad argument");
        for (size_t i = 0; i < n_lines; i++)
            if (lengths_desc[i] < 1 || (i > 0 && lengths_desc[i] > lengths_desc[i - 1]))
                fail(OCRS_ERR_INVALID_ARGUMENT, "lengths must be positive and descending");
        int ncl = 0;
        if (!k::gru_tile_plan(lengths_desc, (int)n_lines, hidden, &ncl, waves, tiles))
            fail(OCRS_ERR_CAPACITY, "no persistent-kernel plan for this shape (the engine then runs the recurrence as one fused launch per time step)");
        *n_clusters = ncl;
    });
}

// Host-only test hook for coalesce.hpp (the engine's queues need a GPU to be driven through the ABI).
ocrs_status ocrs_coalescer_selftest(int n_threads, int requests_per_thread, int max_active, int max_pages, long window_us,
                                    int fail_every, uint64_t out[5]) {
    return guarded([&] {
        if (!out || n_threads < 1 || requests_per_thread < 1 || max_pages < 1) fail(OCRS_ERR_INVALID_ARGUMENT, "bad argument");
        coalescer_selftest(n_threads, requests_per_thread, max_active, max_pages, window_us, fail_every, out);
    });
}

ocrs_status ocrs_ctc_beam_search(const float* logp, int t, int c, uint32_t width, int impl, uint32_t** labels,
                                 uint32_t** positions, size_t* n) {
    return guarded([&] {
        if (!logp || !labels || !positions || !n || t < 0 || c < 1) fail(OCRS_ERR_INVALID_ARGUMENT, "bad argument");
        std::vector<uint32_t> l, p;
        if (impl == 2 && t > 0) {   // the HIP kernel, on this matrix as a one-line packed batch
            DeviceScope bind(-1);
            if (!k::ctc_beam_supported(c, (int)width)) fail(OCRS_ERR_CAPACITY, "beam search on the GPU supports up to 128 classes and width 128");
            Workspace ws;
            std::vector<int32_t> meta(1 + t + 1);
            meta[0] = t;
            for (int i = 0; i <= t; i++) meta[1 + i] = i;   // off[t] = t: one line, row t = time t
            int32_t* d_meta = ws.alloc_n<int32_t>(meta.size());
            float* d_logp = ws.alloc_n<float>((size_t)t * c);
            ws.upload(d_meta, meta.data(), meta.size() * sizeof(int32_t));
            ws.upload(d_logp, logp, (size_t)t * c * sizeof(float));
            const size_t arena = k::ctc_beam_arena_entries(t, (int)width);
            int2* d_nodes = ws.alloc_n<int2>(arena);
            int2* d_posn = ws.alloc_n<int2>(arena);
            uint32_t* d_ol = ws.alloc_n<uint32_t>(t);
            uint32_t* d_op = ws.alloc_n<uint32_t>(t);
            int32_t* d_cnt = ws.alloc_n<int32_t>(1);
            k::ctc_beam_packed(d_logp, d_meta, d_meta + 1, 1, t, c, (int)width, nullptr, d_nodes, d_posn, d_ol, d_op, d_cnt, ws.s());
            std::vector<uint32_t> hl(t), hp(t);
            int32_t cnt = 0;
            ws.download(hl.data(), d_ol, (size_t)t * 4);
            ws.download(hp.data(), d_op, (size_t)t * 4);
            ws.download(&cnt, d_cnt, 4);
            ws.sync();
            OCRS_HIP(hipGetLastError());
            l.assign(hl.begin(), hl.begin() + cnt);
            p.assign(hp.begin(), hp.begin() + cnt);
        } else {
            const std::vector<CtcStep> st = t == 0 ? std::vector<CtcStep>()
                                                  : (impl == 1 ? ctc_beam_search_reference(logp, t, c, c, width) : ctc_beam_search(logp, t, c, c, width));
            l.resize(st.size()); p.resize(st.size());
            for (size_t i = 0; i < st.size(); i++) { l[i] = st[i].label; p[i] = st[i].pos; }
        }
        *labels = dup_buffer(l);
        *positions = dup_buffer(p);
        *n = l.size();
    });
}

uint32_t ocrs_abi_version(void) { return OCRS_ABI_VERSION; }

ocrs_status ocrs_set_option(const char* name, long value) {
    return guarded([&] {
        const int r = set_option(name, value);
        if (r == 1) fail(OCRS_ERR_INVALID_ARGUMENT, "unknown option '%s'", name ? name : "(null)");
        if (r == 2) fail(OCRS_ERR_INVALID_ARGUMENT, "option '%s': value %ld is out of range", name, value);
    });
}

ocrs_status ocrs_engine_set_option(ocrs_engine* e, const char* name, long value) {
    return guarded([&] {
        if (!e) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        const int r = set_option(e->tuning, name, value);
        if (r == 1) fail(OCRS_ERR_INVALID_ARGUMENT, "unknown option '%s'", name ? name : "(null)");
        if (r == 2) fail(OCRS_ERR_INVALID_ARGUMENT, "option '%s': value %ld is out of range", name, value);
    });
}

ocrs_status ocrs_engine_get_option(const ocrs_engine* e, const char* name, long* value) {
    return guarded([&] {
        if (!e || !value) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        if (!get_option(e->tuning, name, value)) fail(OCRS_ERR_INVALID_ARGUMENT, "unknown option '%s'", name ? name : "(null)");
    });
}

ocrs_status ocrs_option_name(int index, const char** name) {
    return guarded([&] {
        if (!name) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        *name = option_name(index);   // NULL past the last option
    });
}

ocrs_status ocrs_device_pool_stats(int device, uint64_t out[12]) {
    return guarded([&] {
        if (!out) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        DeviceContext& c = device_context(device < 0 ? default_device() : device);
        const PoolStats d = c.pool.stats(), h = c.host_pool.stats();
        const uint64_t v[12] = {d.live, d.cached, d.cap, d.peak_live, d.driver_allocs, d.driver_frees,
                                h.live, h.cached, h.cap, h.peak_live, h.driver_allocs, h.driver_frees};
        for (int i = 0; i < 12; i++) out[i] = v[i];
    });
}

ocrs_status ocrs_device_pool_trim(int device) {
    return guarded([&] {
        DeviceContext& c = device_context(device < 0 ? default_device() : device);
        DeviceScope bind(c.device);
        {   // the conv stacks' shared activation arena (kept at its high-water mark between requests): no request may be inside
            std::lock_guard<std::mutex> heavy(c.heavy_phase);
            OCRS_HIP(hipStreamSynchronize(c.heavy_stream()));
            c.heavy_arena.clear();
        }
        c.pool.trim();
    });
}

ocrs_status ocrs_device_set_isolation(int device, ocrs_isolation policy) {
    return guarded([&] {
        if (policy != OCRS_ISOLATION_AUTO && policy != OCRS_ISOLATION_NONE) fail(OCRS_ERR_INVALID_ARGUMENT, "unknown isolation policy %d", (int)policy);
        DeviceContext& c = device_context(device < 0 ? default_device() : device);
        DeviceScope bind(c.device);
        c.set_isolation(policy == OCRS_ISOLATION_NONE ? DeviceContext::ISO_NONE : DeviceContext::ISO_AUTO);
    });
}

ocrs_status ocrs_device_isolation(int device, int out[3]) {
    return guarded([&] {
        if (!out) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        DeviceContext& c = device_context(device < 0 ? default_device() : device);
        DeviceScope bind(c.device);
        out[0] = (int)c.current_mode();
        out[1] = c.relaxed_engine_count();
        out[2] = c.cu_count();
    });
}

ocrs_status ocrs_device_pool_configure(int device, uint64_t device_cached_cap_bytes, uint64_t pinned_cached_cap_bytes) {
    return guarded([&] {
        DeviceContext& c = device_context(device < 0 ? default_device() : device);
        if (device_cached_cap_bytes) c.pool.set_cap(device_cached_cap_bytes);
        if (pinned_cached_cap_bytes) c.host_pool.set_cap(pinned_cached_cap_bytes);
    });
}

// ------------------------------------------------------------------ models
ocrs_status ocrs_model_load_bytes_on_device(const void* data, size_t len, int device, ocrs_model** out) {
    return guarded([&] {
        if (!data || !out) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        auto m = std::make_unique<ocrs_model>();
        m->impl = HipModel::load(data, len, device);   // validates first, binds to the device for the upload only
        *out = m.release();
    });
}

ocrs_status ocrs_model_load_bytes(const void* data, size_t len, ocrs_model** out) {
    return ocrs_model_load_bytes_on_device(data, len, -1, out);
}

ocrs_status ocrs_model_load_file_on_device(const char* path, int device, ocrs_model** out) {
    return guarded([&] {
        if (!path || !out) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        std::ifstream f(path, std::ios::binary);
        if (!f) fail(OCRS_ERR_IO, "cannot open model file %s", path);
        std::vector<char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        auto m = std::make_unique<ocrs_model>();
        m->impl = HipModel::load(buf.data(), buf.size(), device);
        *out = m.release();
    });
}

ocrs_status ocrs_model_load_file(const char* path, ocrs_model** out) { return ocrs_model_load_file_on_device(path, -1, out); }

ocrs_status ocrs_model_device(const ocrs_model* m, int* device) {
    return guarded([&] {
        if (!m || !device) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        *device = m->impl->device;
    });
}

ocrs_status ocrs_model_from_callback(const int64_t input_shape[4], ocrs_model_run_fn run, void* user,
                                     ocrs_model** out) {
    return guarded([&] {
        if (!input_shape || !run || !out) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        auto cb = std::make_unique<CallbackModel>();
        for (int i = 0; i < 4; i++) cb->input_shape[i] = input_shape[i];
        cb->fn = run;
        cb->user = user;
        auto m = std::make_unique<ocrs_model>();
        m->impl = std::move(cb);
        *out = m.release();
    });
}

ocrs_status ocrs_model_input_shape(const ocrs_model* m, int64_t dims[4], uint8_t is_fixed[4]) {
    return guarded([&] {
        if (!m || !dims || !is_fixed) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        for (int i = 0; i < 4; i++) {
            dims[i] = m->impl->input_shape[i];
            is_fixed[i] = dims[i] >= 0;
        }
    });
}

ocrs_status ocrs_model_run(const ocrs_model* m, const float* input, const int64_t in_shape[4],
                           const ocrs_run_options* opts, float** output, int64_t out_shape[4], int* out_ndim) {
    return guarded([&] {
        if (!m || !input || !in_shape || !output || !out_shape || !out_ndim)
            fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        const int64_t n = in_shape[0], c = in_shape[1], h = in_shape[2], w = in_shape[3];
        if (n <= 0 || h <= 0 || w <= 0) fail(OCRS_ERR_RUN_FAILED, "model run failed: empty input");
        if (m->impl->is_callback()) {
            std::vector<float> out;
            static_cast<const CallbackModel*>(m->impl.get())->run(input, in_shape, out, out_shape, out_ndim);
            *output = dup_buffer(out);
            return;
        }
        const auto* hm = static_cast<const HipModel*>(m->impl.get());
        DeviceScope bind(hm->device);
        if (c != 1) fail(OCRS_ERR_RUN_FAILED, "model run failed: expected 1 input channel, got %lld", (long long)c);
        for (int i = 2; i < 4; i++)
            if (hm->input_shape[i] >= 0 && hm->input_shape[i] != in_shape[i])
                fail(OCRS_ERR_RUN_FAILED, "model run failed: input dim %d is %lld, model expects %lld", i,
                     (long long)in_shape[i], (long long)hm->input_shape[i]);
        Workspace ws;
        const size_t cnt = (size_t)n * h * w;
        float* d_in = ws.alloc_n<float>(cnt);
        OCRS_HIP(hipMemcpyAsync(d_in, input, cnt * sizeof(float), hipMemcpyHostToDevice, ws.s()));
        TensorShape os;
        float* d_out = hm->run_device(ws, d_in, (int)n, (int)h, (int)w, &os, nullptr, nullptr, nullptr, true,
                                      opts && opts->timing);
        std::vector<float> host((size_t)os.count());
        ws.download(host.data(), d_out, host.size() * sizeof(float));
        ws.sync();
        if (os.seq) {  // [T, N, C]
            out_shape[0] = os.n; out_shape[1] = os.h; out_shape[2] = os.c; out_shape[3] = 1;
            *out_ndim = 3;
        } else {  // NHWC with C == channels -> report NCHW; C == 1 needs no transpose
            if (os.c != 1) {
                std::vector<float> t(host.size());
                for (int64_t b = 0; b < os.n; b++)
                    for (int64_t y = 0; y < os.h; y++)
                        for (int64_t x = 0; x < os.w; x++)
                            for (int64_t ch = 0; ch < os.c; ch++)
                                t[((b * os.c + ch) * os.h + y) * os.w + x] = host[((b * os.h + y) * os.w + x) * os.c + ch];
                host.swap(t);
            }
            out_shape[0] = os.n; out_shape[1] = os.c; out_shape[2] = os.h; out_shape[3] = os.w;
            *out_ndim = 4;
        }
        *output = dup_buffer(host);
    });
}

ocrs_status ocrs_model_flops(const ocrs_model* m, const int64_t in_shape[4], double* flops) {
    return guarded([&] {
        if (!m || !in_shape || !flops) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        if (m->impl->is_callback()) { *flops = 0; return; }
        *flops = static_cast<const HipModel*>(m->impl.get())->flops((int)in_shape[0], (int)in_shape[2], (int)in_shape[3]);
    });
}

void ocrs_model_free(ocrs_model* m) { delete m; }

// ------------------------------------------------------------------ engine
ocrs_status ocrs_engine_new(const ocrs_engine_params* params, ocrs_engine** out) {
    return guarded([&] {
        if (!params || !out) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        *out = make_engine(*params).release();
    });
}

ocrs_status ocrs_engine_device(const ocrs_engine* e, int* device) {
    return guarded([&] {
        if (!e || !device) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        *device = e->device;
    });
}

void ocrs_engine_free(ocrs_engine* e) { delete e; }

ocrs_status ocrs_image_source_check_bytes(size_t len, uint32_t width, uint32_t height, uint32_t* channels) {
    return guarded([&] {  // preprocess.rs:81-101
        const size_t channel_len = (size_t)width * height;
        if (channel_len == 0) fail(OCRS_ERR_IMAGE_SOURCE, "channel count is not 1, 3 or 4");
        if (len % channel_len != 0) fail(OCRS_ERR_IMAGE_SOURCE, "data length is not a multiple of `width * height`");
        const size_t ch = len / channel_len;
        if (!(ch == 1 || ch == 3 || ch == 4)) fail(OCRS_ERR_IMAGE_SOURCE, "channel count is not 1, 3 or 4");
        if (channels) *channels = (uint32_t)ch;
    });
}

ocrs_status ocrs_engine_prepare_input(const ocrs_engine* e, const void* pixels, ocrs_pixel_type type,
                                      ocrs_dim_order order, int height, int width, int channels, ocrs_page** out) {
    return guarded_engine(e, [&] {
        if (!e || !out) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        check_image_args(pixels, height, width, channels);
        Workspace ws;
        const size_t bytes = (size_t)height * width * channels * (type == OCRS_U8 ? 1 : 4);
        void* d_px = ws.alloc(bytes);
        OCRS_HIP(hipMemcpyAsync(d_px, pixels, bytes, hipMemcpyHostToDevice, ws.s()));
        ocrs_page* p = make_page(d_px, type, order, height, width, channels, ws.s(), e->tm());
        ws.sync();
        if (e->tm()) e->tm()->collect();
        *out = p;
    });
}

ocrs_status ocrs_engine_prepare_input_batch(const ocrs_engine* e, const void* const* pixels, size_t n, ocrs_pixel_type type,
                                            ocrs_dim_order order, int height, int width, int channels, ocrs_page** out) {
    return guarded_engine(e, [&] {
        if (!e || !out || (n > 0 && !pixels)) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        for (size_t i = 0; i < n; i++) check_image_args(pixels[i], height, width, channels);
        Workspace ws;
        const size_t bytes = (size_t)height * width * channels * (type == OCRS_U8 ? 1 : 4);
        std::vector<std::unique_ptr<ocrs_page>> made;   // freed if a later page fails
        for (size_t i = 0; i < n; i++) {
            // every copy and conversion is queued before the one wait below: from pinned memory (ocrs_host_malloc)
            // the copies are DMA transfers that overlap the conversion kernels of the pages before them
            void* d_px = ws.alloc(bytes);
            OCRS_HIP(hipMemcpyAsync(d_px, pixels[i], bytes, hipMemcpyHostToDevice, ws.s()));
            made.emplace_back(make_page(d_px, type, order, height, width, channels, ws.s(), e->tm()));
        }
        ws.sync();
        if (e->tm()) e->tm()->collect();
        for (size_t i = 0; i < n; i++) out[i] = made[i].release();
    });
}

ocrs_status ocrs_engine_prepare_input_device(const ocrs_engine* e, const void* d_pixels, ocrs_pixel_type type,
                                             ocrs_dim_order order, int height, int width, int channels,
                                             ocrs_page** out) {
    return guarded_engine(e, [&] {
        if (!e || !out) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        check_image_args(d_pixels, height, width, channels);
        Workspace ws;
        ocrs_page* p = make_page(d_pixels, type, order, height, width, channels, ws.s(), e->tm());
        ws.sync();
        if (e->tm()) e->tm()->collect();
        *out = p;
    });
}

namespace {
// Host entropy decode -> sparse coefficients to the device -> IDCT / upsampling / colour on the GPU: RGB u8 HWC in `ws`.
// *coef_bytes = what crossed PCIe instead of width * height * 3 bytes of pixels.
uint8_t* jpeg_to_device_rgb(Workspace& ws, const void* jpeg, size_t len, int* height, int* width, size_t* coef_bytes) {
    const ocrs::jpeg::Coefficients c = ocrs::jpeg::decode_coefficients(static_cast<const uint8_t*>(jpeg), len);
    const size_t nb = c.nblocks();
    uint64_t* d_mask = ws.alloc_n<uint64_t>(nb);
    uint32_t* d_off = ws.alloc_n<uint32_t>(nb + 1);
    int16_t* d_val = ws.alloc_n<int16_t>(c.values.size() + 1);
    uint16_t* d_q = ws.alloc_n<uint16_t>(4 * 64);
    ws.upload(d_mask, c.mask.data(), nb * sizeof(uint64_t));
    ws.upload(d_off, c.offset.data(), (nb + 1) * sizeof(uint32_t));
    ws.upload(d_val, c.values.data(), c.values.size() * sizeof(int16_t));
    ws.upload(d_q, c.quant, sizeof c.quant);
    uint8_t* d_samples = ws.alloc_n<uint8_t>(k::jpeg_sample_bytes(c));
    uint8_t* d_rgb = ws.alloc_n<uint8_t>((size_t)c.width * c.height * 3);
    k::jpeg_decode(c, d_mask, d_off, d_val, d_q, d_samples, d_rgb, ws.s());
    OCRS_HIP(hipGetLastError());
    *height = c.height;
    *width = c.width;
    if (coef_bytes) *coef_bytes = nb * 12 + 4 + c.values.size() * 2 + sizeof c.quant;
    return d_rgb;
}
}  // namespace

ocrs_status ocrs_engine_prepare_input_jpeg(const ocrs_engine* e, const void* jpeg, size_t len, ocrs_page** out, size_t* coef_bytes) {
    return guarded_engine(e, [&] {
        if (!e || !jpeg || !out) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        Workspace ws;
        int h = 0, w = 0;
        const uint8_t* d_rgb = jpeg_to_device_rgb(ws, jpeg, len, &h, &w, coef_bytes);
        ocrs_page* p = make_page(d_rgb, OCRS_U8, OCRS_HWC, h, w, 3, ws.s(), e->tm());
        ws.sync();
        if (e->tm()) e->tm()->collect();
        *out = p;
    });
}

ocrs_status ocrs_jpeg_decode_rgb(int device, const void* jpeg, size_t len, uint8_t** rgb, int* height, int* width, size_t* coef_bytes) {
    return guarded_on(device, [&] {
        if (!jpeg || !rgb || !height || !width) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        Workspace ws;
        const uint8_t* d_rgb = jpeg_to_device_rgb(ws, jpeg, len, height, width, coef_bytes);
        std::vector<uint8_t> host((size_t)*height * *width * 3);
        ws.download(host.data(), d_rgb, host.size());
        ws.sync();
        *rgb = dup_buffer(host);
    });
}

ocrs_status ocrs_jpeg_info(const void* jpeg, size_t len, int* height, int* width, int* components, int* progressive, size_t* nonzero) {
    return guarded([&] {
        if (!jpeg) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        const ocrs::jpeg::Coefficients c = ocrs::jpeg::decode_coefficients(static_cast<const uint8_t*>(jpeg), len);
        if (height) *height = c.height;
        if (width) *width = c.width;
        if (components) *components = c.ncomp;
        if (progressive) *progressive = c.progressive ? 1 : 0;
        if (nonzero) *nonzero = c.values.size();
    });
}

ocrs_status ocrs_jpeg_coefficients(const void* jpeg, size_t len, int32_t geom[28], uint16_t quant[256], int16_t** coef, size_t* n_blocks) {
    return guarded([&] {
        if (!jpeg || !geom || !quant || !coef || !n_blocks) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        const ocrs::jpeg::Coefficients c = ocrs::jpeg::decode_coefficients(static_cast<const uint8_t*>(jpeg), len);
        const int32_t head[7] = {c.width, c.height, c.ncomp, c.hmax, c.vmax, c.progressive ? 1 : 0, c.ycc ? 1 : 0};
        memcpy(geom, head, sizeof head);
        for (int i = 0; i < 3; i++) {
            const auto& k = c.comp[i];
            const int32_t row[7] = {k.h, k.v, k.tq, k.width, k.height, k.blocks_w, k.blocks_h};
            memcpy(geom + 7 + 7 * i, row, sizeof row);
        }
        memcpy(quant, c.quant, sizeof c.quant);
        std::vector<int16_t> dense(c.nblocks() * 64, 0);
        for (size_t b = 0; b < c.nblocks(); b++) {
            uint32_t at = c.offset[b];
            for (int p = 0; p < 64; p++) {
                const int z = ocrs::jpeg::Coefficients::kZigzagOfNatural[p];
                if ((c.mask[b] >> z) & 1) dense[b * 64 + p] = c.values[c.offset[b] + __builtin_popcountll(c.mask[b] & ((uint64_t(1) << z) - 1))];
            }
            (void)at;
        }
        *coef = dup_buffer(dense);
        *n_blocks = c.nblocks();
    });
}

void ocrs_page_free(ocrs_page* p) { delete p; }

ocrs_status ocrs_page_dims(const ocrs_page* p, int* height, int* width) {
    return guarded([&] {
        if (!p) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        if (height) *height = p->h;
        if (width) *width = p->w;
    });
}

ocrs_status ocrs_page_image(const ocrs_page* p, float* out_hw) {
    return guarded_on(p ? p->device() : -1, [&] {
        if (!p || !out_hw) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        OCRS_HIP(hipMemcpy(out_hw, p->grey.p, (size_t)p->h * p->w * sizeof(float), hipMemcpyDeviceToHost));
    });
}

ocrs_status ocrs_engine_detect_words_batch(const ocrs_engine* e, const ocrs_page* const* pages, size_t n_pages,
                                           float** rects, size_t* offsets) {
    return guarded_engine(e, [&] {
        if (!e || !pages || !rects || !offsets) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        check_pages_on(e, pages, n_pages);
        std::vector<std::vector<RotatedRect>> rr;
        e->detect(pages, n_pages, &rr, nullptr);
        std::vector<float> flat;
        offsets[0] = 0;
        for (size_t i = 0; i < n_pages; i++) {
            for (const RotatedRect& r : rr[i]) {
                float a[6];
                r.to_array(a);
                flat.insert(flat.end(), a, a + 6);
            }
            offsets[i + 1] = flat.size() / 6;
        }
        *rects = dup_buffer(flat);
    });
}

ocrs_status ocrs_engine_detect_words(const ocrs_engine* e, const ocrs_page* page, float** rects, size_t* n) {
    size_t offs[2] = {0, 0};
    ocrs_status s = ocrs_engine_detect_words_batch(e, &page, page ? 1 : 0, rects, offs);
    if (s == OCRS_OK && n) *n = offs[1];
    return s;
}

ocrs_status ocrs_engine_detect_text_pixels(const ocrs_engine* e, const ocrs_page* page, float* out_hw) {
    return guarded_engine(e, [&] {
        if (!e || !page || !out_hw) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        check_pages_on(e, &page, 1);
        e->detect(&page, 1, nullptr, out_hw);
    });
}

float ocrs_engine_detection_threshold(const ocrs_engine* e) { return e ? e->text_threshold : 0.2f; }

ocrs_status ocrs_engine_find_text_lines(const ocrs_engine* e, const ocrs_page* page, const float* word_rects,
                                        size_t n_words, float** line_rects, size_t** line_offsets, size_t* n_lines) {
    (void)e; (void)page;
    return guarded([&] {
        if (!line_rects || !line_offsets || !n_lines || (n_words && !word_rects))
            fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        std::vector<RotatedRect> words(n_words);
        for (size_t i = 0; i < n_words; i++) words[i] = RotatedRect::from_array(word_rects + 6 * i);
        auto lines = find_text_lines(words);
        std::vector<float> flat;
        std::vector<size_t> offs{0};
        for (const auto& l : lines) {
            for (const RotatedRect& r : l) {
                float a[6];
                r.to_array(a);
                flat.insert(flat.end(), a, a + 6);
            }
            offs.push_back(flat.size() / 6);
        }
        *line_rects = dup_buffer(flat);
        *line_offsets = dup_buffer(offs);
        *n_lines = lines.size();
    });
}

ocrs_status ocrs_engine_find_text_lines_batch(const ocrs_engine* e, size_t n_pages, const float* word_rects,
                                              const size_t* word_offsets, float** line_rects, size_t** line_offsets,
                                              size_t** page_line_offsets) {
    return guarded([&] {
        TuningScope tune(e ? &e->tuning : nullptr);   // option "layout_threads" of this engine
        if (!word_offsets || !line_rects || !line_offsets || !page_line_offsets)
            fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        std::vector<std::vector<std::vector<RotatedRect>>> per_page(n_pages);
        std::vector<std::string> errors(n_pages);
        auto work = [&](size_t p) {
            try {
                std::vector<RotatedRect> words;
                for (size_t k = word_offsets[p]; k < word_offsets[p + 1]; k++)
                    words.push_back(RotatedRect::from_array(word_rects + 6 * k));
                per_page[p] = find_text_lines(words);
            } catch (const std::exception& ex) {
                errors[p] = ex.what();
            }
        };
        // a bounded pool pulling pages from a shared counter: option "layout_threads" (0 = one thread per
        // page up to the host's cores) keeps N ranks x in-flight requests from oversubscribing one host
        const int opt = option(OPT_LAYOUT_THREADS);
        const size_t hw = std::max(1u, std::thread::hardware_concurrency());
        for_pages(n_pages, opt > 0 ? (size_t)opt : hw, work);
        for (const std::string& er : errors)
            if (!er.empty()) fail(OCRS_ERR_RUN_FAILED, "%s", er.c_str());
        std::vector<float> flat;
        std::vector<size_t> loffs{0}, poffs{0};
        for (size_t p = 0; p < n_pages; p++) {
            for (const auto& l : per_page[p]) {
                for (const RotatedRect& r : l) {
                    float a[6];
                    r.to_array(a);
                    flat.insert(flat.end(), a, a + 6);
                }
                loffs.push_back(flat.size() / 6);
            }
            poffs.push_back(loffs.size() - 1);
        }
        *line_rects = dup_buffer(flat);
        *line_offsets = dup_buffer(loffs);
        *page_line_offsets = dup_buffer(poffs);
    });
}

ocrs_status ocrs_engine_recognize_text_batch(const ocrs_engine* e, const ocrs_page* const* pages, size_t n_pages,
                                             const size_t* page_line_offsets, const float* line_rects,
                                             const size_t* line_offsets, size_t n_lines, ocrs_text_char** chars,
                                             size_t** char_offsets) {
    return guarded_engine(e, [&] {
        if (!e || !pages || !page_line_offsets || !line_offsets || !chars || !char_offsets)
            fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        check_pages_on(e, pages, n_pages);
        if (page_line_offsets[n_pages] != n_lines) fail(OCRS_ERR_INVALID_ARGUMENT, "page_line_offsets do not cover n_lines");
        std::vector<std::vector<std::vector<RotatedRect>>> lpp(n_pages);
        for (size_t p = 0; p < n_pages; p++)
            lpp[p] = unpack_lines(line_rects, line_offsets, page_line_offsets[p], page_line_offsets[p + 1]);
        std::vector<std::vector<CtcStep>> steps;
        std::vector<RecLine> rl;
        std::vector<uint32_t> ctc_len;
        e->recognize(pages, n_pages, lpp, &steps, &rl, &ctc_len);
        std::vector<ocrs_text_char> flat;
        std::vector<size_t> offs{0};
        for (size_t i = 0; i < rl.size(); i++) {
            for (const TextChar& c : e->text_line_from_result(rl[i], ctc_len[i], steps[i]))
                flat.push_back(ocrs_text_char{c.ch, c.rect.top, c.rect.left, c.rect.bottom, c.rect.right});
            offs.push_back(flat.size());
        }
        *chars = dup_buffer(flat);
        *char_offsets = dup_buffer(offs);
    });
}

ocrs_status ocrs_engine_recognize_text(const ocrs_engine* e, const ocrs_page* page, const float* line_rects,
                                       const size_t* line_offsets, size_t n_lines, ocrs_text_char** chars,
                                       size_t** char_offsets) {
    size_t plo[2] = {0, n_lines};
    return ocrs_engine_recognize_text_batch(e, &page, 1, plo, line_rects, line_offsets, n_lines, chars, char_offsets);
}

ocrs_status ocrs_engine_recognize_tokens(const ocrs_engine* e, const ocrs_page* page, const float* line_rects,
                                         const size_t* line_offsets, size_t n_lines, uint32_t** labels,
                                         uint32_t** positions, size_t** token_offsets) {
    return guarded_engine(e, [&] {
        if (!e || !page || !line_offsets || !labels || !positions || !token_offsets)
            fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        check_pages_on(e, &page, 1);
        std::vector<std::vector<std::vector<RotatedRect>>> lpp(1);
        lpp[0] = unpack_lines(line_rects, line_offsets, 0, n_lines);
        std::vector<std::vector<CtcStep>> steps;
        std::vector<RecLine> rl;
        std::vector<uint32_t> ctc_len;
        e->recognize(&page, 1, lpp, &steps, &rl, &ctc_len);
        std::vector<uint32_t> fl, fp;
        std::vector<size_t> offs{0};
        for (const auto& s : steps) {
            for (const CtcStep& c : s) { fl.push_back(c.label); fp.push_back(c.pos); }
            offs.push_back(fl.size());
        }
        *labels = dup_buffer(fl);
        *positions = dup_buffer(fp);
        *token_offsets = dup_buffer(offs);
    });
}

ocrs_status ocrs_engine_recognize_logits(const ocrs_engine* e, const ocrs_page* page, const float* line_rects,
                                         const size_t* line_offsets, size_t n_lines, float** logp, size_t** row_offsets, int* classes) {
    return guarded_engine(e, [&] {
        if (!e || !page || !line_offsets || !logp || !row_offsets || !classes) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        check_pages_on(e, &page, 1);
        std::vector<std::vector<float>> per_line;
        e->recognize_logits(page, unpack_lines(line_rects, line_offsets, 0, n_lines), &per_line, classes);
        std::vector<float> flat;
        std::vector<size_t> offs{0};
        for (const auto& l : per_line) {
            flat.insert(flat.end(), l.begin(), l.end());
            offs.push_back(flat.size() / (size_t)*classes);
        }
        *logp = dup_buffer(flat);
        *row_offsets = dup_buffer(offs);
    });
}

ocrs_status ocrs_text_item_rotated_rect(const int32_t* rects_tlbr, size_t n_chars, float out6[6]) {
    return guarded([&] {
        if (!rects_tlbr || !out6 || n_chars == 0) fail(OCRS_ERR_INVALID_ARGUMENT, "expected valid rect");
        RotatedRect rr;
        if (!text_item_rotated_rect(rects_tlbr, n_chars, &rr)) fail(OCRS_ERR_INVALID_ARGUMENT, "expected valid rect");
        rr.to_array(out6);
    });
}

ocrs_status ocrs_rotated_rect_corners(const float rect6[6], float out8[8]) {
    return guarded([&] {
        if (!rect6 || !out8) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        auto c = RotatedRect::from_array(rect6).corners();
        for (int i = 0; i < 4; i++) { out8[2 * i] = c[i].x; out8[2 * i + 1] = c[i].y; }
    });
}

ocrs_status ocrs_engine_prepare_recognition_input(const ocrs_engine* e, const ocrs_page* page, const float* line,
                                                  size_t n_words, float** out, int* height, int* width) {
    return guarded_engine(e, [&] {
        if (!e || !page || !line || !out || !height || !width) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        check_pages_on(e, &page, 1);
        if (!e->recognition) fail(OCRS_ERR_MODEL_NOT_LOADED, "Recognition model not loaded");
        std::vector<RotatedRect> words(n_words);
        for (size_t i = 0; i < n_words; i++) words[i] = RotatedRect::from_array(line + 6 * i);
        RecLine ln = e->make_rec_line(words, 0, 0);
        const int rec_h = (int)e->rec_input_height();
        const int rw = (int)ln.resized_width;
        Workspace ws;
        k::LineDesc d{};
        d.page = 0; d.poly_off = 0; d.poly_n = (int32_t)ln.polygon.size();
        d.top = ln.bounds.top; d.left = ln.bounds.left; d.bh = ln.bounds.height(); d.bw = ln.bounds.width();
        d.resized_w = rw; d.out_w = rw; d.out_off = 0;
        std::vector<int32_t> poly;
        for (const PointI& p : ln.polygon) { poly.push_back(p.y); poly.push_back(p.x); }
        const float* hp = page->grey.as<float>();
        const int32_t hw[2] = {page->h, page->w};
        const float** d_pages = ws.alloc_n<const float*>(1);
        int32_t* d_hw = ws.alloc_n<int32_t>(2);
        k::LineDesc* d_desc = ws.alloc_n<k::LineDesc>(1);
        int32_t* d_poly = ws.alloc_n<int32_t>(poly.size());
        float* d_out = ws.alloc_n<float>((size_t)rec_h * std::max(rw, 1));
        OCRS_HIP(hipMemcpyAsync(d_pages, &hp, sizeof hp, hipMemcpyHostToDevice, ws.s()));
        OCRS_HIP(hipMemcpyAsync(d_hw, hw, sizeof hw, hipMemcpyHostToDevice, ws.s()));
        OCRS_HIP(hipMemcpyAsync(d_desc, &d, sizeof d, hipMemcpyHostToDevice, ws.s()));
        OCRS_HIP(hipMemcpyAsync(d_poly, poly.data(), poly.size() * 4, hipMemcpyHostToDevice, ws.s()));
        std::vector<float> host((size_t)rec_h * rw);
        if (rw > 0) {
            k::crop_lines(d_pages, d_hw, d_desc, d_poly, 1, rec_h, d_out, ws.s());
            ws.download(host.data(), d_out, host.size() * 4);
        }
        ws.sync();
        *out = dup_buffer(host);
        *height = rec_h;
        *width = rw;
    });
}

ocrs_status ocrs_engine_get_text(const ocrs_engine* e, const ocrs_page* page, char** text) {
    return guarded_engine(e, [&] {  // lib.rs:290-300
        if (!e || !page || !text) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        check_pages_on(e, &page, 1);
        std::vector<std::vector<RotatedRect>> rr;
        e->detect(&page, 1, &rr, nullptr);
        std::vector<std::vector<std::vector<RotatedRect>>> lpp(1);
        lpp[0] = find_text_lines(rr[0]);
        std::vector<std::vector<CtcStep>> steps;
        std::vector<RecLine> rl;
        std::vector<uint32_t> ctc_len;
        e->recognize(&page, 1, lpp, &steps, &rl, &ctc_len);
        std::string out;
        bool first = true;
        for (size_t i = 0; i < rl.size(); i++) {
            auto chars = e->text_line_from_result(rl[i], ctc_len[i], steps[i]);
            if (chars.empty()) continue;
            if (!first) out.push_back('\n');
            first = false;
            for (const TextChar& c : chars) append_utf8(out, c.ch);
        }
        char* p = static_cast<char*>(malloc(out.size() + 1));
        if (!p) throw std::bad_alloc();
        memcpy(p, out.c_str(), out.size() + 1);
        *text = p;
    });
}

// ------------------------------------------------------------------ measurement hooks
ocrs_status ocrs_device_malloc(size_t bytes, void** d_ptr) {
    return guarded_on(-1, [&] { OCRS_HIP(hipMalloc(d_ptr, bytes)); });
}
ocrs_status ocrs_device_malloc_on(int device, size_t bytes, void** d_ptr) {
    return guarded_on(device, [&] { OCRS_HIP(hipMalloc(d_ptr, bytes)); });
}
ocrs_status ocrs_device_free(void* d_ptr) {
    return guarded_on(-1, [&] { OCRS_HIP(hipFree(d_ptr)); });
}
ocrs_status ocrs_device_upload(void* d_dst, const void* h_src, size_t bytes) {
    return guarded_on(-1, [&] { OCRS_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice)); });
}
ocrs_status ocrs_host_malloc(size_t bytes, void** h_ptr) {
    return guarded_on(-1, [&] {
        if (!h_ptr) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        OCRS_HIP(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocPortable));   // usable with every device of a group
    });
}
ocrs_status ocrs_host_free(void* h_ptr) {
    return guarded_on(-1, [&] { OCRS_HIP(hipHostFree(h_ptr)); });
}
ocrs_status ocrs_device_synchronize(void) {
    return guarded_on(-1, [&] { OCRS_HIP(hipDeviceSynchronize()); });
}

ocrs_status ocrs_device_measure_peaks(double* mfma_f32_tflops, double* hbm_copy_gbps) {
    return guarded_on(-1, [&] {
        if (!mfma_f32_tflops || !hbm_copy_gbps) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        k::measure_peaks(mfma_f32_tflops, hbm_copy_gbps);
    });
}

ocrs_status ocrs_engine_enable_timing(ocrs_engine* e, int enable) {
    return guarded_engine(e, [&] {
        if (!e) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        e->timers.enabled = enable != 0;
        e->timers.kernels_enabled = enable >= 2;
    });
}
ocrs_status ocrs_engine_set_kernel_timing_mask(ocrs_engine* e, uint32_t mask) {
    return guarded_engine(e, [&] {
        if (!e) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        e->timers.kernel_mask = mask;
    });
}
ocrs_status ocrs_engine_coalesce_stats(const ocrs_engine* e, uint64_t detect[2], uint64_t recognize[2]) {
    return guarded([&] {
        if (!e || !detect || !recognize) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        detect[0] = detect[1] = recognize[0] = recognize[1] = 0;
        if (e->det_queue) e->det_queue->stats(&detect[0], &detect[1]);
        if (e->rec_queue) e->rec_queue->stats(&recognize[0], &recognize[1]);
    });
}
int ocrs_kernel_class_count(void) { return KC_COUNT; }
const char* ocrs_kernel_class_name(int cls) { return cls >= 0 && cls < KC_COUNT ? kKernelClassNames[cls] : ""; }
ocrs_status ocrs_engine_kernel_stats(ocrs_engine* e, double* ms, uint64_t* launches, double* flops, double* bytes,
                                     int reset) {
    return guarded_engine(e, [&] {
        if (!e) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        e->timers.collect();
        for (int i = 0; i < KC_COUNT; i++) {
            if (ms) ms[i] = e->timers.kms[i];
            if (launches) launches[i] = e->timers.klaunches[i];
            if (flops) flops[i] = e->timers.kflops[i];
            if (bytes) bytes[i] = e->timers.kbytes[i];
        }
        if (reset) e->timers.reset();
    });
}
ocrs_status ocrs_engine_kernel_mfma_flops(ocrs_engine* e, double* mfma_flops) {
    return guarded_engine(e, [&] {
        if (!e || !mfma_flops) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        e->timers.collect();
        for (int i = 0; i < KC_COUNT; i++) mfma_flops[i] = e->timers.kmfma[i];
    });
}
int ocrs_stage_count(void) { return ST_COUNT; }
const char* ocrs_stage_name(int stage) { return stage >= 0 && stage < ST_COUNT ? kStageNames[stage] : ""; }
ocrs_status ocrs_engine_stage_times(ocrs_engine* e, double* ms, uint64_t* launches, int reset) {
    return guarded_engine(e, [&] {
        if (!e) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        e->timers.collect();
        for (int i = 0; i < ST_COUNT; i++) {
            if (ms) ms[i] = e->timers.ms[i];
            if (launches) launches[i] = e->timers.launches[i];
        }
        if (reset) e->timers.reset();
    });
}

}  // extern "C"
