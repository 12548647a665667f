#include "common.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>

#include <atomic>

namespace ocrs {

static thread_local std::string g_last_error;

void set_last_error(const std::string& msg) { g_last_error = msg; }
const std::string& last_error() { return g_last_error; }

namespace {
std::atomic<int> g_device{-1};
thread_local int t_bound_device = -1;
}  // namespace

void select_device(int device) {
    OCRS_HIP(hipSetDevice(device));
    g_device.store(device);
    t_bound_device = device;
}

void bind_thread_to_device() {
    const int d = g_device.load();
    if (d >= 0 && t_bound_device != d) {
        OCRS_HIP(hipSetDevice(d));
        t_bound_device = d;
    }
}

const char* const kStageNames[ST_COUNT] = {
    "prepare_image", "resize_to_model", "detection_cnn", "resize_threshold", "ccl",      "contour_rects",
    "line_crop",     "rec_conv",        "rec_gru",       "rec_head",         "ctc_decode"};

const char* const kKernelClassNames[KC_COUNT] = {
    "gemm_conv3x3_mfma", "gemm_pointwise_mfma", "gemm_convt_mfma", "gemm_gru_input_mfma", "gemm_gru_hidden_mfma",
    "gemm_linear_mfma",  "dwconv3x3",           "conv_direct",     "pool",                "padcat",
    "conv1x1_sigmoid",   "gru_gates",           "logsoftmax_argmax", "other"};

// ---------------------------------------------------------------- DevicePool
static size_t round_size(size_t n) {
    // 256 B granularity below 1 MiB, then 1/8-octave buckets: bounded waste, good reuse.
    if (n <= (1u << 20)) return (n + 255) & ~size_t(255);
    size_t p = size_t(1) << 20;
    while (p * 2 <= n) p *= 2;
    size_t step = p / 8;
    return ((n + step - 1) / step) * step;
}

void* DevicePool::alloc(size_t bytes) {
    size_t sz = round_size(bytes);
    {
        std::lock_guard<std::mutex> g(mu_);
        auto it = free_.find(sz);
        if (it != free_.end()) {
            void* p = it->second;
            free_.erase(it);
            live_[p] = sz;
            return p;
        }
    }
    void* p = nullptr;
    static const bool trace = getenv("OCRS_POOL_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
        trim();
        OCRS_HIP(hipMalloc(&p, sz));
    }
    if (trace) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        fprintf(stderr, "[pool] t=%.3f hipMalloc %.1f MB took %.2f ms\n", now, sz / 1e6, ms);
    }
    std::lock_guard<std::mutex> g(mu_);
    live_[p] = sz;
    return p;
}

void DevicePool::release(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu_);
    auto it = live_.find(p);
    if (it == live_.end()) return;
    free_.emplace(it->second, p);
    live_.erase(it);
}

void DevicePool::trim() {
    std::lock_guard<std::mutex> g(mu_);
    for (auto& kv : free_) (void)hipFree(kv.second);
    free_.clear();
}

DevicePool::~DevicePool() {
    // Process teardown: the HIP runtime may already be gone; leak on purpose.
}

DevicePool& pool() {
    static DevicePool* p = new DevicePool();
    return *p;
}

// ---------------------------------------------------------------- HostPool
void* HostPool::alloc(size_t bytes) {
    const size_t sz = round_size(bytes);
    {
        std::lock_guard<std::mutex> g(mu_);
        auto it = free_.find(sz);
        if (it != free_.end()) {
            void* p = it->second;
            free_.erase(it);
            live_[p] = sz;
            return p;
        }
    }
    void* p = nullptr;
    OCRS_HIP(hipHostMalloc(&p, sz, hipHostMallocDefault));
    std::lock_guard<std::mutex> g(mu_);
    live_[p] = sz;
    return p;
}

void HostPool::release(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu_);
    auto it = live_.find(p);
    if (it == live_.end()) return;
    free_.emplace(it->second, p);
    live_.erase(it);
}

HostPool& host_pool() {
    static HostPool* p = new HostPool();
    return *p;
}

hipStream_t heavy_stream() {
    static hipStream_t s = [] {
        // Lowest queue priority: the conv stacks are long throughput-bound grids whose blocks live ~200 us;
        // the latency-bound kernels of the request streams (1 200 dependent GRU steps per request) must get
        // the slots those blocks free first.  Measured on the default bench: 182/176 -> 188/185 pages/s, and
        // with the request streams at the highest priority and 6 steps in flight 199/207.
        hipStream_t h;
        int least = 0, greatest = 0;
        OCRS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        OCRS_HIP(hipStreamCreateWithPriority(&h, hipStreamNonBlocking, least));
        return h;
    }();
    return s;
}

// ---------------------------------------------------------------- streams
namespace {
std::mutex g_stream_mu;
std::vector<std::pair<hipStream_t, hipEvent_t>> g_streams[2];  // [0] default priority, [1] highest priority
}  // namespace

StreamLease::StreamLease(bool high_priority) : high_(high_priority) {
    high_ = true;  // every request stream outranks the shared conv-stack stream (see heavy_stream())
    {
        std::lock_guard<std::mutex> g(g_stream_mu);
        auto& v = g_streams[high_ ? 1 : 0];
        if (!v.empty()) {
            s_ = v.back().first;
            done_ = v.back().second;
            v.pop_back();
            return;
        }
    }
    if (high_) {
        int least = 0, greatest = 0;
        OCRS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        OCRS_HIP(hipStreamCreateWithPriority(&s_, hipStreamNonBlocking, greatest));
    } else {
        OCRS_HIP(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking));
    }
    OCRS_HIP(hipEventCreateWithFlags(&done_, hipEventBlockingSync | hipEventDisableTiming));
}

StreamLease::~StreamLease() {
    std::lock_guard<std::mutex> g(g_stream_mu);
    g_streams[high_ ? 1 : 0].emplace_back(s_, done_);
}

// ---------------------------------------------------------------- timers
std::vector<StageTimers::Pending>& StageTimers::pending() {
    static thread_local std::vector<Pending> p;
    return p;
}
std::vector<hipEvent_t>& StageTimers::free_events() {
    static thread_local std::vector<hipEvent_t> f;
    return f;
}

hipEvent_t StageTimers::get_event() {
    auto& fe = free_events();
    if (!fe.empty()) {
        hipEvent_t e = fe.back();
        fe.pop_back();
        return e;
    }
    hipEvent_t e;
    OCRS_HIP(hipEventCreate(&e));
    return e;
}

int StageTimers::begin(int stage, hipStream_t s, uint64_t n_launches) {
    if (!enabled) return -1;
    Pending p{stage, get_event(), get_event(), n_launches, false, 0.0, 0.0};
    OCRS_HIP(hipEventRecord(p.a, s));
    pending().push_back(p);
    return (int)pending().size() - 1;
}

int StageTimers::kbegin(int cls, hipStream_t s, double flops, double bytes) {
    if (!enabled || !kernels_enabled || !((kernel_mask >> cls) & 1u)) return -1;
    Pending p{cls, get_event(), get_event(), 1, true, flops, bytes};
    OCRS_HIP(hipEventRecord(p.a, s));
    pending().push_back(p);
    return (int)pending().size() - 1;
}

void StageTimers::end(int token, hipStream_t s) {
    if (token < 0) return;
    auto& pd = pending();
    if ((size_t)token < pd.size()) (void)hipEventRecord(pd[token].b, s);
}

void StageTimers::collect() {
    auto& pd = pending();
    if (pd.empty()) return;
    double lms[ST_COUNT] = {0}, lkms[KC_COUNT] = {0}, lkf[KC_COUNT] = {0}, lkb[KC_COUNT] = {0};
    uint64_t ln[ST_COUNT] = {0}, lkn[KC_COUNT] = {0};
    for (auto& p : pd) {
        float t = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) {
            if (p.kernel) { lkms[p.stage] += t; lkn[p.stage] += 1; lkf[p.stage] += p.flops; lkb[p.stage] += p.bytes; }
            else { lms[p.stage] += t; ln[p.stage] += p.n; }
        }
        free_events().push_back(p.a);
        free_events().push_back(p.b);
    }
    pd.clear();
    std::lock_guard<std::mutex> g(mu);
    for (int i = 0; i < ST_COUNT; i++) { ms[i] += lms[i]; launches[i] += ln[i]; }
    for (int i = 0; i < KC_COUNT; i++) { kms[i] += lkms[i]; klaunches[i] += lkn[i]; kflops[i] += lkf[i]; kbytes[i] += lkb[i]; }
}

void StageTimers::reset() {
    std::lock_guard<std::mutex> g(mu);
    for (int i = 0; i < ST_COUNT; i++) { ms[i] = 0; launches[i] = 0; }
    for (int i = 0; i < KC_COUNT; i++) { kms[i] = 0; klaunches[i] = 0; kflops[i] = 0; kbytes[i] = 0; }
}

}  // namespace ocrs
