#include "common.hpp"

#include <chrono>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <system_error>

namespace ocrs {

static thread_local std::string g_last_error;

void set_last_error(const std::string& msg) { g_last_error = msg; }
const std::string& last_error() { return g_last_error; }

namespace {
std::atomic<int> g_default_device{0};
thread_local DeviceContext* t_ctx = nullptr;   // set by DeviceScope
constexpr int kMaxDevices = 64;
std::mutex g_ctx_mu;
std::atomic<DeviceContext*> g_ctx[kMaxDevices];
}  // namespace

DeviceContext& device_context(int device) {
    if (device < 0 || device >= kMaxDevices) fail(OCRS_ERR_INVALID_ARGUMENT, "device index %d out of range", device);
    DeviceContext* c = g_ctx[device].load(std::memory_order_acquire);
    if (c) return *c;
    std::lock_guard<std::mutex> g(g_ctx_mu);
    c = g_ctx[device].load(std::memory_order_relaxed);
    if (!c) {
        c = new DeviceContext(device);   // lives for the process (HIP may be gone by the time statics are destroyed)
        g_ctx[device].store(c, std::memory_order_release);
    }
    return *c;
}

int default_device() { return g_default_device.load(); }

void select_device(int device) {
    int n = 0;
    OCRS_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) fail(OCRS_ERR_INVALID_ARGUMENT, "device %d does not exist (%d visible)", device, n);
    OCRS_HIP(hipSetDevice(device));
    g_default_device.store(device);
}

DeviceContext& ctx() {
    if (t_ctx) return *t_ctx;
    return device_context(default_device());
}

DeviceScope::DeviceScope(int device) : prev_(t_ctx) {
    DeviceContext& c = device_context(device < 0 ? default_device() : device);
    // The binding lasts for the call only: the outermost scope remembers the device the caller's thread was on (torch
    // or the caller's own HIP code may have selected another GPU) and puts it back on the way out.
    if (!prev_ && hipGetDevice(&restore_) != hipSuccess) {
        (void)hipGetLastError();
        restore_ = -1;
    }
    // always: other code on this thread (torch, the caller) may have switched the thread's device between our calls
    const hipError_t e = hipSetDevice(c.device);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // HIP keeps the failure as the thread's "last error": do not leave it for a later, unrelated check
        fail(OCRS_ERR_DEVICE, "cannot bind to HIP device %d: %s", c.device, hipGetErrorString(e));
    }
    t_ctx = &c;
}

DeviceScope::~DeviceScope() {
    const int here = t_ctx ? t_ctx->device : -1;
    t_ctx = prev_;
    const int back = prev_ ? prev_->device : restore_;
    if (back >= 0 && back != here && hipSetDevice(back) != hipSuccess) (void)hipGetLastError();
}

const char* const kStageNames[ST_COUNT] = {
    "prepare_image", "resize_to_model", "detection_cnn", "resize_threshold", "ccl",      "contour_rects",
    "line_crop",     "rec_conv",        "rec_gru",       "rec_head",         "ctc_decode"};

const char* const kKernelClassNames[KC_COUNT] = {
    "gemm_conv3x3_mfma", "gemm_pointwise_mfma", "gemm_convt_mfma", "gemm_gru_input_mfma", "gemm_gru_hidden_mfma",
    "gemm_linear_mfma",  "dwconv3x3",           "conv_direct",     "pool",                "padcat",
    "conv1x1_sigmoid",   "gru_gates",           "logsoftmax_argmax", "other",
    "det_fused_block",   "det_stream_wave_block", "det_stream_rows_block"};

// ---------------------------------------------------------------- DevicePool
static size_t round_size(size_t n) {
    // 256 B granularity below 1 MiB, then 1/8-octave buckets: bounded waste, good reuse.
    if (n <= (1u << 20)) return (n + 255) & ~size_t(255);
    size_t p = size_t(1) << 20;
    while (p * 2 <= n) p *= 2;
    size_t step = p / 8;
    return ((n + step - 1) / step) * step;
}

// ---- the trimmer: one detached thread per process that returns memory to the driver.  hipFree / hipHostFree wait for
// the device to go idle, which under load takes as long as the queued work: a request's thread never pays that.
namespace {
struct Trimmer;
Trimmer& trimmer();
struct Trimmer {
    struct Item { int device; void* p; std::atomic<uint64_t>* freed; };   // device -1: pinned host memory; freed: the pool's counter
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Item> q;
    bool started = false;
    std::atomic<bool> exiting{false};   // set by an atexit handler: the HIP runtime may be tearing down, leave the blocks to the OS
    std::mutex freeing;                 // held across every hipFree / hipHostFree: the atexit handler takes it to wait one out
    void run() {
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !q.empty(); });
                it = q.front();
                q.pop_front();
            }
            std::lock_guard<std::mutex> f(freeing);
            if (exiting.load()) continue;       // checked under `freeing`: once the handler has had the lock no free starts
            bool ok;
            if (it.device >= 0) ok = hipSetDevice(it.device) == hipSuccess && hipFree(it.p) == hipSuccess;
            else ok = hipHostFree(it.p) == hipSuccess;
            (void)hipGetLastError();
            if (ok && it.freed) it.freed->fetch_add(1, std::memory_order_relaxed);   // counted when the driver has the block back
        }
    }
    void give(int device, void* p, std::atomic<uint64_t>* freed) {
        std::lock_guard<std::mutex> lk(mu);
        if (!started) {
            started = true;
            // process exit: no free may be in progress while the HIP runtime's own exit handlers and static destructors run
            std::atexit([] { Trimmer& t = trimmer(); t.exiting.store(true); std::lock_guard<std::mutex> f(t.freeing); });
            try {
                std::thread([this] { run(); }).detach();
            } catch (const std::system_error&) {
                started = false;
            }
        }
        if (!started) {   // no thread to be had: free here after all
            const bool ok = (device >= 0 ? hipFree(p) : hipHostFree(p)) == hipSuccess;
            if (ok && freed) freed->fetch_add(1, std::memory_order_relaxed);
            return;
        }
        q.push_back(Item{device, p, freed});
        cv.notify_one();
    }
};
Trimmer& trimmer() {
    static Trimmer* t = new Trimmer;   // leaked on purpose: its thread may outlive static destruction
    return *t;
}
}  // namespace

uint64_t DevicePool::cap_locked() {
    if (!cap_) {
        // default: a quarter of the device's memory (72 GB on an MI355X: the 16-page bench request peaks at ~20 GB of
        // scratch, five requests in flight cache ~50 GB); OCRS_POOL_CAP_GB, read once per pool, overrides
        const char* e = getenv("OCRS_POOL_CAP_GB");
        if (e && *e) {
            cap_ = (uint64_t)(atof(e) * (double)(uint64_t(1) << 30));
        } else {
            size_t fr = 0, tot = 0;
            int cur = -1;
            const bool rebind = hipGetDevice(&cur) == hipSuccess && cur != device_;
            if (rebind) (void)hipSetDevice(device_);
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); tot = size_t(64) << 30; }
            if (rebind) (void)hipSetDevice(cur);
            cap_ = tot / 4;
        }
        if (!cap_) cap_ = 1;
    }
    return cap_;
}

void DevicePool::set_cap(uint64_t bytes) {
    std::vector<void*> drop;
    {
        std::lock_guard<std::mutex> g(mu_);
        cap_ = bytes ? bytes : 1;
        while (cached_ > cap_ && !free_.empty()) {
            auto big = std::prev(free_.end());
            drop.push_back(big->second);
            cached_ -= big->first;
            free_.erase(big);
        }
    }
    for (void* p : drop) trimmer().give(device_, p, &frees_);
}

PoolStats DevicePool::stats() {
    std::lock_guard<std::mutex> g(mu_);
    PoolStats st;
    st.live = live_bytes_; st.cached = cached_; st.cap = cap_locked(); st.peak_live = peak_live_; st.driver_allocs = allocs_; st.driver_frees = frees_;
    return st;
}

void* DevicePool::alloc(size_t bytes) {
    size_t sz = round_size(bytes);
    if (ctx().device != device_) fail(OCRS_ERR_DEVICE, "internal: allocation from device %d's pool on a thread bound to device %d", device_, ctx().device);
    {
        // smallest cached block that fits, if it wastes at most a quarter of the request
        std::lock_guard<std::mutex> g(mu_);
        auto it = free_.lower_bound(sz);
        if (it != free_.end() && it->first <= sz + sz / 4) {
            void* p = it->second;
            const size_t got = it->first;
            free_.erase(it);
            cached_ -= got;
            live_[p] = got;
            live_bytes_ += got;
            peak_live_ = std::max(peak_live_, live_bytes_);
            return p;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // the retry below decides; do not leave a stale out-of-memory for later checks
        trim();
        OCRS_HIP(hipMalloc(&p, sz));
    }
    std::lock_guard<std::mutex> g(mu_);
    live_[p] = sz;
    live_bytes_ += sz;
    peak_live_ = std::max(peak_live_, live_bytes_);
    allocs_++;
    return p;
}

void DevicePool::release(void* p) {
    if (!p) return;
    std::vector<void*> drop;
    {
        std::lock_guard<std::mutex> g(mu_);
        auto it = live_.find(p);
        if (it == live_.end()) return;
        free_.emplace(it->second, p);
        cached_ += it->second;
        live_bytes_ -= it->second;
        live_.erase(it);
        const uint64_t cap = cap_locked();
        while (cached_ > cap && !free_.empty()) {   // largest first; returned to the driver by the trimmer thread
            auto big = std::prev(free_.end());
            drop.push_back(big->second);
            cached_ -= big->first;
            free_.erase(big);
        }
    }
    for (void* q : drop) trimmer().give(device_, q, &frees_);
}

void DevicePool::trim() {
    std::lock_guard<std::mutex> g(mu_);
    for (auto& kv : free_) { (void)hipFree(kv.second); frees_++; }
    free_.clear();
    cached_ = 0;
}

DevicePool::~DevicePool() {
    // Process teardown: the HIP runtime may already be gone; leak on purpose.
}

// ---------------------------------------------------------------- HostPool
void* HostPool::alloc(size_t bytes) {
    const size_t sz = round_size(bytes);
    {
        std::lock_guard<std::mutex> g(mu_);
        auto it = free_.lower_bound(sz);
        if (it != free_.end() && it->first <= sz + sz / 4) {
            void* p = it->second;
            const size_t got = it->first;
            free_.erase(it);
            cached_ -= got;
            live_[p] = got;
            live_bytes_ += got;
            peak_live_ = std::max(peak_live_, live_bytes_);
            return p;
        }
    }
    void* p = nullptr;
    OCRS_HIP(hipHostMalloc(&p, sz, hipHostMallocPortable));
    std::lock_guard<std::mutex> g(mu_);
    live_[p] = sz;
    live_bytes_ += sz;
    peak_live_ = std::max(peak_live_, live_bytes_);
    allocs_++;
    return p;
}

void HostPool::release(void* p) {
    if (!p) return;
    std::vector<void*> drop;
    {
        std::lock_guard<std::mutex> g(mu_);
        auto it = live_.find(p);
        if (it == live_.end()) return;
        free_.emplace(it->second, p);
        cached_ += it->second;
        live_bytes_ -= it->second;
        live_.erase(it);
        while (cached_ > cap_ && !free_.empty()) {
            auto big = std::prev(free_.end());
            drop.push_back(big->second);
            cached_ -= big->first;
            free_.erase(big);
        }
    }
    for (void* q : drop) trimmer().give(-1, q, &frees_);
}

void HostPool::set_cap(uint64_t bytes) {
    std::lock_guard<std::mutex> g(mu_);
    cap_ = bytes;
}

PoolStats HostPool::stats() {
    std::lock_guard<std::mutex> g(mu_);
    PoolStats st;
    st.live = live_bytes_; st.cached = cached_; st.cap = cap_; st.peak_live = peak_live_; st.driver_allocs = allocs_; st.driver_frees = frees_;
    return st;
}

hipStream_t DeviceContext::heavy_stream() {
    std::lock_guard<std::mutex> g(lazy_mu_);
    if (!heavy_) {
        // Highest queue priority.  The conv stacks are the critical resource of the pipeline: their stream never runs
        // dry in steady state and a step takes as long as its conv stack does.  (Round 1 ran this stream at the LOWEST
        // priority so that the 1 200 dependent GRU step launches of a request would get freed CU slots first; with
        // the recurrence in one persistent launch per layer that reason is gone.  The GPU is work-conserving — every
        // combination of stream priorities measured the same 256-259 pages/s — but here the dominant kernels are
        // stretched least by what runs beside them: 9.9-10.4 ms per launch against 11.2-11.4.)
        // Round 3 tried the lowest priority again: +0.8 % on the 16-page bench — and 2-8 pages/s instead of 180 for one-page
        // calls from 12 threads (detect latencies of seconds): its kernels starve as long as any other request has something queued.
        DeviceScope bind(device);
        int least = 0, greatest = 0;
        OCRS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        OCRS_HIP(hipStreamCreateWithPriority(&heavy_, hipStreamNonBlocking, greatest));
    }
    return heavy_;
}

hipStream_t DeviceContext::recurrent_stream(int mode) {
    if (mode == MODE_SERIAL) return heavy_stream();
    std::lock_guard<std::mutex> g(lazy_mu_);
    if (!recurrent_) {
        DeviceScope bind(device);
        int least = 0, greatest = 0;
        OCRS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        OCRS_HIP(hipStreamCreateWithPriority(&recurrent_, hipStreamNonBlocking, greatest));
    }
    return recurrent_;
}

// per host thread and device: leases the thread already holds (a nested lease must not wait for a switch that waits for it)
static thread_local int t_lease_depth[kMaxDevices] = {0};

// ---- isolation regime changes.  `change` runs with no request in flight on the device, new ones held back, the device idle.
template <class F> void DeviceContext::switch_isolation(F&& change) {
    if (t_lease_depth[device] > 0) fail(OCRS_ERR_INVALID_ARGUMENT, "the isolation regime of device %d cannot change from inside one of its requests", device);
    std::unique_lock<std::mutex> lk(iso_mu_);
    iso_cv_.wait(lk, [&] { return !switching_; });
    const Mode before = mode_locked();
    switching_ = true;                                   // from here on lease_begin() waits
    struct Done { DeviceContext* c; ~Done() { c->switching_ = false; c->iso_cv_.notify_all(); } } done{this};
    // what changes for a request is known only after `change`; a no-op costs no drain
    const Isolation p0 = policy_;
    const int r0 = relaxed_;
    change();
    if (mode_locked() == before) return;
    // put the old regime back while the requests that were told about it finish, then flip
    const Isolation p1 = policy_;
    const int r1 = relaxed_;
    policy_ = p0; relaxed_ = r0;
    iso_cv_.wait(lk, [&] { return leases_ == 0; });
    {
        DeviceScope bind(device);
        if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
    }
    policy_ = p1; relaxed_ = r1;
}

DeviceContext::Mode DeviceContext::lease_begin() {
    std::unique_lock<std::mutex> lk(iso_mu_);
    if (t_lease_depth[device] == 0) iso_cv_.wait(lk, [&] { return !switching_; });
    t_lease_depth[device]++;
    leases_++;
    return mode_locked();
}

void DeviceContext::lease_end() {
    std::lock_guard<std::mutex> lk(iso_mu_);
    t_lease_depth[device]--;
    if (--leases_ == 0) iso_cv_.notify_all();
}

void DeviceContext::add_relaxed_engine(int delta) {
    switch_isolation([&] { relaxed_ += delta; });
}

void DeviceContext::set_isolation(Isolation policy) {
    switch_isolation([&] { policy_ = policy; });
}

DeviceContext::Mode DeviceContext::current_mode() {
    std::lock_guard<std::mutex> lk(iso_mu_);
    return mode_locked();
}

int DeviceContext::relaxed_engine_count() {
    std::lock_guard<std::mutex> lk(iso_mu_);
    return relaxed_;
}

int DeviceContext::cu_count() {
    std::lock_guard<std::mutex> g(lazy_mu_);
    if (!cus_) {
        hipDeviceProp_t prop;
        OCRS_HIP(hipGetDeviceProperties(&prop, device));
        cus_ = prop.multiProcessorCount;
    }
    return cus_;
}

// ---------------------------------------------------------------- options
namespace {
struct OptDef { const char* name; const char* env; long def; };
const OptDef kOptDefs[OPT_COUNT] = {
    {"gru_mode", "OCRS_GRU_MODE", GRU_PERSISTENT},      // 0 persistent recurrence kernel, 1 one launch per time step
    {"gru_gates", "OCRS_GRU_GATES", 1},                 // persistent GRU: gate-per-wave kernel when every row tile gets its own cluster
    {"gru_local", "OCRS_GRU_LOCAL", 1},                 // persistent GRU: 1 same-XCD clusters hand off through L2, 0 always write-through
    {"det_fuse", "OCRS_DET_FUSE", 1},                   // fused DoubleConv blocks: 1 where they win, 2 every shape, 0 none
    {"det_mfma", "OCRS_DET_MFMA", 1},                   // fused detection blocks: pointwise convs + ConvTranspose on MFMA (1); VALU kernels (0)
    {"det_stream", "OCRS_DET_STREAM", 1},               // DoubleConv blocks of the full-resolution levels: row-streaming wave kernels (1; 8 / 14 / 32 rows per wave) or LDS-tiled (0)
    {"det_rows", "OCRS_DET_ROWS", 1},                   // DoubleConv blocks of the 16-64-channel levels: row-streaming workgroup kernels (1; 8 / 14 / 20 / 32 rows) or LDS-tiled (0)
    {"ccl_quad", "OCRS_CCL_QUAD", 1},                   // component labelling / root compaction: four pixels per thread on word-aligned masks (1) or one (0)
    {"conv12_fuse", "OCRS_CONV12_FUSE", 1},             // first two recognition convs (+ their pools) in one kernel
    {"conv_flat", "OCRS_CONV_FLAT", 1},                 // recognition 3x3 convs: patches tile a width group's whole strip of images (0: every image on its own)
    {"beam_gpu", "OCRS_BEAM_GPU", 1},                   // 1 CTC beam search on the GPU, 0 on the host
    // not options: ocrs_engine_params fields (no name, no environment variable)
    // (ocrs_engine_set_option accepts these names too, except numerics: an engine's numerics are fixed when it is created)
    {"numerics", nullptr, 0},                           // exact
    {"coalesce", nullptr, 2},                           // merged batches of small requests in flight per engine and stage (0 = no merging)
    {"coalesce_pages", nullptr, 16},                    // pages per merged batch
    {"coalesce_window_us", nullptr, 300},
    {"layout_threads", nullptr, 0},                     // host threads of find_text_lines_batch (0 = automatic)
    {"rec_max_pixels", nullptr, 0},                     // input pixels per recognition sub-request (0 = 2e9, the memory budget)
};
// accepted values per entry (the kernels index tables with some of these): {lo, hi} and, for the row-count selectors, the
// allowed set beyond 0 / 1
struct OptRange { long lo, hi; long also[4]; };
const OptRange kOptRanges[OPT_COUNT] = {
    {0, 1, {}}, {0, 1, {}}, {0, 1, {}}, {0, 2, {}}, {0, 2, {}}, {0, 1, {8, 14, 32}}, {0, 1, {8, 14, 20, 32}}, {0, 1, {}}, {0, 1, {}}, {0, 1, {}}, {0, 1, {}},
    {0, 2, {}}, {0, 64, {}}, {1, 4096, {}}, {0, 10000000, {}}, {0, 4096, {}}, {0, INT64_MAX, {}},
};
bool in_range(int i, long v) {
    const OptRange& r = kOptRanges[i];
    if (v >= r.lo && v <= r.hi) return true;
    for (long a : r.also) if (a && v == a) return true;
    return false;
}
// options of rounds 2-4 that round 5 removed with their kernels: still accepted by ocrs_set_option as no-ops, so that a caller
// (or a launch script) written against the older header keeps working; their OCRS_* environment variables are ignored
const char* const kRetired[] = {"det_heavy", "det_tail", "gru_waves", "gru_background", "gru_scatter", "gru_gates_pack", "conv_occupancy",
                                "gx_heavy", "gemm_nfast", "conv_a_lds", "gru_heavy", "heavy_priority"};
std::atomic<long> g_opts[OPT_COUNT];
std::once_flag g_opts_once;
void init_options() {   // the only getenv of the option system: once per process
    for (int i = 0; i < OPT_COUNT; i++) {
        const char* e = kOptDefs[i].env ? getenv(kOptDefs[i].env) : nullptr;
        const long v = e && *e ? strtol(e, nullptr, 10) : kOptDefs[i].def;
        g_opts[i].store(in_range(i, v) ? v : kOptDefs[i].def);
    }
}
int find_option(const char* name, int count = OPT_PUBLIC_COUNT) {
    for (int i = 0; i < count; i++)
        if (name && i != OPT_NUMERICS && strcmp(name, kOptDefs[i].name) == 0) return i;
    return -1;
}
thread_local const Tuning* t_tuning = nullptr;
}  // namespace

TuningScope::TuningScope(const Tuning* t) : prev_(t_tuning) { t_tuning = t; }
TuningScope::~TuningScope() { t_tuning = prev_; }
const Tuning* current_tuning() { return t_tuning; }

long option_long(Option o) {
    // relaxed atomic accesses: ocrs_engine_set_option may run while another thread serves a request of the same engine
    if (t_tuning) return __atomic_load_n(&t_tuning->v[o], __ATOMIC_RELAXED);
    std::call_once(g_opts_once, init_options);
    return g_opts[o].load(std::memory_order_relaxed);
}
int option(Option o) { return (int)option_long(o); }

Tuning default_tuning() {
    std::call_once(g_opts_once, init_options);
    Tuning t;
    for (int i = 0; i < OPT_COUNT; i++) t.v[i] = g_opts[i].load(std::memory_order_relaxed);
    return t;
}

const char* option_name(int i) { return i >= 0 && i < OPT_PUBLIC_COUNT ? kOptDefs[i].name : nullptr; }

int set_option(const char* name, long value) {
    std::call_once(g_opts_once, init_options);
    const int i = find_option(name);
    if (i < 0) {
        for (const char* r : kRetired) if (name && strcmp(name, r) == 0) return 0;
        return 1;
    }
    if (!in_range(i, value)) return 2;
    g_opts[i].store(value);
    return 0;
}

int set_option(Tuning& t, const char* name, long value) {
    const int i = find_option(name, OPT_COUNT);   // an engine's copy: also the configuration fields, by their field names
    if (i < 0) return 1;
    if (!in_range(i, value)) return 2;
    __atomic_store_n(&t.v[i], value, __ATOMIC_RELAXED);
    return 0;
}

bool get_option(const Tuning& t, const char* name, long* value) {
    // "numerics" is listed by neither ocrs_option_name nor set_option (fixed at creation) but can be read
    const int i = name && strcmp(name, "numerics") == 0 ? (int)OPT_NUMERICS : find_option(name, OPT_COUNT);
    if (i < 0) return false;
    *value = __atomic_load_n(&t.v[i], __ATOMIC_RELAXED);
    return true;
}

// ---------------------------------------------------------------- streams
StreamLease::StreamLease(bool high_priority) : ctx_(&ctx()), high_(high_priority) {
    high_ = true;  // every request stream outranks nothing and is outranked by nothing: all at the highest priority (see heavy_stream())
    mode_ = ctx_->lease_begin();
    try {
        if (mode_ == DeviceContext::MODE_SERIAL) {   // one stream for everything on this device; the lease owns only its event
            s_ = ctx_->heavy_stream();
            shared_ = true;
            OCRS_HIP(hipEventCreateWithFlags(&done_, hipEventBlockingSync | hipEventDisableTiming));
            return;
        }
        {
            std::lock_guard<std::mutex> g(ctx_->stream_mu);
            auto& v = ctx_->streams;
            if (!v.empty()) {
                s_ = v.back().first;
                done_ = v.back().second;
                v.pop_back();
                return;
            }
        }
        int least = 0, greatest = 0;
        OCRS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        OCRS_HIP(hipStreamCreateWithPriority(&s_, hipStreamNonBlocking, greatest));
        OCRS_HIP(hipEventCreateWithFlags(&done_, hipEventBlockingSync | hipEventDisableTiming));
    } catch (...) {
        ctx_->lease_end();
        throw;
    }
}

StreamLease::~StreamLease() {
    if (shared_) {
        (void)hipEventDestroy(done_);
    } else {
        std::lock_guard<std::mutex> g(ctx_->stream_mu);
        ctx_->streams.emplace_back(s_, done_);
    }
    ctx_->lease_end();
}

// ---------------------------------------------------------------- timers
std::vector<StageTimers::Pending>& StageTimers::pending() {
    static thread_local std::vector<Pending> p;
    return p;
}
std::map<int, std::vector<hipEvent_t>>& StageTimers::thread_events() {
    // events belong to the device they were created on: one free list per (host thread, device)
    static thread_local std::map<int, std::vector<hipEvent_t>> f;
    return f;
}
std::vector<hipEvent_t>& StageTimers::free_events() { return thread_events()[ctx().device]; }

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
    Pending p{stage, get_event(), get_event(), n_launches, false, 0.0, 0.0, 0.0};
    OCRS_HIP(hipEventRecord(p.a, s));
    pending().push_back(p);
    return (int)pending().size() - 1;
}

int StageTimers::kbegin(int cls, hipStream_t s, double flops, double bytes, double mfma_flops) {
    if (!enabled || !kernels_enabled || !((kernel_mask >> cls) & 1u)) return -1;
    if (mfma_flops < 0.0) mfma_flops = cls <= KC_GEMM_LINEAR ? flops : 0.0;   // the gemm_*_mfma classes
    Pending p{cls, get_event(), get_event(), 1, true, flops, bytes, mfma_flops};
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
    double lms[ST_COUNT] = {0}, lkms[KC_COUNT] = {0}, lkf[KC_COUNT] = {0}, lkb[KC_COUNT] = {0}, lkm[KC_COUNT] = {0};
    uint64_t ln[ST_COUNT] = {0}, lkn[KC_COUNT] = {0};
    for (auto& p : pd) {
        float t = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) {
            if (p.kernel) { lkms[p.stage] += t; lkn[p.stage] += 1; lkf[p.stage] += p.flops; lkb[p.stage] += p.bytes; lkm[p.stage] += p.mfma; }
            else { lms[p.stage] += t; ln[p.stage] += p.n; }
        }
        free_events().push_back(p.a);
        free_events().push_back(p.b);
    }
    pd.clear();
    std::lock_guard<std::mutex> g(mu);
    for (int i = 0; i < ST_COUNT; i++) { ms[i] += lms[i]; launches[i] += ln[i]; }
    for (int i = 0; i < KC_COUNT; i++) { kms[i] += lkms[i]; klaunches[i] += lkn[i]; kflops[i] += lkf[i]; kbytes[i] += lkb[i]; kmfma[i] += lkm[i]; }
}

void StageTimers::release_thread_events() {
    for (auto& p : pending()) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    pending().clear();
    for (auto& kv : thread_events()) {
        for (hipEvent_t e : kv.second) (void)hipEventDestroy(e);
        kv.second.clear();
    }
}

void StageTimers::reset() {
    std::lock_guard<std::mutex> g(mu);
    for (int i = 0; i < ST_COUNT; i++) { ms[i] = 0; launches[i] = 0; }
    for (int i = 0; i < KC_COUNT; i++) { kms[i] = 0; klaunches[i] = 0; kflops[i] = 0; kbytes[i] = 0; kmfma[i] = 0; }
}

}  // namespace ocrs
