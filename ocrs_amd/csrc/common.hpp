// Shared host-side plumbing for libocrs_amd: error reporting, HIP checks,
// a caching device allocator and per-call streams.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ocrs_amd.h"

namespace ocrs {

// Error carried up to the C ABI, where it becomes (status, thread-local message).
struct Error : std::runtime_error {
    ocrs_status status;
    Error(ocrs_status s, const std::string& msg) : std::runtime_error(msg), status(s) {}
};

[[noreturn]] inline void fail(ocrs_status s, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(s, buf);
}

#define OCRS_HIP(expr)                                                                              \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            ::ocrs::fail(OCRS_ERR_DEVICE, "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, \
                         __LINE__, #expr);                                                          \
    } while (0)

void set_last_error(const std::string& msg);

// ---------------------------------------------------------------------------------------------------------
// Devices.  Every handle (model, engine, page, engine group member) belongs to ONE HIP device; every API entry
// binds the calling host thread to its handle's device for the duration of the call (DeviceScope) — HIP's current
// device is per host thread.  One process can therefore drive several GPUs (ocrs_engine_group_*), or one GPU per
// process (ocrs_set_device = the default device of handles created without an explicit one).
// Everything that used to be a process singleton — the caching allocators, the stream pool, the shared conv-stack
// and recurrence streams with the mutexes that order submissions to them — lives in the device's DeviceContext.
// ---------------------------------------------------------------------------------------------------------

// Size-bucketed caching allocator: hipMalloc is far too slow to sit on the
// per-page path, and stages need scratch whose size depends on the page.
// Cached (free) bytes are bounded: above the cap (default: a quarter of the device's memory; OCRS_POOL_CAP_GB read once
// at first use, ocrs_device_pool_configure) the largest cached blocks are handed to the process's trimmer thread, which
// returns them to the driver — hipFree waits for the device, so it never runs on a request's thread.
struct PoolStats { uint64_t live = 0, cached = 0, cap = 0, peak_live = 0, driver_allocs = 0, driver_frees = 0; };
class DevicePool {
  public:
    explicit DevicePool(int device) : device_(device) {}
    ~DevicePool();
    void* alloc(size_t bytes);     // the calling thread must be bound to this pool's device
    void release(void* p);         // any thread
    void trim();                   // returns every cached block to the driver, on the calling thread
    void set_cap(uint64_t bytes);
    PoolStats stats();
    int device() const { return device_; }

  private:
    uint64_t cap_locked();
    const int device_;
    std::mutex mu_;
    std::multimap<size_t, void*> free_;
    std::map<void*, size_t> live_;
    uint64_t cached_ = 0, live_bytes_ = 0, peak_live_ = 0, cap_ = 0, allocs_ = 0;
    std::atomic<uint64_t> frees_{0};   // blocks the driver has back (counted by whoever called hipFree, after it returned)
};

// Pinned host staging (hipHostMalloc), cached in the same size buckets with the same reuse rule (smallest cached block
// that wastes at most a quarter) and a cap on the cached bytes (default 1 GiB per device context).  Device-to-host
// results go through it: a hipMemcpyAsync into PAGEABLE memory does not return until the stream has reached and
// finished the copy, and the calling thread spins for all of that time — tens of milliseconds per request when
// the copy is queued behind the request's own kernels.
class HostPool {
  public:
    void* alloc(size_t bytes);
    void release(void* p);
    void set_cap(uint64_t bytes);
    PoolStats stats();

  private:
    std::mutex mu_;
    std::multimap<size_t, void*> free_;
    std::map<void*, size_t> live_;
    uint64_t cached_ = 0, live_bytes_ = 0, peak_live_ = 0, cap_ = uint64_t(1) << 30, allocs_ = 0;
    std::atomic<uint64_t> frees_{0};
};

struct DeviceContext {
    const int device;
    DevicePool pool;
    HostPool host_pool;
    // One stream per device for the throughput-bound conv stacks of all in-flight requests (model.cpp), and one
    // for the persistent GRU recurrences (kernels_gru.hip): their workgroups wait on each other, so they are
    // serialised on the device; both at the highest queue priority.  The *_phase mutexes keep "record event, make
    // the shared stream wait, launch, record, make the request stream wait" atomic per request.
    std::mutex heavy_phase, rec_phase;
    hipStream_t heavy_stream();
    hipStream_t recurrent_stream(int mode = 0);   // of a request in Mode `mode`: MODE_SERIAL -> the device's one stream
    // ---- isolation of the bf16-MFMA kernels (numerics != exact).  Round 5 saw another request's line crops change while the
    // split conv / GEMM kernels ran beside them; round 6 reproduced it stand-alone (tools/hazard_repro.hip: a dense bf16-MFMA
    // kernel whose matrix instructions source their operands from VGPRs corrupts 16-lane pieces of OTHER waves on its
    // compute unit; DESIGN.md §4.4 "Concurrency").  A device that has an engine with numerics != exact therefore does not let
    // kernels of different requests overlap.  The POLICY is per device (ocrs_device_set_isolation):
    //   ISO_AUTO (default)  one stream for every call on the device while such an engine exists — no two kernels at once;
    //   ISO_NONE            never (diagnostics only: tools/hazard_canary.py reproduces the hazard with it).
    // What a REQUEST does is decided once, when its StreamLease is made (StreamLease::mode()), and travels with it: stream,
    // recurrence stream and activation arena all follow that one decision.  A change of what new requests would be told (the
    // first such engine created, the last destroyed, a new policy) waits until no request is in flight on the device and the
    // device is idle (switch_isolation): requests of the two regimes never run side by side.
    enum Isolation { ISO_AUTO = 0, ISO_NONE = 1 };
    enum Mode { MODE_FREE = 0, MODE_SERIAL = 1 };
    Mode lease_begin();                        // a request starts: waits while a switch is pending; returns what applies to it
    void lease_end();
    void add_relaxed_engine(int delta);        // +1 / -1; drains the device when the regime changes
    void set_isolation(Isolation policy);
    Mode current_mode();                       // what a request starting now would be told (stats, tests)
    int relaxed_engine_count();
    int cu_count();
    // intermediate activations of the conv stacks that run on heavy_stream(): shared by all requests (stream order is
    // the exclusion), guarded by heavy_phase; see HipModel::run_prefix_ragged
    std::vector<struct DevBuf> heavy_arena;
    // recycled per-call streams (StreamLease)
    std::mutex stream_mu;
    std::vector<std::pair<hipStream_t, hipEvent_t>> streams;

    explicit DeviceContext(int d) : device(d), pool(d) {}

  private:
    Mode mode_locked() const { return policy_ == ISO_NONE || relaxed_ <= 0 ? MODE_FREE : MODE_SERIAL; }
    template <class F> void switch_isolation(F&& change);
    std::mutex lazy_mu_;
    hipStream_t heavy_ = nullptr, recurrent_ = nullptr;
    int cus_ = 0;
    std::mutex iso_mu_;
    std::condition_variable iso_cv_;
    int leases_ = 0, relaxed_ = 0;
    bool switching_ = false;
    Isolation policy_ = ISO_AUTO;
    friend class StreamLease;
};

// Registry (contexts are created on first use and live for the process).
DeviceContext& device_context(int device);
// Default device of the process: ocrs_set_device(), initially 0.
int default_device();
void select_device(int device);
// The context the calling thread is bound to (the default device's if no DeviceScope is active).
DeviceContext& ctx();

// Binds the calling thread to `device` (< 0: the default device) for the lifetime of the object; nests.
class DeviceScope {
  public:
    explicit DeviceScope(int device);
    ~DeviceScope();
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;

  private:
    DeviceContext* prev_;
    int restore_ = -1;   // outermost scope: the HIP device the thread was on before the call (-1: unknown)
};

inline DevicePool& pool() { return ctx().pool; }
inline HostPool& host_pool() { return ctx().host_pool; }
inline hipStream_t heavy_stream() { return ctx().heavy_stream(); }

// Tuning options (include/ocrs_amd.h, ocrs_set_option / ocrs_engine_set_option): integer-valued, looked up by name.
// The process holds the DEFAULTS (initial value from the environment variable OCRS_<NAME>, read once at first use;
// ocrs_set_option changes them); an engine copies the defaults when it is created and keeps its own copy (Tuning),
// which ocrs_engine_set_option changes — two engines in one process can differ, and changing a default never
// affects an engine that already exists.  Every engine entry point installs its engine's copy for the calling
// thread (TuningScope); code outside an engine call (ocrs_model_run on a bare model) sees the process defaults.
// The entries after OPT_PUBLIC_COUNT are not options: they are fields of ocrs_engine_params that travel the same way.
enum Option { OPT_GRU_MODE = 0, OPT_GRU_GATES, OPT_GRU_LOCAL, OPT_DET_FUSE, OPT_DET_MFMA, OPT_DET_STREAM, OPT_DET_ROWS, OPT_CCL_QUAD,
              OPT_CONV12_FUSE, OPT_CONV_FLAT, OPT_BEAM_GPU, OPT_PUBLIC_COUNT,
              OPT_NUMERICS = OPT_PUBLIC_COUNT,   // ocrs_engine_params.numerics: 0 exact (the numeric spec), 1 relaxed, 2 reduced
              OPT_COALESCE, OPT_COALESCE_PAGES, OPT_COALESCE_WINDOW_US,   // ocrs_engine_params.coalesce*
              OPT_LAYOUT_THREADS, OPT_REC_MAX_PIXELS,                     // ocrs_engine_params.layout_threads / rec_max_pixels
              OPT_COUNT };
struct Tuning { long v[OPT_COUNT]; };
Tuning default_tuning();                                   // the process defaults as they are now
// 0 = set (or a retired name: accepted, ignored), 1 = unknown name, 2 = value outside the option's range
int set_option(const char* name, long value);             // process default
int set_option(Tuning& t, const char* name, long value);  // one engine's copy (safe while the engine serves requests)
bool get_option(const Tuning& t, const char* name, long* value);
const char* option_name(int i);                            // i < OPT_PUBLIC_COUNT
class TuningScope {   // installs `t` for the calling thread for the lifetime of the object; nests
  public:
    explicit TuningScope(const Tuning* t);
    ~TuningScope();
    TuningScope(const TuningScope&) = delete;
    TuningScope& operator=(const TuningScope&) = delete;
  private:
    const Tuning* prev_;
};
const Tuning* current_tuning();   // what the calling thread has installed (nullptr: none) — handed to worker threads
enum { GRU_PERSISTENT = 0, GRU_STEP = 1 };
int option(Option o);

// Kernels that need more than 64 KB of dynamic LDS have to opt in with hipFuncSetAttribute — per DEVICE (the function
// object belongs to the device that is current): once per (kernel, device), remembered in `done` (bit = device ordinal
// mod 64).  A process that drives several GPUs (engine group) launches the same kernel on each of them.
inline void allow_dynamic_lds(const void* kernel, std::atomic<uint64_t>& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_relaxed) & bit) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done.fetch_or(bit, std::memory_order_relaxed);
}
long option_long(Option o);
inline int gru_mode() { return option(OPT_GRU_MODE); }

// RAII device buffer from the pool of the device the allocating thread is bound to; remembers its pool, so it may
// be released from any thread (e.g. ocrs_page_free on a thread bound to another device).
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevicePool* owner = nullptr;
    DevBuf() = default;
    explicit DevBuf(size_t n) : bytes(n), owner(n ? &pool() : nullptr) { p = n ? owner->alloc(n) : nullptr; }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), owner(o.owner) { o.p = nullptr; o.bytes = 0; o.owner = nullptr; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { reset(); p = o.p; bytes = o.bytes; owner = o.owner; o.p = nullptr; o.bytes = 0; o.owner = nullptr; }
        return *this;
    }
    ~DevBuf() { reset(); }
    void reset() { if (p && owner) owner->release(p); p = nullptr; bytes = 0; owner = nullptr; }
    int device() const { return owner ? owner->device() : -1; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// Pinned host staging buffer.
struct PinnedBuf {
    void* p = nullptr;
    size_t bytes = 0;
    PinnedBuf() = default;
    explicit PinnedBuf(size_t n) : bytes(n) { if (n) OCRS_HIP(hipHostMalloc(&p, n, hipHostMallocPortable)); }
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// One stream per API call, recycled: calls from different host threads run on
// different streams (the reference calls Model::run concurrently,
// recognition.rs:465-485).
class StreamLease {
  public:
    explicit StreamLease(bool high_priority = false);
    ~StreamLease();
    hipStream_t get() const { return s_; }
    // Host wait for everything enqueued so far.  hipStreamSynchronize / hipEventSynchronize spin a core for the
    // whole wait here (measured: every request thread at ~100 % CPU, also with hipEventBlockingSync), which with
    // several requests in flight per process and several processes per host costs more cores than the layout
    // analysis does.  So: record an event, poll it briefly, then sleep between polls (<= 0.1 ms added latency).
    hipError_t wait_done() const noexcept {
        hipError_t e = hipEventRecord(done_, s_);
        if (e != hipSuccess) return e;
        for (int i = 0;; i++) {
            e = hipEventQuery(done_);
            if (e != hipErrorNotReady) return e;
            if (i < 64) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(i < 256 ? 20 : 100));
        }
    }
    void sync() const { OCRS_HIP(wait_done()); }
    hipError_t sync_noexcept() const noexcept { return wait_done(); }
    // the isolation regime of this request (DeviceContext::Mode), decided when the lease was made
    DeviceContext::Mode mode() const { return mode_; }
    bool serial() const { return mode_ == DeviceContext::MODE_SERIAL; }
    // where this request's conv stacks go: the device's shared conv-stack stream
    hipStream_t conv_stream() const { return ctx_->heavy_stream(); }
    hipStream_t recurrent_stream() const { return ctx_->recurrent_stream((int)mode_); }

  private:
    DeviceContext* ctx_;   // the pool the stream goes back to
    DeviceContext::Mode mode_ = DeviceContext::MODE_FREE;
    bool shared_ = false;  // s_ is the device's one serial stream (MODE_SERIAL), not a leased one
    hipStream_t s_;
    hipEvent_t done_;
    bool high_;
};

// Stage timers (HIP events on the launching stream).
enum Stage {
    ST_PREPARE = 0,
    ST_RESIZE_IN,
    ST_DET_CNN,
    ST_RESIZE_THRESH,
    ST_CCL,
    ST_CONTOUR_RECTS,
    ST_LINE_CROP,
    ST_REC_CONV,
    ST_REC_GRU,
    ST_REC_HEAD,
    ST_CTC,
    ST_COUNT
};
extern const char* const kStageNames[ST_COUNT];

// Kernel classes of the model executor, timed individually (HIP events around each
// launch) when kernel timing is on; flops/bytes are ALGORITHMIC.
enum KernelClass {
    KC_GEMM_CONV3X3 = 0, KC_GEMM_POINTWISE, KC_GEMM_CONVT, KC_GEMM_GRU_INPUT, KC_GEMM_GRU_HIDDEN, KC_GEMM_LINEAR,
    KC_DWCONV3X3, KC_CONV_DIRECT, KC_POOL, KC_PADCAT, KC_CONV1X1_SIGMOID, KC_GRU_GATES, KC_LOGSOFTMAX_ARGMAX,
    KC_OTHER, KC_DET_BLOCK, KC_DET_STREAM_WAVE, KC_DET_STREAM_ROWS, KC_COUNT
};
extern const char* const kKernelClassNames[KC_COUNT];

struct StageTimers {
    bool enabled = false;
    bool kernels_enabled = false;
    uint32_t kernel_mask = 0xffffffffu;  // which KernelClass values get per-launch events
    std::mutex mu;
    double ms[ST_COUNT] = {0};
    uint64_t launches[ST_COUNT] = {0};
    double kms[KC_COUNT] = {0};
    uint64_t klaunches[KC_COUNT] = {0};
    double kflops[KC_COUNT] = {0};
    double kbytes[KC_COUNT] = {0};
    double kmfma[KC_COUNT] = {0};   // the part of kflops that ran on the matrix cores
    struct Pending { int stage; hipEvent_t a, b; uint64_t n; bool kernel; double flops, bytes, mfma; };
    // Pending events are per host thread (an API call runs on one thread and collects its own
    // events after draining its stream), so concurrent calls never wait on each other's events.
    static std::vector<Pending>& pending();
    int begin(int stage, hipStream_t s, uint64_t n_launches);  // returns token (-1 when disabled)
    // mfma_flops < 0: all of `flops` for the MFMA GEMM classes, none for the others
    int kbegin(int cls, hipStream_t s, double flops, double bytes, double mfma_flops = -1.0);
    void end(int token, hipStream_t s);
    void collect();  // after a stream sync
    void reset();
    // destroys the calling thread's cached timing events (a pooled worker thread calls it before it exits)
    static void release_thread_events();
  private:
    static std::vector<hipEvent_t>& free_events();
    static std::map<int, std::vector<hipEvent_t>>& thread_events();
    hipEvent_t get_event();
};

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

struct StageScope {
    StageTimers* t; int token; hipStream_t s;
    StageScope(StageTimers* timers, int stage, hipStream_t stream, uint64_t n_launches = 1)
        : t(timers), token(timers ? timers->begin(stage, stream, n_launches) : -1), s(stream) {}
    ~StageScope() { if (t && token >= 0) t->end(token, s); }
};

// Everything one API call launches: a leased stream plus the scratch buffers its
// kernels use.  The destructor drains the stream BEFORE the buffers go back to
// the pool, so no other call can be handed memory that is still in flight.
struct Workspace {
    StreamLease stream;
    HostPool& hpool = host_pool();             // of the device this call is bound to (fixed at construction)
    std::vector<DevBuf> bufs;
    std::vector<hipEvent_t> events;            // cross-stream dependencies created by this call
    hipEvent_t make_event() {
        hipEvent_t e;
        OCRS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        events.push_back(e);
        return e;
    }
    Workspace() = default;
    explicit Workspace(bool high_priority) : stream(high_priority) {}
    // copy `bytes` from a host temporary to the device without a sync: the bytes are parked in pinned staging
    // owned by the workspace (released once its stream has drained), so the copy is a real asynchronous DMA
    std::vector<void*> upload_staging;
    void upload(void* d_dst, const void* h_src, size_t bytes) {
        if (!bytes) return;
        void* pin = hpool.alloc(bytes);
        memcpy(pin, h_src, bytes);
        upload_staging.push_back(pin);
        OCRS_HIP(hipMemcpyAsync(d_dst, pin, bytes, hipMemcpyHostToDevice, stream.get()));
    }
    // device -> host without blocking the caller: staged in pinned memory on `st` (default: this workspace's
    // stream) and handed to `h_dst` by the next sync() of this workspace — which must cover `st`
    struct Download { void* pinned; void* dst; size_t bytes; };
    std::vector<Download> downloads;
    void download(void* h_dst, const void* d_src, size_t bytes, hipStream_t st = nullptr) {
        if (!bytes) return;
        void* pin = hpool.alloc(bytes);
        downloads.push_back(Download{pin, h_dst, bytes});
        OCRS_HIP(hipMemcpyAsync(pin, d_src, bytes, hipMemcpyDeviceToHost, st ? st : stream.get()));
    }
    // strided device -> host: `rows` pieces of `width` bytes, `d_pitch` apart on the device, packed on the host; handed
    // over like download()
    void download_2d(void* h_dst, const void* d_src, size_t d_pitch, size_t width, size_t rows, hipStream_t st = nullptr) {
        if (!width || !rows) return;
        void* pin = hpool.alloc(width * rows);
        downloads.push_back(Download{pin, h_dst, width * rows});
        OCRS_HIP(hipMemcpy2DAsync(pin, width, d_src, d_pitch, width, rows, hipMemcpyDeviceToHost, st ? st : stream.get()));
    }
    hipStream_t s() const { return stream.get(); }
    void* alloc(size_t bytes) { bufs.emplace_back(bytes ? bytes : 4); return bufs.back().p; }
    template <class T> T* alloc_n(size_t n) { return static_cast<T*>(alloc(n * sizeof(T))); }
    void finish_downloads(bool copy) {  // call only after the stream has drained
        for (const Download& d : downloads) {
            if (copy) memcpy(d.dst, d.pinned, d.bytes);
            hpool.release(d.pinned);
        }
        downloads.clear();
        for (void* p : upload_staging) hpool.release(p);
        upload_staging.clear();
    }
    void sync() {
        stream.sync();
        finish_downloads(true);
    }
    ~Workspace() {
        (void)stream.sync_noexcept();
        finish_downloads(false);
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
    }
};

const std::string& last_error();

}  // namespace ocrs
