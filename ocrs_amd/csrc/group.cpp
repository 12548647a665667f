// Several GPUs behind ONE handle in ONE process (include/ocrs_amd.h "engine group").
//
// The reference engine is immutable `&self` (ocrs/src/lib.rs:183-256) and pages are independent units
// (SURVEY.md §8e): an engine group holds one engine per member device (own weight replica, own stream / memory
// pools — DeviceContext), routes every page of a call to a member of the device the page lives on, runs the members'
// shares on the group's worker threads and merges the results in page order.  There is no collective on the compute
// path.
//
// Dealing.  A page is processed where it is resident.  Where the group places pages itself (host pixels), and
// among members that share a device, pages go out in CONTIGUOUS BLOCKS of at least `group_min_block` pages
// (`group_shared_block` between members of one device): a member handed two pages of a sixteen-page call runs its
// conv stacks and recurrences at a fraction of their batch efficiency, so a small call uses fewer members and
// successive calls start at successive members (round 3 dealt page i to member i mod G: two members on one GPU,
// 8 pages each, ran 12 % below one 16-page request).
//
// Result gather.  Inside one process every member's results reach the host through that member's own pinned staging
// and PCIe link, and the calling thread concatenates them: that is the transport of every per-request gather unless
// the group was created with OCRS_GATHER_RCCL.  RCCL is the transport of the FINAL result gather
// (ocrs_group_final_gather; north_star: "RCCL over xGMI only for the final result gather"): the members' packed
// bytes, prefixed by their length, go into a device buffer on each member, one grouped ncclAllGather
// (ncclCommInitAll communicator, one rank per member) moves them over xGMI and the root member's copy is read back.
// librccl is loaded lazily (dlopen) the first time a gather asks for it — a host-only group never touches it — and
// every failure on that path (library absent, communicator refused) resolves to the host transport with the reason
// kept for ocrs_group_last_gather; it is never fatal.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library itself is bound at run time (RcclApi)

#include <atomic>

#include "abi_util.hpp"
#include "host_pool.hpp"
#include "numa.hpp"
#include "kernels.hpp"

using namespace ocrs;
using namespace ocrs::abi;
using namespace ocrs::geom;

namespace {

// ------------------------------------------------------------------------------------------------ librccl, lazily
struct RcclApi {
    void* handle = nullptr;
    std::string path, error;
    bool accepts_duplicate_devices = false;   // only the test double (tests/stubs/rccl_stub.cpp) says so
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok() const { return handle != nullptr; }
};

// One binding per library path for the life of the process (OCRS_RCCL_LIB names another library — the tests'
// double; read when a group first needs RCCL, so a process may use both).
RcclApi& rccl_api() {
    static std::mutex mu;
    static std::map<std::string, std::unique_ptr<RcclApi>> apis;
    const char* env = getenv("OCRS_RCCL_LIB");
    const std::string want = env && *env ? env : "";
    std::lock_guard<std::mutex> lk(mu);
    auto it = apis.find(want);
    if (it != apis.end()) return *it->second;
    auto api = std::make_unique<RcclApi>();
    std::vector<std::string> candidates;
    if (!want.empty()) candidates = {want};
    else candidates = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const auto& c : candidates) {
        api->handle = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (api->handle) { api->path = c; break; }
        const char* e = dlerror();
        api->error = std::string("cannot load ") + c + ": " + (e ? e : "?");
    }
    if (api->handle) {
        auto sym = [&](const char* name) {
            void* p = dlsym(api->handle, name);
            if (!p && api->error.find("lacks") == std::string::npos) api->error = api->path + " lacks " + name;
            return p;
        };
        api->error.clear();
        api->CommInitAll = reinterpret_cast<decltype(api->CommInitAll)>(sym("ncclCommInitAll"));
        api->CommDestroy = reinterpret_cast<decltype(api->CommDestroy)>(sym("ncclCommDestroy"));
        api->GroupStart = reinterpret_cast<decltype(api->GroupStart)>(sym("ncclGroupStart"));
        api->GroupEnd = reinterpret_cast<decltype(api->GroupEnd)>(sym("ncclGroupEnd"));
        api->AllGather = reinterpret_cast<decltype(api->AllGather)>(sym("ncclAllGather"));
        api->GetErrorString = reinterpret_cast<decltype(api->GetErrorString)>(sym("ncclGetErrorString"));
        if (!api->error.empty()) {   // not an RCCL: unusable (kept loaded; the entry is per path and immutable)
            api->handle = nullptr;
        } else {
            api->accepts_duplicate_devices = dlsym(api->handle, "ocrs_rccl_stub_accepts_duplicate_devices") != nullptr;
        }
    }
    RcclApi& ref = *api;
    apis.emplace(want, std::move(api));
    return ref;
}

}  // namespace

struct ocrs_engine_group {
    struct Member {
        int device = 0;
        std::unique_ptr<ocrs_model> detection, recognition;
        std::unique_ptr<ocrs_engine> engine;
        // host placement (numa.hpp): the NUMA node the member's GPU hangs off and that node's CPUs; a share binds its thread
        // to them for its duration (its page-locked staging is then first touched there); empty = unknown, no binding
        int numa_node = -1;
        std::vector<int> cpus;
        // accounting for ocrs_group_member_stats (the first multi-GPU run must be diagnosable): shares run, pages they
        // carried, CPU time of the threads that ran them (CLOCK_THREAD_CPUTIME_ID deltas), their wall time, shares that ran bound
        std::atomic<uint64_t> shares{0}, pages{0}, cpu_ns{0}, wall_ns{0}, bound_shares{0};
        Member() = default;
        Member(Member&& o) noexcept : device(o.device), detection(std::move(o.detection)), recognition(std::move(o.recognition)),
                                      engine(std::move(o.engine)), numa_node(o.numa_node), cpus(std::move(o.cpus)) {}
    };
    std::vector<Member> members;
    std::vector<int> devices;                          // distinct devices, in order of first appearance
    std::vector<std::vector<size_t>> members_of;       // members_of[k]: members on devices[k], ascending
    ocrs_gather_mode gather = OCRS_GATHER_AUTO;        // transport of the per-request gathers (AUTO = host)
    bool distinct = true;

    // RCCL state, created by the first gather that asks for it
    std::mutex init_mu;
    bool rccl_tried = false, rccl_ready = false;
    RcclApi* api = nullptr;
    std::vector<ncclComm_t> comms;
    std::mutex issue_mu;   // collectives on one communicator are ENQUEUED by one thread at a time (not awaited under it)

    std::mutex info_mu;    // last_* and why_host
    int last_mode = 0;     // transport of the most recent gather: 1 host, 2 RCCL
    size_t last_bytes = 0;
    std::string why_host;  // why the most recent gather that wanted RCCL (or AUTO) used the host ("" if it did not)
    std::string rccl_fail_reason;   // why the communicators could not be created (init_mu); outlives later gathers
    std::string why_host_out;   // stable copy handed out by ocrs_group_last_gather

    // ---- host-side pre-flight (ocrs_group_set_replay; test hook, SURVEY §8e "expected scaling limiter: the host").
    // mode 1 (record): the members run as usual and the group keeps every page's word rects and recognised lines, keyed by the
    // host pixels the page was prepared from.  mode 2 (replay): a member's share does NO GPU work — it sleeps the configured
    // time of its stage (what a share of that stage takes on one GPU at full load) and returns the recorded results of ITS
    // pages — while everything on the host stays real: dealing, worker threads and their NUMA binding, payload packing,
    // per-request and final gathers, reassembly in page order, find_text_lines_batch on the real rects, the caller's loop.
    // N members on ONE physical GPU then behave, for the host, like N GPUs each at the single-GPU page rate.
    int replay_mode = 0;
    double replay_s[3] = {0, 0, 0};   // prepare, detect, recognize share
    std::mutex replay_mu;
    std::map<const void*, std::vector<RotatedRect>> replay_rects;
    std::map<const void*, std::vector<std::vector<ocrs_text_char>>> replay_lines;

    std::atomic<size_t> next_start{0};   // rotation of the block deal
    size_t min_block = 8, shared_block = 16;   // ocrs_group_params.min_block / shared_block
    WorkerPool workers{[] { StageTimers::release_thread_events(); }};   // the members' shares of a call run here

    ~ocrs_engine_group() {
        if (api && api->ok())
            for (size_t i = 0; i < comms.size(); i++)
                if (comms[i]) (void)api->CommDestroy(comms[i]);
    }
    size_t size() const { return members.size(); }
};

namespace {

// ------------------------------------------------------------------------------------------------ dealing
size_t block_size(size_t n_items, size_t n_takers, size_t min_block) {
    if (n_items == 0 || n_takers == 0) return 1;
    const size_t even = (n_items + n_takers - 1) / n_takers;
    return std::max(even, std::min(std::max<size_t>(min_block, 1), n_items));
}

// items[0..n) (in order) -> takers, contiguous blocks, the first block to taker `start`
void deal_blocks(const std::vector<size_t>& items, const std::vector<size_t>& takers, size_t min_block, size_t start,
                 std::vector<std::vector<size_t>>* of_taker) {
    if (items.empty()) return;
    const size_t b = block_size(items.size(), takers.size(), min_block);
    for (size_t j = 0; j < items.size(); j++) (*of_taker)[takers[(start + j / b) % takers.size()]].push_back(items[j]);
}

// Pages that live on devices (device_of_page[i], an index into g->devices): each device's pages to its members.
std::vector<std::vector<size_t>> deal_resident(ocrs_engine_group* g, const std::vector<size_t>& device_of_page) {
    std::vector<std::vector<size_t>> of_member(g->size());
    std::vector<std::vector<size_t>> on_device(g->devices.size());
    for (size_t i = 0; i < device_of_page.size(); i++) on_device[device_of_page[i]].push_back(i);
    for (size_t k = 0; k < g->devices.size(); k++) {
        const auto& mem = g->members_of[k];
        if (on_device[k].empty()) continue;
        if (mem.size() == 1) {
            of_member[mem[0]] = on_device[k];
        } else {
            const size_t b = block_size(on_device[k].size(), mem.size(), g->shared_block);
            const size_t used = (on_device[k].size() + b - 1) / b;
            const size_t start = used < mem.size() ? g->next_start.fetch_add(used) : 0;
            deal_blocks(on_device[k], mem, g->shared_block, start, &of_member);
        }
    }
    return of_member;
}

// Pages the group places itself: contiguous blocks over the distinct devices, then as above.
std::vector<std::vector<size_t>> deal_free(ocrs_engine_group* g, size_t n) {
    const size_t D = g->devices.size();
    std::vector<size_t> items(n), takers(D);
    for (size_t i = 0; i < n; i++) items[i] = i;
    for (size_t k = 0; k < D; k++) takers[k] = k;
    const size_t b = block_size(n, D, g->min_block);
    const size_t used = (n + b - 1) / b;
    const size_t start = used < D ? g->next_start.fetch_add(used) : 0;
    std::vector<std::vector<size_t>> on_device(D);
    deal_blocks(items, takers, g->min_block, start, &on_device);
    std::vector<size_t> device_of_page(n, 0);
    for (size_t k = 0; k < D; k++)
        for (size_t i : on_device[k]) device_of_page[i] = k;
    return deal_resident(g, device_of_page);
}

size_t device_slot(const ocrs_engine_group* g, int device) {
    for (size_t k = 0; k < g->devices.size(); k++)
        if (g->devices[k] == device) return k;
    return g->devices.size();
}

std::vector<std::vector<size_t>> deal_pages(ocrs_engine_group* g, const ocrs_page* const* pages, size_t n) {
    std::vector<size_t> dev(n);
    for (size_t i = 0; i < n; i++) {
        if (!pages[i]) fail(OCRS_ERR_INVALID_ARGUMENT, "null page");
        dev[i] = device_slot(g, pages[i]->device());
        if (dev[i] == g->devices.size())
            fail(OCRS_ERR_INVALID_ARGUMENT, "page %zu lives on device %d, which has no member in this group", i, pages[i]->device());
    }
    return deal_resident(g, dev);
}

// Runs fn(m) for every member with work, each on a worker thread bound to its member's device (the first share on
// the calling thread); the first failure (lowest member) is rethrown after all have finished.
template <class Fn>
void for_each_member(ocrs_engine_group* g, const std::vector<std::vector<size_t>>& of_member, Fn&& fn) {
    std::vector<char> has_work(g->size(), 0);
    for (size_t m = 0; m < g->size(); m++) has_work[m] = !of_member[m].empty();
    std::vector<std::exception_ptr> errs;
    run_shares(g->workers, has_work, errs, [&](size_t m) {
        auto& mem = g->members[m];
        numa::BindScope place(mem.cpus);
        timespec c0{}, c1{};
        (void)clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c0);
        const auto w0 = std::chrono::steady_clock::now();
        struct Account {   // also when the share throws
            ocrs_engine_group::Member& mem; timespec& c0; timespec& c1; std::chrono::steady_clock::time_point w0; size_t n; bool bound;
            ~Account() {
                (void)clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c1);
                mem.shares++;
                mem.pages += n;
                mem.bound_shares += bound ? 1 : 0;
                mem.cpu_ns += (uint64_t)((c1.tv_sec - c0.tv_sec) * 1000000000ll + (c1.tv_nsec - c0.tv_nsec));
                mem.wall_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count();
            }
        } account{mem, c0, c1, w0, of_member[m].size(), place.bound()};
        DeviceScope bind(mem.device);
        TuningScope tune(&mem.engine->tuning);   // the member engine's own options, on whichever thread runs its share
        fn(m);
    });
}

// ------------------------------------------------------------------------------------------------ gather
// payloads[m] (host bytes) -> one buffer in member order.  offsets: G + 1 entries.
std::vector<uint8_t> gather_host(const std::vector<std::vector<uint8_t>>& payloads, std::vector<size_t>* offsets) {
    std::vector<uint8_t> out;
    offsets->assign(1, 0);
    for (const auto& p : payloads) {
        out.insert(out.end(), p.begin(), p.end());
        offsets->push_back(out.size());
    }
    return out;
}

// Communicators on first use.  false: RCCL cannot serve this group; *why says so.
bool ensure_comms(ocrs_engine_group* g, std::string* why) {
    std::lock_guard<std::mutex> lk(g->init_mu);
    if (g->rccl_tried) {
        if (!g->rccl_ready && why) *why = g->rccl_fail_reason;   // (not why_host: every later gather overwrites that)
        return g->rccl_ready;
    }
    g->rccl_tried = true;
    std::string reason;
    RcclApi& api = rccl_api();
    g->api = &api;
    if (!api.ok()) {
        reason = "librccl unavailable (" + api.error + ")";
    } else if (!g->distinct && !api.accepts_duplicate_devices) {
        reason = "a device appears more than once in the group: RCCL refuses such a communicator";
    } else {
        std::vector<int> devs;
        for (const auto& mem : g->members) devs.push_back(mem.device);
        g->comms.assign(devs.size(), nullptr);
        // RCCL checks hipGetLastError() after its own HIP calls: an error word this thread's earlier HIP calls left behind must not
        // come back from it as "unhandled cuda error" (defensive: every HIP call of this library checks its own return value).
        (void)hipGetLastError();
        const ncclResult_t r = api.CommInitAll(g->comms.data(), (int)devs.size(), devs.data());
        if (r != ncclSuccess) {
            reason = std::string("ncclCommInitAll failed: ") + api.GetErrorString(r);
            (void)hipGetLastError();
            g->comms.clear();
        } else {
            g->rccl_ready = true;
        }
    }
    if (!g->rccl_ready) {
        g->rccl_fail_reason = reason;
        std::lock_guard<std::mutex> li(g->info_mu);
        g->why_host = reason;
        if (why) *why = reason;
    }
    return g->rccl_ready;
}

// Through the devices: member m's payload, prefixed by its length, is uploaded to member m's device on a stream of
// its own, one grouped ncclAllGather leaves every member with all G slots, and the root's copy is downloaded.
// Several gathers may be in flight: only the ENQUEUE of a collective is serialised (one communicator, one issue
// order); uploads, the transfers themselves and the read-back of different calls overlap.
std::vector<uint8_t> gather_rccl(ocrs_engine_group* g, const std::vector<std::vector<uint8_t>>& payloads,
                                 std::vector<size_t>* offsets) {
    const size_t G = g->size();
    RcclApi& api = *g->api;
    size_t cap = 0;
    for (const auto& p : payloads) cap = std::max(cap, p.size());
    const size_t slot = ((cap + sizeof(uint64_t) + 15) / 16) * 16;   // [u64 length | bytes | padding]
    std::vector<std::unique_ptr<Workspace>> ws(G);
    std::vector<uint8_t*> d_send(G), d_recv(G);
    std::vector<uint8_t> stage(slot);
    struct Cleanup {   // workspaces drain and return their buffers while bound to their own device
        ocrs_engine_group* g; std::vector<std::unique_ptr<Workspace>>& ws;
        ~Cleanup() {
            for (size_t m = 0; m < ws.size(); m++)
                if (ws[m]) {
                    try { DeviceScope bind(g->members[m].device); ws[m].reset(); } catch (...) { ws[m].release(); }
                }
        }
    } cleanup{g, ws};
    for (size_t m = 0; m < G; m++) {
        DeviceScope bind(g->members[m].device);
        ws[m] = std::make_unique<Workspace>();
        d_send[m] = ws[m]->alloc_n<uint8_t>(slot);
        d_recv[m] = ws[m]->alloc_n<uint8_t>(slot * G);
        const uint64_t len = payloads[m].size();
        std::fill(stage.begin(), stage.end(), 0);
        memcpy(stage.data(), &len, sizeof len);
        if (len) memcpy(stage.data() + sizeof len, payloads[m].data(), len);
        ws[m]->upload(d_send[m], stage.data(), slot);
    }
    {
        std::lock_guard<std::mutex> lk(g->issue_mu);
        (void)hipGetLastError();   // see ensure_comms
        ncclResult_t r = api.GroupStart();
        if (r != ncclSuccess) fail(OCRS_ERR_DEVICE, "RCCL error %s in ncclGroupStart", api.GetErrorString(r));
        for (size_t m = 0; m < G; m++) {
            r = api.AllGather(d_send[m], d_recv[m], slot, ncclUint8, g->comms[m], ws[m]->s());
            if (r != ncclSuccess) {
                (void)api.GroupEnd();
                fail(OCRS_ERR_DEVICE, "RCCL error %s in ncclAllGather (member %zu)", api.GetErrorString(r), m);
            }
        }
        r = api.GroupEnd();
        if (r != ncclSuccess) fail(OCRS_ERR_DEVICE, "RCCL error %s in ncclGroupEnd", api.GetErrorString(r));
    }
    std::vector<uint8_t> all(slot * G);
    {
        DeviceScope bind(g->members[0].device);
        ws[0]->download(all.data(), d_recv[0], all.size());
        ws[0]->sync();
    }
    for (size_t m = 1; m < G; m++) {   // the other members' collectives must have finished before their buffers are reused
        DeviceScope bind(g->members[m].device);
        ws[m]->sync();
    }
    std::vector<uint8_t> out;
    offsets->assign(1, 0);
    for (size_t m = 0; m < G; m++) {
        uint64_t len = 0;
        memcpy(&len, all.data() + m * slot, sizeof len);
        if (len != payloads[m].size())
            fail(OCRS_ERR_DEVICE, "result gather: member %zu's slot carries %llu bytes, %zu were sent", m, (unsigned long long)len,
                 payloads[m].size());
        out.insert(out.end(), all.begin() + m * slot + sizeof len, all.begin() + m * slot + sizeof len + len);
        offsets->push_back(out.size());
    }
    return out;
}

// mode: the transport asked for.  HOST: host.  RCCL: RCCL if it can be had.  AUTO (only the final gather passes it
// through; per-request gathers map AUTO to HOST before they get here): RCCL when there is something to move between
// devices (two or more members) and it can be had.
std::vector<uint8_t> gather(ocrs_engine_group* g, ocrs_gather_mode mode, const std::vector<std::vector<uint8_t>>& payloads,
                            std::vector<size_t>* offsets) {
    size_t total = 0;
    for (const auto& p : payloads) total += p.size();
    std::string why;
    bool use_rccl = false;
    if (mode == OCRS_GATHER_HOST) {
        why = "host transport requested";
    } else if (mode == OCRS_GATHER_AUTO && g->size() == 1) {
        why = "one member: nothing to gather";
    } else {
        use_rccl = ensure_comms(g, &why);
    }
    {
        std::lock_guard<std::mutex> li(g->info_mu);
        g->last_mode = use_rccl ? 2 : 1;
        g->last_bytes = total;
        g->why_host = use_rccl ? "" : why;
    }
    return use_rccl ? gather_rccl(g, payloads, offsets) : gather_host(payloads, offsets);
}

std::vector<uint8_t> gather_request(ocrs_engine_group* g, const std::vector<std::vector<uint8_t>>& payloads, std::vector<size_t>* offsets) {
    return gather(g, g->gather == OCRS_GATHER_RCCL ? OCRS_GATHER_RCCL : OCRS_GATHER_HOST, payloads, offsets);
}

template <class T>
void append_bytes(std::vector<uint8_t>& v, const T* p, size_t n) {
    const uint8_t* b = reinterpret_cast<const uint8_t*>(p);
    v.insert(v.end(), b, b + n * sizeof(T));
}

// Bounds-checked reader of one member's slice of the gathered bytes.
struct Cursor {
    const uint8_t* base; size_t at, end, member;
    uint64_t count() {
        uint64_t c = 0;
        need(sizeof c);
        memcpy(&c, base + at, sizeof c);
        at += sizeof c;
        return c;
    }
    const uint8_t* take(size_t bytes) {
        need(bytes);
        const uint8_t* p = base + at;
        at += bytes;
        return p;
    }
    void need(size_t bytes) const {
        if (bytes > end - at) fail(OCRS_ERR_DEVICE, "result gather: member %zu's payload is shorter than its contents claim", member);
    }
};

std::vector<Cursor> cursors(const std::vector<uint8_t>& all, const std::vector<size_t>& moffs) {
    std::vector<Cursor> c;
    for (size_t m = 0; m + 1 < moffs.size(); m++) c.push_back(Cursor{all.data(), moffs[m], moffs[m + 1], m});
    return c;
}

std::vector<size_t> member_of_page(const std::vector<std::vector<size_t>>& of_member, size_t n) {
    std::vector<size_t> mo(n, 0);
    for (size_t m = 0; m < of_member.size(); m++)
        for (size_t i : of_member[m]) mo[i] = m;
    return mo;
}

}  // namespace

extern "C" {

ocrs_status ocrs_engine_group_new(const ocrs_group_params* params, ocrs_engine_group** out) {
    return guarded([&] {
        if (!params || !out) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        if (!params->devices || params->n_devices == 0 || params->n_devices > 64)
            fail(OCRS_ERR_INVALID_ARGUMENT, "an engine group needs 1..64 member devices");
        if (params->gather != OCRS_GATHER_AUTO && params->gather != OCRS_GATHER_HOST && params->gather != OCRS_GATHER_RCCL)
            fail(OCRS_ERR_INVALID_ARGUMENT, "unknown gather mode %d", (int)params->gather);
        auto g = std::make_unique<ocrs_engine_group>();
        g->gather = params->gather;
        if (params->min_block > 0) g->min_block = (size_t)params->min_block;
        if (params->shared_block > 0) g->shared_block = (size_t)params->shared_block;
        g->members.resize(params->n_devices);
        for (size_t m = 0; m < params->n_devices; m++) {
            auto& mem = g->members[m];
            mem.device = params->devices[m];
            size_t k = device_slot(g.get(), mem.device);
            if (k == g->devices.size()) {
                g->devices.push_back(mem.device);
                g->members_of.emplace_back();
            } else {
                g->distinct = false;
            }
            g->members_of[k].push_back(m);
            // one weight replica per member (a few MB), on the member's device
            if (params->detection_model) {
                mem.detection = std::make_unique<ocrs_model>();
                mem.detection->impl = HipModel::load(params->detection_model, params->detection_model_len, mem.device);
            }
            if (params->recognition_model) {
                mem.recognition = std::make_unique<ocrs_model>();
                mem.recognition->impl = HipModel::load(params->recognition_model, params->recognition_model_len, mem.device);
            }
            ocrs_engine_params ep{};
            ep.detection_model = mem.detection.get();
            ep.recognition_model = mem.recognition.get();
            ep.debug = params->debug;
            ep.decode_method = params->decode_method;
            ep.beam_width = params->beam_width;
            ep.alphabet = params->alphabet;
            ep.allowed_chars = params->allowed_chars;
            ep.numerics = params->numerics;
            ep.coalesce = params->coalesce;
            ep.coalesce_pages = params->coalesce_pages;
            ep.coalesce_window_us = params->coalesce_window_us;
            ep.layout_threads = params->layout_threads;
            ep.rec_max_pixels = params->rec_max_pixels;
            mem.engine = make_engine(ep);
            mem.engine->device = mem.device;   // (an engine without weights has nothing else to pin it to its device)
            if (mem.engine->counted_relaxed && mem.engine->counted_relaxed != &device_context(mem.device))
                mem.engine->count_relaxed(device_context(mem.device));   // ... and is counted where it runs
            {   // where the member's host side should run: the NUMA node of its GPU (silent when the host does not say)
                char bus[64] = {0};
                if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, mem.device) == hipSuccess) {
                    mem.numa_node = numa::node_of_pci(bus);
                    if (!numa::cpus_of_node(mem.numa_node, &mem.cpus)) mem.cpus.clear();
                } else {
                    (void)hipGetLastError();
                }
            }
        }
        *out = g.release();
    });
}

void ocrs_engine_group_free(ocrs_engine_group* g) { delete g; }

ocrs_status ocrs_engine_group_size(const ocrs_engine_group* g, size_t* n) {
    return guarded([&] {
        if (!g || !n) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        *n = g->size();
    });
}

ocrs_status ocrs_engine_group_member(const ocrs_engine_group* g, size_t i, const ocrs_engine** engine, int* device) {
    return guarded([&] {
        if (!g || i >= g->size()) fail(OCRS_ERR_INVALID_ARGUMENT, "no such member");
        if (engine) *engine = g->members[i].engine.get();
        if (device) *device = g->members[i].device;
    });
}

ocrs_status ocrs_group_deal(size_t n_pages, size_t n_members, size_t min_block, size_t* member_of_page_out, size_t* pages_per_member) {
    return guarded([&] {
        if (n_members == 0 || (n_pages && !member_of_page_out)) fail(OCRS_ERR_INVALID_ARGUMENT, "bad argument");
        if (pages_per_member)
            for (size_t m = 0; m < n_members; m++) pages_per_member[m] = 0;
        const size_t b = block_size(n_pages, n_members, min_block ? min_block : 8);
        for (size_t i = 0; i < n_pages; i++) {
            const size_t m = (i / b) % n_members;
            member_of_page_out[i] = m;
            if (pages_per_member) pages_per_member[m]++;
        }
    });
}

ocrs_status ocrs_group_last_gather(const ocrs_engine_group* gc, int* transport, size_t* bytes, const char** why_host) {
    return guarded([&] {
        if (!gc) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        ocrs_engine_group* g = const_cast<ocrs_engine_group*>(gc);
        std::lock_guard<std::mutex> li(g->info_mu);
        if (transport) *transport = g->last_mode;
        if (bytes) *bytes = g->last_bytes;
        if (why_host) {
            g->why_host_out = g->why_host;
            *why_host = g->why_host_out.c_str();
        }
    });
}

static void group_prepare(ocrs_engine_group* g, const void* const* pixels, size_t n, bool on_device, ocrs_pixel_type type,
                          ocrs_dim_order order, int height, int width, int channels, ocrs_page** out) {
    if (!g || !out || (n > 0 && !pixels)) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
    for (size_t i = 0; i < n; i++) check_image_args(pixels[i], height, width, channels);
    std::vector<std::vector<size_t>> of_member;
    if (on_device) {   // a page is converted where its pixels are
        std::vector<size_t> dev(n);
        for (size_t i = 0; i < n; i++) {
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, pixels[i]) != hipSuccess) {
                (void)hipGetLastError();
                fail(OCRS_ERR_INVALID_ARGUMENT, "image %zu: not a device pointer", i);
            }
            dev[i] = device_slot(g, attr.device);
            if (dev[i] == g->devices.size())
                fail(OCRS_ERR_INVALID_ARGUMENT, "image %zu lives on device %d, which has no member in this group", i, attr.device);
        }
        of_member = deal_resident(g, dev);
    } else {
        of_member = deal_free(g, n);
    }
    const size_t bytes = (size_t)height * width * channels * (type == OCRS_U8 ? 1 : 4);
    std::vector<std::unique_ptr<ocrs_page>> made(n);   // freed if any member fails
    for_each_member(g, of_member, [&](size_t m) {
        const ocrs_engine* e = g->members[m].engine.get();
        if (g->replay_mode == 2) {   // no upload, no conversion: a page object of the right size on the member's device, after the share's time
            for (size_t i : of_member[m]) {
                auto page = std::make_unique<ocrs_page>();
                page->h = height; page->w = width; page->source = pixels[i];
                page->grey = DevBuf((size_t)height * width * sizeof(float));
                made[i] = std::move(page);
            }
            std::this_thread::sleep_for(std::chrono::duration<double>(g->replay_s[0]));
            return;
        }
        Workspace ws;
        for (size_t i : of_member[m]) {
            const void* d_px = pixels[i];
            if (!on_device) {
                void* d = ws.alloc(bytes);
                OCRS_HIP(hipMemcpyAsync(d, pixels[i], bytes, hipMemcpyHostToDevice, ws.s()));
                d_px = d;
            }
            made[i].reset(make_page(d_px, type, order, height, width, channels, ws.s(), e->tm()));
            made[i]->source = pixels[i];
        }
        ws.sync();
        if (e->tm()) e->tm()->collect();
    });
    for (size_t i = 0; i < n; i++) out[i] = made[i].release();
}

ocrs_status ocrs_group_prepare_input_batch(const ocrs_engine_group* g, const void* const* pixels, size_t n, ocrs_pixel_type type,
                                           ocrs_dim_order order, int height, int width, int channels, ocrs_page** out) {
    return guarded([&] { group_prepare(const_cast<ocrs_engine_group*>(g), pixels, n, false, type, order, height, width, channels, out); });
}

ocrs_status ocrs_group_prepare_input_device_batch(const ocrs_engine_group* g, const void* const* d_pixels, size_t n,
                                                  ocrs_pixel_type type, ocrs_dim_order order, int height, int width, int channels,
                                                  ocrs_page** out) {
    return guarded([&] { group_prepare(const_cast<ocrs_engine_group*>(g), d_pixels, n, true, type, order, height, width, channels, out); });
}

ocrs_status ocrs_group_detect_words_batch(ocrs_engine_group* g, const ocrs_page* const* pages, size_t n_pages, float** rects,
                                          size_t* offsets) {
    return guarded([&] {
        if (!g || !rects || !offsets || (n_pages && !pages)) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        const auto of_member = deal_pages(g, pages, n_pages);
        const size_t G = g->size();
        // member m's payload: for each of its pages (in page order) [u64 word count | count x 6 f32]
        std::vector<std::vector<uint8_t>> payloads(G);
        for_each_member(g, of_member, [&](size_t m) {
            std::vector<const ocrs_page*> mine;
            for (size_t i : of_member[m]) mine.push_back(pages[i]);
            std::vector<std::vector<RotatedRect>> rr;
            if (g->replay_mode == 2) {
                {
                    std::lock_guard<std::mutex> lk(g->replay_mu);
                    for (const ocrs_page* pg : mine) {
                        auto it = g->replay_rects.find(pg->source);
                        if (it == g->replay_rects.end()) fail(OCRS_ERR_INVALID_ARGUMENT, "replay: no recorded detection for this page");
                        rr.push_back(it->second);
                    }
                }
                std::this_thread::sleep_for(std::chrono::duration<double>(g->replay_s[1]));
            } else {
                g->members[m].engine->detect(mine.data(), mine.size(), &rr, nullptr);
                if (g->replay_mode == 1) {
                    std::lock_guard<std::mutex> lk(g->replay_mu);
                    for (size_t j = 0; j < mine.size(); j++) g->replay_rects[mine[j]->source] = rr[j];
                }
            }
            auto& pl = payloads[m];
            for (const auto& page_rects : rr) {
                const uint64_t cnt = page_rects.size();
                append_bytes(pl, &cnt, 1);
                for (const RotatedRect& r : page_rects) {
                    float a[6];
                    r.to_array(a);
                    append_bytes(pl, a, 6);
                }
            }
        });
        std::vector<size_t> moffs;
        const std::vector<uint8_t> all = gather_request(g, payloads, &moffs);
        // back to page order: a member's pages appear in its payload in ascending page order
        std::vector<Cursor> cur = cursors(all, moffs);
        const std::vector<size_t> mo = member_of_page(of_member, n_pages);
        std::vector<float> flat;
        offsets[0] = 0;
        for (size_t i = 0; i < n_pages; i++) {
            Cursor& c = cur[mo[i]];
            const uint64_t cnt = c.count();
            if (cnt > (c.end - c.at) / (6 * sizeof(float))) c.need(SIZE_MAX);
            const float* src = reinterpret_cast<const float*>(c.take(cnt * 6 * sizeof(float)));
            flat.insert(flat.end(), src, src + cnt * 6);
            offsets[i + 1] = flat.size() / 6;
        }
        *rects = dup_buffer(flat);
    });
}

ocrs_status ocrs_group_recognize_text_batch(ocrs_engine_group* g, const ocrs_page* const* pages, size_t n_pages,
                                            const size_t* page_line_offsets, const float* line_rects, const size_t* line_offsets,
                                            size_t n_lines, ocrs_text_char** chars, size_t** char_offsets) {
    return guarded([&] {
        if (!g || !page_line_offsets || !line_offsets || !chars || !char_offsets || (n_pages && !pages))
            fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        if (page_line_offsets[0] != 0 || page_line_offsets[n_pages] != n_lines)
            fail(OCRS_ERR_INVALID_ARGUMENT, "page_line_offsets do not cover n_lines");
        for (size_t i = 0; i < n_pages; i++)
            if (page_line_offsets[i] > page_line_offsets[i + 1]) fail(OCRS_ERR_INVALID_ARGUMENT, "page_line_offsets must not decrease");
        for (size_t l = 0; l < n_lines; l++)
            if (line_offsets[l] > line_offsets[l + 1]) fail(OCRS_ERR_INVALID_ARGUMENT, "line_offsets must not decrease");
        if (n_lines && line_offsets[n_lines] > 0 && !line_rects) fail(OCRS_ERR_INVALID_ARGUMENT, "null line_rects");
        const auto of_member = deal_pages(g, pages, n_pages);
        const size_t G = g->size();
        // member m's payload: for each of its lines (page order, then line order) [u64 char count | count x ocrs_text_char]
        std::vector<std::vector<uint8_t>> payloads(G);
        for_each_member(g, of_member, [&](size_t m) {
            const ocrs_engine* e = g->members[m].engine.get();
            std::vector<const ocrs_page*> mine;
            std::vector<std::vector<std::vector<RotatedRect>>> lpp;
            for (size_t i : of_member[m]) {
                mine.push_back(pages[i]);
                lpp.push_back(unpack_lines(line_rects, line_offsets, page_line_offsets[i], page_line_offsets[i + 1]));
            }
            std::vector<ocrs_text_char> flat;
            std::vector<size_t> offs;
            if (g->replay_mode == 2) {
                offs.assign(1, 0);
                {
                    std::lock_guard<std::mutex> lk(g->replay_mu);
                    for (size_t j = 0; j < mine.size(); j++) {
                        auto it = g->replay_lines.find(mine[j]->source);
                        if (it == g->replay_lines.end() || it->second.size() != lpp[j].size())
                            fail(OCRS_ERR_INVALID_ARGUMENT, "replay: no recorded recognition for this page and these lines");
                        for (const auto& line : it->second) {
                            flat.insert(flat.end(), line.begin(), line.end());
                            offs.push_back(flat.size());
                        }
                    }
                }
                std::this_thread::sleep_for(std::chrono::duration<double>(g->replay_s[2]));
            } else {
                std::vector<std::vector<CtcStep>> steps;
                std::vector<RecLine> rl;
                std::vector<uint32_t> ctc_len;
                e->recognize(mine.data(), mine.size(), lpp, &steps, &rl, &ctc_len);
                flatten_chars(e, rl, ctc_len, steps, &flat, &offs);
                if (g->replay_mode == 1) {
                    std::lock_guard<std::mutex> lk(g->replay_mu);
                    size_t l = 0;
                    for (size_t j = 0; j < mine.size(); j++) {
                        auto& rec = g->replay_lines[mine[j]->source];
                        rec.clear();
                        for (size_t q = 0; q < lpp[j].size(); q++, l++)
                            rec.emplace_back(flat.begin() + offs[l], flat.begin() + offs[l + 1]);
                    }
                }
            }
            auto& pl = payloads[m];
            for (size_t l = 0; l + 1 < offs.size(); l++) {
                const uint64_t cnt = offs[l + 1] - offs[l];
                append_bytes(pl, &cnt, 1);
                append_bytes(pl, flat.data() + offs[l], cnt);
            }
        });
        std::vector<size_t> moffs;
        const std::vector<uint8_t> all = gather_request(g, payloads, &moffs);
        std::vector<Cursor> cur = cursors(all, moffs);
        const std::vector<size_t> mo = member_of_page(of_member, n_pages);
        std::vector<ocrs_text_char> flat;
        std::vector<size_t> offs{0};
        for (size_t i = 0; i < n_pages; i++) {
            Cursor& c = cur[mo[i]];
            for (size_t l = page_line_offsets[i]; l < page_line_offsets[i + 1]; l++) {
                const uint64_t cnt = c.count();
                if (cnt > (c.end - c.at) / sizeof(ocrs_text_char)) c.need(SIZE_MAX);
                const uint8_t* src = c.take(cnt * sizeof(ocrs_text_char));
                const size_t old = flat.size();
                flat.resize(old + cnt);
                if (cnt) memcpy(flat.data() + old, src, cnt * sizeof(ocrs_text_char));
                offs.push_back(flat.size());
            }
        }
        *chars = dup_buffer(flat);
        *char_offsets = dup_buffer(offs);
    });
}

static void gather_entry(ocrs_engine_group* g, ocrs_gather_mode mode, const void* const* payloads, const size_t* bytes, void** out,
                         size_t* offsets) {
    if (!g || !payloads || !bytes || !out || !offsets) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
    const size_t G = g->size();
    std::vector<std::vector<uint8_t>> pl(G);
    for (size_t m = 0; m < G; m++) {
        if (bytes[m] && !payloads[m]) fail(OCRS_ERR_INVALID_ARGUMENT, "null payload");
        const uint8_t* p = static_cast<const uint8_t*>(payloads[m]);
        pl[m].assign(p, p + bytes[m]);
    }
    std::vector<size_t> offs;
    const std::vector<uint8_t> all = gather(g, mode, pl, &offs);
    for (size_t m = 0; m <= G; m++) offsets[m] = offs[m];
    *out = dup_buffer(all);
}

ocrs_status ocrs_group_gather(ocrs_engine_group* g, const void* const* payloads, const size_t* bytes, void** out, size_t* offsets) {
    return guarded([&] {
        if (!g) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        gather_entry(g, g->gather == OCRS_GATHER_RCCL ? OCRS_GATHER_RCCL : OCRS_GATHER_HOST, payloads, bytes, out, offsets);
    });
}

ocrs_status ocrs_group_final_gather(ocrs_engine_group* g, ocrs_gather_mode mode, const void* const* payloads, const size_t* bytes,
                                    void** out, size_t* offsets) {
    return guarded([&] {
        if (mode != OCRS_GATHER_AUTO && mode != OCRS_GATHER_HOST && mode != OCRS_GATHER_RCCL)
            fail(OCRS_ERR_INVALID_ARGUMENT, "unknown gather mode %d", (int)mode);
        gather_entry(g, mode, payloads, bytes, out, offsets);
    });
}

ocrs_status ocrs_group_set_replay(ocrs_engine_group* g, int mode, const double seconds[3]) {
    return guarded([&] {
        if (!g || mode < 0 || mode > 2) fail(OCRS_ERR_INVALID_ARGUMENT, "bad argument");
        if (mode == 2) {
            if (!seconds) fail(OCRS_ERR_INVALID_ARGUMENT, "replay needs the share times");
            for (int i = 0; i < 3; i++) {
                if (!(seconds[i] >= 0.0 && seconds[i] < 60.0)) fail(OCRS_ERR_INVALID_ARGUMENT, "share time out of range");
                g->replay_s[i] = seconds[i];
            }
        }
        if (mode == 0) {
            std::lock_guard<std::mutex> lk(g->replay_mu);
            g->replay_rects.clear();
            g->replay_lines.clear();
        }
        g->replay_mode = mode;
    });
}

ocrs_status ocrs_group_member_stats(const ocrs_engine_group* g, size_t m, uint64_t out[8]) {
    return guarded([&] {
        if (!g || !out || m >= g->size()) fail(OCRS_ERR_INVALID_ARGUMENT, "no such member");
        const auto& mem = g->members[m];
        out[0] = mem.shares.load(); out[1] = mem.pages.load(); out[2] = mem.cpu_ns.load(); out[3] = mem.wall_ns.load();
        out[4] = (uint64_t)(mem.numa_node + 1);   // 0 = unknown
        out[5] = mem.cpus.size();
        out[6] = mem.bound_shares.load();
        out[7] = (uint64_t)mem.device;
    });
}

ocrs_status ocrs_numa_parse_cpulist(const char* list, int32_t* cpus, size_t capacity, size_t* n_cpus) {
    return guarded([&] {
        if (!list || !n_cpus) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        std::vector<int> v;
        if (!numa::parse_cpulist(list, &v)) fail(OCRS_ERR_INVALID_ARGUMENT, "malformed cpu list");
        *n_cpus = v.size();
        for (size_t i = 0; i < v.size() && i < capacity && cpus; i++) cpus[i] = v[i];
    });
}

ocrs_status ocrs_numa_bind_selftest(const char* sysfs_root, const char* pci_bus_id, int* node, int* cpus_inside, int* cpus_after) {
    return guarded([&] {
        if (!node || !cpus_inside || !cpus_after) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        const char* root = sysfs_root && *sysfs_root ? sysfs_root : "/sys";
        *node = numa::node_of_pci(pci_bus_id, root);
        std::vector<int> cpus;
        if (!numa::cpus_of_node(*node, &cpus, root)) cpus.clear();
        {
            numa::BindScope place(cpus);
            *cpus_inside = place.bound() ? numa::affinity_count() : -1;
        }
        *cpus_after = numa::affinity_count();
    });
}

ocrs_status ocrs_group_worker_threads(const ocrs_engine_group* g, size_t* n) {
    return guarded([&] {
        if (!g || !n) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        *n = const_cast<ocrs_engine_group*>(g)->workers.threads();
    });
}

}  // extern "C"
