// Several GPUs behind ONE handle in ONE process (include/ocrs_amd.h "engine group").
//
// The reference engine is immutable `&self` (ocrs/src/lib.rs:183-256) and pages are independent units
// (SURVEY.md §8e): an engine group holds one engine per member device (own weight replica, own stream / memory
// pools — DeviceContext), deals page i to member i mod G, runs the members' shares of a call on their own host
// threads and merges the results in page order.  There is no collective on the compute path.
//
// Result gather.  Two transports, selectable per group, delivering the same bytes:
//   * host: every member hands its packed results over through its own pinned staging (its own PCIe link) and the
//     calling thread concatenates them.  The fast path inside one process, and the default.
//   * RCCL: the packed results of every member — {rect f32 x 6} per word, {char u32, box i32 x 4} per character —
//     go into a device buffer on that member, one grouped ncclAllGather (ncclCommInitAll communicator, one rank per
//     member, librccl linked directly) moves them over xGMI, and the root member's copy is read back in one D2H.
//     This is the "RCCL only for the final result gather" leg north_star names.  RCCL refuses a communicator with
//     the same device twice, so a group with repeated devices (tests on a one-GPU box) falls back to host and says so
//     (ocrs_group_last_gather).
#include <rccl/rccl.h>

#include <atomic>
#include <system_error>
#include <thread>

#include "abi_util.hpp"
#include "kernels.hpp"

using namespace ocrs;
using namespace ocrs::abi;
using namespace ocrs::geom;

struct ocrs_engine_group {
    struct Member {
        int device = 0;
        std::unique_ptr<ocrs_model> detection, recognition;
        std::unique_ptr<ocrs_engine> engine;
    };
    std::vector<Member> members;
    ocrs_gather_mode gather = OCRS_GATHER_AUTO;
    bool rccl_ready = false;           // communicators exist (distinct devices, mode != HOST)
    std::vector<ncclComm_t> comms;
    std::mutex comm_mu;                // collectives on one communicator are issued by one thread at a time
    std::atomic<int> last_mode{0};     // transport of the most recent gather: 1 host, 2 RCCL
    std::atomic<size_t> last_bytes{0};
    std::string why_host;              // why AUTO / RCCL resolved to host ("" if RCCL)

    ~ocrs_engine_group() {
        for (size_t i = 0; i < comms.size(); i++)
            if (comms[i]) (void)ncclCommDestroy(comms[i]);
    }
    size_t size() const { return members.size(); }
};

namespace {

#define OCRS_NCCL(expr)                                                                                     \
    do {                                                                                                    \
        ncclResult_t _r = (expr);                                                                           \
        if (_r != ncclSuccess)                                                                              \
            ::ocrs::fail(OCRS_ERR_DEVICE, "RCCL error %s at %s:%d (%s)", ncclGetErrorString(_r), __FILE__, __LINE__, #expr); \
    } while (0)

// Runs fn(m) for every member with work on its own host thread (member 0's share on the calling thread), each bound
// to its member's device; the first failure (lowest member) is rethrown after all have finished.
template <class Fn>
void for_each_member(const ocrs_engine_group* g, const std::vector<char>& has_work, Fn&& fn) {
    const size_t G = g->size();
    std::vector<std::exception_ptr> errs(G);
    auto body = [&](size_t m) {
        try {
            DeviceScope bind(g->members[m].device);
            fn(m);
        } catch (...) {
            errs[m] = std::current_exception();
        }
    };
    std::vector<std::thread> th;
    size_t mine = G;
    for (size_t m = 0; m < G; m++) {
        if (!has_work[m]) continue;
        if (mine == G) { mine = m; continue; }
        try {
            th.emplace_back(body, m);
        } catch (const std::system_error&) {   // no thread to be had: this member's share runs here, after the others were started
            body(m);
        }
    }
    if (mine < G) body(mine);
    for (auto& t : th) t.join();
    for (size_t m = 0; m < G; m++)
        if (errs[m]) std::rethrow_exception(errs[m]);
}

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

// The same through the devices: member m's payload, prefixed by its length, is uploaded to member m's device, one
// grouped ncclAllGather leaves every member with all G slots, and the root's copy is downloaded.
std::vector<uint8_t> gather_rccl(ocrs_engine_group* g, const std::vector<std::vector<uint8_t>>& payloads,
                                 std::vector<size_t>* offsets) {
    const size_t G = g->size();
    size_t cap = 0;
    for (const auto& p : payloads) cap = std::max(cap, p.size());
    const size_t slot = ((cap + sizeof(uint64_t) + 15) / 16) * 16;   // [u64 length | bytes | padding]
    std::lock_guard<std::mutex> lk(g->comm_mu);
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
    OCRS_NCCL(ncclGroupStart());
    for (size_t m = 0; m < G; m++) {
        ncclResult_t r = ncclAllGather(d_send[m], d_recv[m], slot, ncclUint8, g->comms[m], ws[m]->s());
        if (r != ncclSuccess) {
            (void)ncclGroupEnd();
            fail(OCRS_ERR_DEVICE, "RCCL error %s in ncclAllGather (member %zu)", ncclGetErrorString(r), m);
        }
    }
    OCRS_NCCL(ncclGroupEnd());
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

std::vector<uint8_t> gather(ocrs_engine_group* g, const std::vector<std::vector<uint8_t>>& payloads, std::vector<size_t>* offsets) {
    size_t total = 0;
    for (const auto& p : payloads) total += p.size();
    g->last_bytes.store(total);
    if (g->rccl_ready) {
        g->last_mode.store(2);
        return gather_rccl(g, payloads, offsets);
    }
    g->last_mode.store(1);
    return gather_host(payloads, offsets);
}

template <class T>
void append_bytes(std::vector<uint8_t>& v, const T* p, size_t n) {
    const uint8_t* b = reinterpret_cast<const uint8_t*>(p);
    v.insert(v.end(), b, b + n * sizeof(T));
}

void check_group_pages(const ocrs_engine_group* g, const ocrs_page* const* pages, size_t n) {
    const size_t G = g->size();
    for (size_t i = 0; i < n; i++) {
        if (!pages[i]) fail(OCRS_ERR_INVALID_ARGUMENT, "null page");
        const int want = g->members[i % G].device;
        if (pages[i]->device() != want)
            fail(OCRS_ERR_INVALID_ARGUMENT, "page %zu lives on device %d; the group deals page i to member i mod %zu, here device %d", i,
                 pages[i]->device(), G, want);
    }
}

}  // namespace

extern "C" {

ocrs_status ocrs_engine_group_new(const ocrs_group_params* params, ocrs_engine_group** out) {
    return guarded([&] {
        if (!params || !out) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        if (!params->devices || params->n_devices == 0 || params->n_devices > 64)
            fail(OCRS_ERR_INVALID_ARGUMENT, "an engine group needs 1..64 member devices");
        auto g = std::make_unique<ocrs_engine_group>();
        g->gather = params->gather;
        g->members.resize(params->n_devices);
        bool distinct = true;
        for (size_t m = 0; m < params->n_devices; m++) {
            auto& mem = g->members[m];
            mem.device = params->devices[m];
            for (size_t k = 0; k < m; k++)
                if (g->members[k].device == mem.device) distinct = false;
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
            mem.engine = make_engine(ep);
            mem.engine->device = mem.device;   // (an engine without weights has nothing else to pin it to its device)
        }
        if (params->gather == OCRS_GATHER_HOST) {
            g->why_host = "host transport requested";
        } else if (!distinct) {
            g->why_host = "a device appears more than once in the group: RCCL refuses such a communicator";
        } else if (params->gather == OCRS_GATHER_AUTO && params->n_devices == 1) {
            g->why_host = "one member: nothing to gather";
        } else {
            std::vector<int> devs;
            for (const auto& mem : g->members) devs.push_back(mem.device);
            g->comms.assign(devs.size(), nullptr);
            OCRS_NCCL(ncclCommInitAll(g->comms.data(), (int)devs.size(), devs.data()));
            g->rccl_ready = true;
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

ocrs_status ocrs_group_deal(size_t n_pages, size_t n_members, size_t* member_of_page, size_t* pages_per_member) {
    return guarded([&] {
        if (n_members == 0 || (n_pages && !member_of_page)) fail(OCRS_ERR_INVALID_ARGUMENT, "bad argument");
        if (pages_per_member)
            for (size_t m = 0; m < n_members; m++) pages_per_member[m] = 0;
        for (size_t i = 0; i < n_pages; i++) {
            member_of_page[i] = i % n_members;   // page i -> GPU i mod G
            if (pages_per_member) pages_per_member[i % n_members]++;
        }
    });
}

ocrs_status ocrs_group_last_gather(const ocrs_engine_group* g, int* transport, size_t* bytes, const char** why_host) {
    return guarded([&] {
        if (!g) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        if (transport) *transport = g->last_mode.load();
        if (bytes) *bytes = g->last_bytes.load();
        if (why_host) *why_host = g->why_host.c_str();
    });
}

static void group_prepare(const ocrs_engine_group* g, const void* const* pixels, size_t n, bool on_device, ocrs_pixel_type type,
                          ocrs_dim_order order, int height, int width, int channels, ocrs_page** out) {
    if (!g || !out || (n > 0 && !pixels)) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
    for (size_t i = 0; i < n; i++) check_image_args(pixels[i], height, width, channels);
    const size_t G = g->size();
    std::vector<char> work(G, 0);
    for (size_t i = 0; i < n && i < G; i++) work[i] = 1;
    const size_t bytes = (size_t)height * width * channels * (type == OCRS_U8 ? 1 : 4);
    std::vector<std::unique_ptr<ocrs_page>> made(n);   // freed if any member fails
    for_each_member(g, work, [&](size_t m) {
        const ocrs_engine* e = g->members[m].engine.get();
        Workspace ws;
        for (size_t i = m; i < n; i += G) {
            const void* d_px = pixels[i];
            if (!on_device) {
                void* d = ws.alloc(bytes);
                OCRS_HIP(hipMemcpyAsync(d, pixels[i], bytes, hipMemcpyHostToDevice, ws.s()));
                d_px = d;
            }
            made[i].reset(make_page(d_px, type, order, height, width, channels, ws.s(), e->tm()));
        }
        ws.sync();
        if (e->tm()) e->tm()->collect();
    });
    for (size_t i = 0; i < n; i++) out[i] = made[i].release();
}

ocrs_status ocrs_group_prepare_input_batch(const ocrs_engine_group* g, const void* const* pixels, size_t n, ocrs_pixel_type type,
                                           ocrs_dim_order order, int height, int width, int channels, ocrs_page** out) {
    return guarded([&] { group_prepare(g, pixels, n, false, type, order, height, width, channels, out); });
}

ocrs_status ocrs_group_prepare_input_device_batch(const ocrs_engine_group* g, const void* const* d_pixels, size_t n,
                                                  ocrs_pixel_type type, ocrs_dim_order order, int height, int width, int channels,
                                                  ocrs_page** out) {
    return guarded([&] { group_prepare(g, d_pixels, n, true, type, order, height, width, channels, out); });
}

ocrs_status ocrs_group_detect_words_batch(ocrs_engine_group* g, const ocrs_page* const* pages, size_t n_pages, float** rects,
                                          size_t* offsets) {
    return guarded([&] {
        if (!g || !rects || !offsets || (n_pages && !pages)) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        check_group_pages(g, pages, n_pages);
        const size_t G = g->size();
        std::vector<char> work(G, 0);
        for (size_t i = 0; i < n_pages && i < G; i++) work[i] = 1;
        // member m's payload: for each of its pages (in page order) [u64 word count | count x 6 f32]
        std::vector<std::vector<uint8_t>> payloads(G);
        for_each_member(g, work, [&](size_t m) {
            std::vector<const ocrs_page*> mine;
            for (size_t i = m; i < n_pages; i += G) mine.push_back(pages[i]);
            std::vector<std::vector<RotatedRect>> rr;
            g->members[m].engine->detect(mine.data(), mine.size(), &rr, nullptr);
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
        const std::vector<uint8_t> all = gather(g, payloads, &moffs);
        // back to page order: page i is the (i / G)-th page of member i mod G
        std::vector<size_t> cursor(moffs.begin(), moffs.end() - 1);
        std::vector<float> flat;
        offsets[0] = 0;
        for (size_t i = 0; i < n_pages; i++) {
            size_t& at = cursor[i % G];
            uint64_t cnt = 0;
            memcpy(&cnt, all.data() + at, sizeof cnt);
            at += sizeof cnt;
            const float* src = reinterpret_cast<const float*>(all.data() + at);
            flat.insert(flat.end(), src, src + cnt * 6);
            at += cnt * 6 * sizeof(float);
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
        if (page_line_offsets[n_pages] != n_lines) fail(OCRS_ERR_INVALID_ARGUMENT, "page_line_offsets do not cover n_lines");
        check_group_pages(g, pages, n_pages);
        const size_t G = g->size();
        std::vector<char> work(G, 0);
        for (size_t i = 0; i < n_pages && i < G; i++) work[i] = 1;
        // member m's payload: for each of its lines (page order, then line order) [u64 char count | count x ocrs_text_char]
        std::vector<std::vector<uint8_t>> payloads(G);
        for_each_member(g, work, [&](size_t m) {
            const ocrs_engine* e = g->members[m].engine.get();
            std::vector<const ocrs_page*> mine;
            std::vector<std::vector<std::vector<RotatedRect>>> lpp;
            for (size_t i = m; i < n_pages; i += G) {
                mine.push_back(pages[i]);
                lpp.push_back(unpack_lines(line_rects, line_offsets, page_line_offsets[i], page_line_offsets[i + 1]));
            }
            std::vector<std::vector<CtcStep>> steps;
            std::vector<RecLine> rl;
            std::vector<uint32_t> ctc_len;
            e->recognize(mine.data(), mine.size(), lpp, &steps, &rl, &ctc_len);
            std::vector<ocrs_text_char> flat;
            std::vector<size_t> offs;
            flatten_chars(e, rl, ctc_len, steps, &flat, &offs);
            auto& pl = payloads[m];
            for (size_t l = 0; l + 1 < offs.size(); l++) {
                const uint64_t cnt = offs[l + 1] - offs[l];
                append_bytes(pl, &cnt, 1);
                append_bytes(pl, flat.data() + offs[l], cnt);
            }
        });
        std::vector<size_t> moffs;
        const std::vector<uint8_t> all = gather(g, payloads, &moffs);
        std::vector<size_t> cursor(moffs.begin(), moffs.end() - 1);
        std::vector<ocrs_text_char> flat;
        std::vector<size_t> offs{0};
        for (size_t i = 0; i < n_pages; i++) {
            size_t& at = cursor[i % G];
            for (size_t l = page_line_offsets[i]; l < page_line_offsets[i + 1]; l++) {
                uint64_t cnt = 0;
                memcpy(&cnt, all.data() + at, sizeof cnt);
                at += sizeof cnt;
                const size_t old = flat.size();
                flat.resize(old + cnt);
                if (cnt) memcpy(flat.data() + old, all.data() + at, cnt * sizeof(ocrs_text_char));
                at += cnt * sizeof(ocrs_text_char);
                offs.push_back(flat.size());
            }
        }
        *chars = dup_buffer(flat);
        *char_offsets = dup_buffer(offs);
    });
}

ocrs_status ocrs_group_gather(ocrs_engine_group* g, const void* const* payloads, const size_t* bytes, void** out, size_t* offsets) {
    return guarded([&] {
        if (!g || !payloads || !bytes || !out || !offsets) fail(OCRS_ERR_INVALID_ARGUMENT, "null argument");
        const size_t G = g->size();
        std::vector<std::vector<uint8_t>> pl(G);
        for (size_t m = 0; m < G; m++) {
            if (bytes[m] && !payloads[m]) fail(OCRS_ERR_INVALID_ARGUMENT, "null payload");
            const uint8_t* p = static_cast<const uint8_t*>(payloads[m]);
            pl[m].assign(p, p + bytes[m]);
        }
        std::vector<size_t> offs;
        const std::vector<uint8_t> all = gather(g, pl, &offs);
        for (size_t m = 0; m <= G; m++) offsets[m] = offs[m];
        *out = dup_buffer(all);
    });
}

}  // extern "C"
