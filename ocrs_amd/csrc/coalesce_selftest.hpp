// Body of ocrs_coalescer_selftest (include/ocrs_amd.h), header-only and HIP-free so that tests/sanitize can run the
// SAME code under -fsanitize=thread: n_threads callers submit requests_per_thread requests each (two incompatible
// kinds, weights 1..2 pages) to a Coalescer whose "GPU work" is a short sleep; requests whose id is divisible by
// fail_every fail.  out = {batches run, requests carried, errors delivered to their callers, requests that were run
// twice / not at all / got a wrong result or shared a batch with the other kind, largest batch in pages}.
#pragma once
#include <atomic>
#include <cstdint>
#include <stdexcept>
#include <thread>

#include "coalesce.hpp"

namespace ocrs {

inline void coalescer_selftest(int n_threads, int requests_per_thread, int max_active, int max_pages, long window_us, int fail_every,
                               uint64_t out[5]) {
    struct TReq : CoalescedBase { int id = 0, kind = 0; long result = 0; int runs = 0; };
    std::atomic<uint64_t> max_batch_pages{0}, mixed{0};
    Coalescer<TReq> q(
        [&](std::vector<TReq*>& batch) {
            uint64_t w = 0;
            for (TReq* r : batch) { w += r->weight; if (r->kind != batch[0]->kind) mixed++; }
            uint64_t prev = max_batch_pages.load();
            while (w > prev && !max_batch_pages.compare_exchange_weak(prev, w)) {}
            std::this_thread::sleep_for(std::chrono::microseconds(300));   // the "GPU work" of a batch
            for (TReq* r : batch) {
                r->runs++;
                if (fail_every > 0 && r->id % fail_every == 0) {
                    try { throw std::invalid_argument("request " + std::to_string(r->id) + " is bad"); } catch (...) { r->error = std::current_exception(); }
                } else {
                    r->result = 3L * r->id + 1;
                }
            }
        },
        [](const TReq& a, const TReq& b) { return a.kind == b.kind; });
    std::atomic<uint64_t> wrong{0}, errors{0};
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&, t] {
            for (int k = 0; k < requests_per_thread; k++) {
                TReq r;
                r.id = t * requests_per_thread + k + 1;
                r.kind = r.id % 2;
                r.weight = 1 + (size_t)(r.id % 3 == 0);
                try {
                    q.submit(r, max_active, (size_t)max_pages, window_us);
                    if (r.runs != 1 || r.result != 3L * r.id + 1 || (fail_every > 0 && r.id % fail_every == 0)) wrong++;
                } catch (const std::invalid_argument&) {
                    errors++;
                    if (r.runs != 1 || !(fail_every > 0 && r.id % fail_every == 0)) wrong++;
                }
            }
        });
    for (auto& t : th) t.join();
    uint64_t batches = 0, reqs = 0;
    q.stats(&batches, &reqs);
    out[0] = batches; out[1] = reqs; out[2] = errors.load(); out[3] = wrong.load() + mixed.load(); out[4] = max_batch_pages.load();
}

}  // namespace ocrs
