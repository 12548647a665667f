// Request coalescing ("group commit") for the engine's GPU stages.
//
// The reference API works one page per call and gets its concurrency from host threads
// (ocrs-cli/src/main.rs:420-446; recognition.rs:465-485 runs Model::run from a rayon pool).  On the GPU a one-page
// recognition request is a poor unit: its BiGRU recurrence is a chain of ~600 dependent steps whatever the number
// of lines, and the recurrences of all requests share one stream per device.  Rows (lines) are independent, so
// requests that are waiting at the same time can be run as ONE ragged request without changing a bit of anybody's
// result — the engine does that here, behind the unchanged one-page entry points.
//
// Mechanism: leader / follower, no dispatcher thread.  A caller enqueues its request; if a batch slot is free it
// becomes the leader of the next batch: it takes the oldest waiting request and every compatible waiting request
// (up to a page budget), runs them as one merged request on its own thread and stream, scatters the results and
// wakes the owners.  While `max_active` batches are in flight new arrivals simply wait — and are taken together
// by whoever leads next: the batch size adapts to the load by itself (one caller alone runs immediately and
// alone; twelve callers settle at a few pages per batch).
#pragma once
#include <chrono>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <vector>

namespace ocrs {

struct CoalescedBase {
    size_t weight = 1;          // pages
    bool taken = false, done = false;
    std::exception_ptr error;
    std::chrono::steady_clock::time_point enqueued;
};

template <class Req>
class Coalescer {
  public:
    // runs a batch (>= 1 requests, all compatible with batch[0]); must not throw: failures go into Req::error
    using RunFn = std::function<void(std::vector<Req*>&)>;
    // may `other` share a batch with `head`?
    using FitFn = std::function<bool(const Req& head, const Req& other)>;

    Coalescer(RunFn run, FitFn fits) : run_(std::move(run)), fits_(std::move(fits)) {}

    // Blocks until `r` has been executed (by this thread as a leader, or by another leader); rethrows its error.
    // max_active: batches in flight at once; max_weight: pages per batch; window_us: while other batches are in
    // flight a would-be leader lets the queue fill for this long (measured from the oldest waiting request)
    // unless it already holds `max_weight / 2` pages.
    void submit(Req& r, int max_active, size_t max_weight, long window_us) {
        std::unique_lock<std::mutex> lk(mu_);
        r.enqueued = std::chrono::steady_clock::now();
        q_.push_back(&r);
        queued_weight_ += r.weight;
        while (!r.done) {
            if (!r.taken && active_ < max_active && !q_.empty()) {
                const auto now = std::chrono::steady_clock::now();
                const auto ripe = q_.front()->enqueued + std::chrono::microseconds(window_us);
                if (active_ == 0 || queued_weight_ * 2 >= max_weight || now >= ripe) {
                    lead(lk, max_weight);
                    continue;
                }
                cv_.wait_until(lk, ripe);
                continue;
            }
            cv_.wait(lk);
        }
        if (r.error) std::rethrow_exception(r.error);
    }

    // statistics (bench / tests)
    void stats(uint64_t* batches, uint64_t* requests) {
        std::lock_guard<std::mutex> g(mu_);
        *batches = n_batches_;
        *requests = n_requests_;
    }

  private:
    void lead(std::unique_lock<std::mutex>& lk, size_t max_weight) {
        std::vector<Req*> batch;
        Req* head = q_.front();
        size_t w = 0;
        for (auto it = q_.begin(); it != q_.end();) {
            Req* x = *it;
            if (x == head || (w + x->weight <= max_weight && fits_(*head, *x))) {
                batch.push_back(x);
                w += x->weight;
                x->taken = true;
                queued_weight_ -= x->weight;
                it = q_.erase(it);
            } else {
                ++it;
            }
        }
        active_++;
        n_batches_++;
        n_requests_ += batch.size();
        lk.unlock();
        run_(batch);
        lk.lock();
        active_--;
        for (Req* x : batch) x->done = true;
        cv_.notify_all();
    }

    RunFn run_;
    FitFn fits_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Req*> q_;
    size_t queued_weight_ = 0;
    int active_ = 0;
    uint64_t n_batches_ = 0, n_requests_ = 0;
};

}  // namespace ocrs
