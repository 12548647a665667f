// Host-side thread helpers of the engine (no HIP in this header: tests/sanitize builds it with -fsanitize=thread).
//
//   WorkerPool     persistent worker threads of an engine group: the members' shares of a call run on them
//   run_shares     one call's fan-out over the pool: share 0 on the calling thread, the rest on workers, wait for all
//   for_pages      bounded pool of ocrs_engine_find_text_lines_batch: `threads` workers pull pages from a shared counter
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace ocrs {

// Threads are kept between calls (a thread per call and member cost a creation on the request path, and per-thread
// caches died with it); the pool grows with the number of shares in flight and never shrinks below what it reached.
class WorkerPool {
  public:
    // on_thread_exit runs on every worker thread before it ends (the engine releases per-thread HIP events there)
    explicit WorkerPool(std::function<void()> on_thread_exit = nullptr) : on_exit_(std::move(on_thread_exit)) {}
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;

    // false: no thread could be had — the caller runs the task itself
    bool submit(std::function<void()> fn) {
        std::unique_lock<std::mutex> lk(mu_);
        if (idle_ <= queue_.size()) {   // every idle worker already has a task coming
            bool made = false;
            if (threads_.size() < kMaxThreads) {
                try {
                    threads_.emplace_back([this] { run(); });
                    made = true;
                } catch (const std::system_error&) {
                }
            }
            if (!made && threads_.empty()) return false;
        }
        queue_.push_back(std::move(fn));
        lk.unlock();
        cv_.notify_one();
        return true;
    }
    size_t threads() {
        std::lock_guard<std::mutex> lk(mu_);
        return threads_.size();
    }

  private:
    static constexpr size_t kMaxThreads = 512;
    void run() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            idle_++;
            cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
            idle_--;
            if (queue_.empty()) break;   // stop_
            std::function<void()> fn = std::move(queue_.front());
            queue_.pop_front();
            lk.unlock();
            fn();
            fn = nullptr;   // the task's captures die outside the lock
            lk.lock();
        }
        lk.unlock();
        if (on_exit_) on_exit_();
    }
    std::function<void()> on_exit_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> queue_;
    std::vector<std::thread> threads_;
    size_t idle_ = 0;
    bool stop_ = false;
};

// Runs body(m) for every m with has_work[m]: the first such share on the calling thread, the others on the pool;
// returns when all have finished.  body must not throw out (it stores its own failures); the first stored failure
// (lowest m) is rethrown here.
template <class Body>
void run_shares(WorkerPool& pool, const std::vector<char>& has_work, std::vector<std::exception_ptr>& errs, Body&& body) {
    const size_t G = has_work.size();
    errs.assign(G, nullptr);
    auto guarded_body = [&](size_t m) {
        try {
            body(m);
        } catch (...) {
            errs[m] = std::current_exception();
        }
    };
    std::mutex mu;
    std::condition_variable cv;
    size_t outstanding = 0;
    size_t mine = G;
    for (size_t m = 0; m < G; m++) {
        if (!has_work[m]) continue;
        if (mine == G) { mine = m; continue; }
        {
            std::lock_guard<std::mutex> lk(mu);
            outstanding++;
        }
        bool queued = false;
        try {
            queued = pool.submit([&, m] {
                guarded_body(m);
                std::lock_guard<std::mutex> lk(mu);   // notify under the lock: `cv` lives on the waiter's stack
                if (--outstanding == 0) cv.notify_all();
            });
        } catch (...) {   // out of memory while queueing: as if no thread could be had
        }
        if (!queued) {   // this share runs here, after the others were started
            guarded_body(m);
            std::lock_guard<std::mutex> lk(mu);
            outstanding--;
        }
    }
    if (mine < G) guarded_body(mine);
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return outstanding == 0; });
    }
    for (size_t m = 0; m < G; m++)
        if (errs[m]) std::rethrow_exception(errs[m]);
}

// work(p) for p in [0, n_pages) on `threads` threads (the caller's included); work must not throw.
template <class Work>
void for_pages(size_t n_pages, size_t threads, Work&& work) {
    if (n_pages <= 1 || threads <= 1) {
        for (size_t p = 0; p < n_pages; p++) work(p);
        return;
    }
    const size_t nth = std::min(n_pages, threads);
    std::atomic<size_t> next{0};
    auto loop = [&] { for (size_t p; (p = next.fetch_add(1)) < n_pages;) work(p); };
    std::vector<std::thread> th;
    for (size_t t = 1; t < nth; t++) {
        try {
            th.emplace_back(loop);
        } catch (const std::system_error&) {
            break;   // fewer threads: the pages are still all done
        }
    }
    loop();
    for (auto& t : th) t.join();
}

}  // namespace ocrs
