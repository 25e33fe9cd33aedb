// c3_host.h -- host-side helpers of the transport path: a tiny persistent thread pool that copies the
// caller's pageable windows into the pinned staging buffer.
//
// Why: one host thread moves ~10 GB/s; a full-alignment batch of 1000 windows is 23.5 MB (clair3/CallVariantsFromCffi.py:
// 265-269 -> 1000 x 89 x 33 x 8 int8), i.e. 2-3 ms of memcpy against ~1.3 ms of kernels and ~0.5 ms of PCIe Gen5 DMA:
// the staging copy, not the GPU, bounded the host-inclusive rate of round 1.  The pool splits every
// staged piece over three helpers plus the calling thread.  Threads are created lazily on the
// first staged copy.  With the rebound batch generator (callvar.install) that is the generator's first next(), i.e. BEFORE the
// loop's first executor.submit makes the ProcessPoolExecutor fork its decode workers (CallVariantsFromCffi.py:246 vs :302-353):
// the children inherit none of these threads (fork copies the calling thread only) and only ever run numpy / Python, and the
// pool's mutexes are never held across the loop's submit calls, so nothing a child touches is left locked.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace c3 {

class StagePool {
public:
    static StagePool &get() {
        static StagePool *p = new StagePool;  // never destroyed: the helpers may outlive static destruction at exit
        return *p;
    }
    // memcpy(dst, src, n) split over the helpers and the caller; returns when every byte has been copied
    void copy(void *dst, const void *src, size_t n) {
        std::lock_guard<std::mutex> one_call(call_mu_);  // handles on different threads share the pool
        const int helpers = ensure_started();
        const size_t kMin = (size_t)512 << 10;  // below 512 KiB per part the hand-off costs more than it saves
        int parts = (int)std::min<size_t>((size_t)helpers + 1, n / kMin);
        if (parts <= 1) {
            memcpy(dst, src, n);
            return;
        }
        const size_t per = ((n / parts) + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> lk(mu_);
            dst_ = (char *)dst, src_ = (const char *)src, n_ = n, per_ = per, parts_ = parts;
            next_.store(1, std::memory_order_relaxed);  // part 0 is the caller's
            left_ = parts - 1;
            ++gen_;
        }
        cv_.notify_all();
        memcpy(dst, src, std::min(per, n));
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return left_ == 0; });
    }

private:
    StagePool() = default;
    int ensure_started() {
        if (started_) return (int)threads_.size();
        started_ = true;
        const int n = 3;  // measured against 1, 2, 5 and 7 in round 2: the copy is memory-bound beyond four threads
        for (int i = 0; i < n; ++i) {
            threads_.emplace_back([this] { run(); });
            threads_.back().detach();
        }
        return n;
    }
    void run() {
        unsigned long long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            for (;;) {
                const int part = next_.fetch_add(1, std::memory_order_relaxed);
                if (part >= parts_) break;
                const size_t off = (size_t)part * per_;
                char *d = dst_;
                const char *s = src_;
                const size_t len = off < n_ ? std::min(per_, n_ - off) : 0;
                lk.unlock();
                if (len) memcpy(d + off, s + off, len);
                lk.lock();
                if (--left_ == 0) done_.notify_one();
            }
        }
    }
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    bool started_ = false, stop_ = false;
    unsigned long long gen_ = 0;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t n_ = 0, per_ = 0;
    int parts_ = 0, left_ = 0;
    std::atomic<int> next_{0};
};

}  // namespace c3
