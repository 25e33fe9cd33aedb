// c3_feed.h -- a feeder thread per handle: c3_feed_push / c3_feed_wait / c3_feed_drain (include/c3hip.h).
//
// The reference's stage-B loop (clair3/CallVariantsFromCffi.py:302-353) is ONE Python thread: it pulls a batch from its generator,
// makes one blocking model call, copies the rows into shared memory, submits a decode task and then sits in as_completed until a
// decode process is done.  With the submit / wait ring of c3_hostring.h driven from that thread (the rebound generator submits
// groups of batches ahead) the GPU only gets new work when the loop happens to call the generator: measured on the unmodified
// loop (tests/diag/loop_timeline.py, 240 k full-alignment windows, 8 decode processes) the thread spent 0.23 s in the generator
// (staging copies of 23 MB per batch), 0.21 s waiting for rows and 0.32 s waiting for decode processes -- one after the other.
// A Python feeder thread does not fix it (2.06 s instead of 0.98: it waits for the interpreter lock behind the loop thread).
// Here the ring is driven by a native thread: the caller pushes (windows, rows) pairs -- descriptors only, no copy -- and the
// thread runs predict_submit / c3_predict_wait over them in order, kFeedRing batches in flight, whatever the caller is doing;
// c3_feed_wait(ticket) blocks (outside the interpreter lock, for a Python caller) until that batch's rows are in its buffer.
// Same rows as c3_predict on the same windows (a window's row does not depend on the batch it travels in, and the feeder uses
// the very ring c3_predict uses).
#pragma once
#include <condition_variable>
#include <deque>
#include <thread>

#include "c3_hostring.h"

struct FeedJob {
    const void *x = nullptr;
    float *y = nullptr;
    int64_t batch = 0, id = 0;
    int dtype = 0;
    bool submitted = false;
};

constexpr int kFeedRing = 3;  // batches in flight: staging copy + H2D of one, kernels of another, D2H of a third (slots 0..2 of the ring)
static_assert(kFeedRing <= kHostSlots, "the feeder uses slots of the submit / wait ring");

struct c3_feeder {
    c3_model *m = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv_push, cv_done;
    std::deque<FeedJob> queue;  // pushed, not yet submitted
    int64_t next_id = 0;        // tickets are consecutive
    int64_t done_upto = 0;      // jobs [0, done_upto) are complete, in order
    std::map<int64_t, std::pair<int, std::string>> errors;  // failed jobs only: return code and message
    bool stop = false;
    // where the thread's time went (c3_model_describe reports it: "feed=<batches>:<ms in submit>:<ms in wait>:<ms idle>")
    double ms_submit = 0, ms_wait = 0, ms_idle = 0;
    static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    void finish(const FeedJob &j, int rc) {  // (feeder thread) the job's rows are in its buffer, or it failed
        std::lock_guard<std::mutex> lk(mu);
        if (rc != 0) errors[j.id] = {rc, g_err};
        done_upto = j.id + 1;
        cv_done.notify_all();
    }

    void run() {
        (void)hipSetDevice(m->device);
        std::deque<FeedJob> flight;  // submitted (or failed at submit), oldest first; slot = id % kFeedRing
        for (;;) {
            FeedJob j;
            bool have = false;
            {
                std::unique_lock<std::mutex> lk(mu);
                if (flight.empty()) {
                    const double t0 = now_ms();
                    cv_push.wait(lk, [&] { return stop || !queue.empty(); });
                    ms_idle += now_ms() - t0;
                }
                if (stop && flight.empty()) {
                    // whatever was pushed and never started is reported as cancelled
                    for (const FeedJob &q : queue) errors[q.id] = {-1, "cancelled: the handle is being destroyed"};
                    if (!queue.empty()) done_upto = queue.back().id + 1;
                    queue.clear();
                    cv_done.notify_all();
                    return;
                }
                if (!stop && !queue.empty() && (int)flight.size() < kFeedRing) {
                    j = queue.front();
                    queue.pop_front();
                    have = true;
                }
            }
            if (have) {
                const double t0 = now_ms();
                const int rc = predict_submit(m, j.x, j.dtype, j.batch, j.y, (int)(j.id % kFeedRing), false);
                ms_submit += now_ms() - t0;
                j.submitted = rc == 0;
                if (rc != 0) {
                    // a failed submit completes in order, behind the batches in flight before it
                    while (!flight.empty()) retire(flight);
                    finish(j, rc);
                } else {
                    flight.push_back(j);
                }
            } else if (!flight.empty()) {
                retire(flight);  // the ring is full, or nothing is waiting to go in: the oldest batch's rows go home
            }
        }
    }

    void retire(std::deque<FeedJob> &flight) {
        const FeedJob j = flight.front();
        flight.pop_front();
        const double t0 = now_ms();
        const int rc = c3_predict_wait(m, (int)(j.id % kFeedRing));
        ms_wait += now_ms() - t0;
        finish(j, rc);
    }
};

// the ring belongs to the feeder while it has batches outstanding: a direct call from another thread would share its slots
static int feeder_owns_ring(c3_model *m) {
    c3_feeder *f = m->feeder;
    if (!f || std::this_thread::get_id() == f->th.get_id()) return 0;
    std::lock_guard<std::mutex> lk(f->mu);
    if (f->done_upto != f->next_id) return fail("the handle's feeder has batches in flight: c3_feed_drain first");
    return 0;
}

extern "C" {

int c3_feed_push(c3_model *m, const void *x_host, int x_dtype, int64_t batch, float *y_host, int64_t *ticket) {
    if (!m) return fail("null model");
    if (!ticket) return fail("null ticket");
    if (batch < 0) return fail("negative batch");
    if (batch > 0 && (!x_host || !y_host)) return fail("null buffer");
    if (!m->loaded) return fail("model weights not loaded");
    if (c3_model_window_bytes(m, x_dtype) <= 0) return fail("unsupported window dtype %d", x_dtype);
    if (!m->feeder) {
        for (const HostSlot &sl : m->slot)
            if (sl.busy) return fail("a prediction is in flight: call c3_predict_wait first");
        m->feeder = new c3_feeder;
        m->feeder->m = m;
        m->feeder->th = std::thread([f = m->feeder] { f->run(); });
    }
    c3_feeder *f = m->feeder;
    FeedJob j;
    j.x = x_host, j.y = y_host, j.batch = batch, j.dtype = x_dtype;
    {
        std::lock_guard<std::mutex> lk(f->mu);
        j.id = f->next_id++;
        f->queue.push_back(j);
    }
    f->cv_push.notify_one();
    *ticket = j.id;
    return 0;
}

int c3_feed_wait(c3_model *m, int64_t ticket) {
    if (!m) return fail("null model");
    c3_feeder *f = m->feeder;
    if (!f) return fail("nothing was pushed on this handle");
    std::unique_lock<std::mutex> lk(f->mu);
    if (ticket < 0 || ticket >= f->next_id) return fail("unknown ticket %lld", (long long)ticket);
    f->cv_done.wait(lk, [&] { return f->done_upto > ticket; });
    const auto it = f->errors.find(ticket);
    if (it == f->errors.end()) return 0;
    const std::pair<int, std::string> e = it->second;
    f->errors.erase(it);
    lk.unlock();
    g_err = e.second;
    return e.first;
}

int c3_feed_drain(c3_model *m) {
    if (!m) return fail("null model");
    c3_feeder *f = m->feeder;
    if (!f) return 0;
    std::unique_lock<std::mutex> lk(f->mu);
    f->cv_done.wait(lk, [&] { return f->done_upto == f->next_id; });
    if (f->errors.empty()) return 0;
    const std::pair<int, std::string> e = f->errors.begin()->second;  // the oldest failure nobody waited for
    f->errors.clear();
    lk.unlock();
    g_err = e.second;
    return e.first;
}

}  // extern "C"

static void feeder_destroy(c3_model *m) {
    c3_feeder *f = m->feeder;
    if (!f) return;
    {
        std::lock_guard<std::mutex> lk(f->mu);
        f->stop = true;
    }
    f->cv_push.notify_all();
    if (f->th.joinable()) f->th.join();
    delete f;
    m->feeder = nullptr;
}
