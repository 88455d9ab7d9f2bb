// A tiny low-latency helper pool for the prover's host-side critical path.  Each sum-check round needs a handful of independent
// fixed-base scalar multiplications and point compressions (5-9 us each) between two device launches; spreading them over a few
// spinning host threads halves the per-round latency.  Workers spin only while a proof is in flight (Session), otherwise they sleep.
#pragma once
#include <sched.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <stdlib.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace otti {

class SpinPool {
public:
    static SpinPool &get() { thread_local SpinPool p; return p; }   // helpers belong to the prover thread that uses them
    int workers() const { return (int)th_.size(); }

    struct Session {                                  // RAII: workers spin while one is alive
        Session() { SpinPool::get().set_active(true); }
        ~Session() { SpinPool::get().set_active(false); }
    };
    // run f on worker w (0-based) asynchronously; one outstanding task per worker
    void submit(int w, std::function<void()> f) {
        Slot &s = slots_[w];
        s.fn = std::move(f);
        s.state.store(1, std::memory_order_release);
    }
    void wait(int w) {
        Slot &s = slots_[w];
        while (s.state.load(std::memory_order_acquire) == 1) relax();
        s.state.store(0, std::memory_order_relaxed);
    }
    // run the given tasks concurrently: task 0 on the calling thread, the others on workers (falls back to inline when short of workers)
    void parallel(std::function<void()> *tasks, int n) {
        int nw = workers(), used = 0;
        for (int i = 1; i < n && used < nw; i++, used++) submit(used, tasks[i]);
        tasks[0]();
        for (int i = 1 + used; i < n; i++) tasks[i]();
        for (int w = 0; w < used; w++) wait(w);
    }

private:
    struct alignas(64) Slot { std::atomic<int> state{0}; std::function<void()> fn; };
    std::vector<std::thread> th_;
    std::vector<Slot> slots_;
    std::atomic<bool> quit_{false};
    std::atomic<int> active_{0};
    std::mutex mu_; std::condition_variable cv_;

    static void relax() {
#if defined(__x86_64__)
        _mm_pause();
#else
        std::this_thread::yield();
#endif
    }
    SpinPool() {
        // cores this process may use: its affinity mask, shared with the other rank processes of the node (one per GPU)
        unsigned hc = std::thread::hardware_concurrency();
        cpu_set_t set; CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) hc = (unsigned)CPU_COUNT(&set);
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {      // a container's CPU quota is usually far below its affinity mask
            long long quota = 0, period = 0; char q[32] = {0};
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0 && (quota = atoll(q)) > 0)
                hc = std::min<unsigned>(hc, (unsigned)std::max<long long>(1, quota / period));
            fclose(f);
        }
        if (const char *e = getenv("LOCAL_WORLD_SIZE")) { int v = atoi(e); if (v > 1) hc = hc / (unsigned)v; }
        int n = hc >= 8 ? 3 : hc >= 4 ? 2 : hc >= 2 ? 1 : 0;
        if (const char *e = getenv("OTTI_HOST_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 16) n = v - 1; }
        slots_ = std::vector<Slot>(n > 0 ? n : 1);
        for (int i = 0; i < n; i++) th_.emplace_back([this, i] { run(i); });
    }
    ~SpinPool() {
        quit_.store(true);
        { std::lock_guard<std::mutex> lk(mu_); }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void set_active(bool on) {
        if (on) { active_.fetch_add(1); std::lock_guard<std::mutex> lk(mu_); cv_.notify_all(); }
        else active_.fetch_sub(1);
    }
    void run(int i) {
        Slot &s = slots_[i];
        for (;;) {
            if (quit_.load()) return;
            if (s.state.load(std::memory_order_acquire) == 1) { s.fn(); s.state.store(2, std::memory_order_release); continue; }
            if (active_.load(std::memory_order_relaxed) > 0) { relax(); continue; }
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait_for(lk, std::chrono::milliseconds(50), [&] { return quit_.load() || active_.load() > 0 || s.state.load() == 1; });
        }
    }
};

}  // namespace otti
