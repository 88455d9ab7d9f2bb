// A tiny low-latency helper pool for the prover's host-side critical path.  Each sum-check round needs a handful of independent
// fixed-base scalar multiplications and point compressions (5-9 us each) between two device launches; spreading them over a few
// spinning host threads halves the per-round latency.  Workers spin only while a proof is in flight (Session), otherwise they sleep.
#pragma once
#include <sched.h>
#include <pthread.h>
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

// ---- where helper threads run.  A prover thread hands its helpers a few microseconds of work dozens of times per proof; on a
// two-socket, many-CCX host (the GPU boxes: 2 x 64 cores, 16 L3 groups) the scheduler is free to put a helper on the other socket,
// where every hand-over and every table lookup crosses the fabric (measured: a 1.3 us task taking 7 us).  Helpers are therefore
// kept on cores that share the last-level cache with the thread they serve, one core each, never that thread's own core
// (OTTI_PIN_HELPERS=0: leave them to the scheduler).  Linux sysfs; anything missing or refused leaves the affinity alone.
namespace cpu_place {
inline bool read_cpu_list(const char *path, std::vector<int> &out) {
    out.clear();
    FILE *f = fopen(path, "r"); if (!f) return false;
    char buf[512] = {0}; const bool ok = fgets(buf, sizeof buf, f) != nullptr; fclose(f);
    if (!ok) return false;
    for (char *p = buf; *p;) {
        if (*p < '0' || *p > '9') { p++; continue; }
        const long a = strtol(p, &p, 10); long b = a;
        if (*p == '-') b = strtol(p + 1, &p, 10);
        for (long x = a; x <= b && out.size() < 4096; x++) out.push_back((int)x);
    }
    return !out.empty();
}
inline std::vector<int> siblings_of(int cpu) {
    char path[128]; snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
    std::vector<int> v; if (!read_cpu_list(path, v)) v = {cpu};
    return v;
}
// one entry per physical core that shares the L3 with `cpu`, the core of `cpu` itself excluded: each entry = that core's logical CPUs
inline std::vector<std::vector<int>> neighbour_cores(int cpu) {
    std::vector<std::vector<int>> cores;
    char path[128]; snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
    std::vector<int> l3; if (!read_cpu_list(path, l3)) return cores;
    const std::vector<int> mine = siblings_of(cpu);
    std::vector<char> seen(4096, 0);
    for (int m : mine) if (m >= 0 && m < 4096) seen[m] = 1;
    for (int x : l3) {
        if (x < 0 || x >= 4096 || seen[x]) continue;
        std::vector<int> sib = siblings_of(x);
        for (int y : sib) if (y >= 0 && y < 4096) seen[y] = 1;
        cores.push_back(sib);
    }
    return cores;
}
inline std::atomic<int> &sessions() { static std::atomic<int> n{0}; return n; }       // prover threads with a Session open, process-wide
inline bool enabled() { static const bool on = [] { const char *e = getenv("OTTI_PIN_HELPERS"); return !(e && e[0] == '0'); }(); return on; }
}  // namespace cpu_place

class SpinPool {
public:
    static SpinPool &get() { thread_local SpinPool p; return p; }   // helpers belong to the prover thread that uses them
    // helpers this prover thread may use right now: all of them while it is the process's only prover at work; with several proofs in flight
    // on threads of their own, the cores are shared out (cores / provers - 1 each): six provers with three spinning helpers each on a 16-core
    // share ran at 470 M constraints/s, with one helper each at 730 (tools/inflight_variants.sh, profiles/r4_inflight_variants.txt).
    // The helpers beyond the budget stop spinning (run()).
    int workers() const {
        const int n = (int)th_.size(), s = cpu_place::sessions().load(std::memory_order_relaxed);
        if (s <= 1 || fixed_) return n;
        return std::max(0, std::min(n, (int)cores_ / s - 1));
    }

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
    // A task its helper has not picked up by the time the caller wants the result is taken back and run by the caller: the helper is most
    // likely off its core (it spins pinned to one core; another process's thread scheduled there costs it a time slice: proofs of 6-11 ms among
    // the 3.0 ms ones, 0.4 % of them on a busy box, tools/outlier_probe.py) — waiting for it would be waiting for the scheduler.
    void wait(int w) {
        Slot &s = slots_[w];
        for (unsigned idle = 0;; idle++) {
            const int st = s.state.load(std::memory_order_acquire);
            if (st == 2) break;
            // (a helper that is on its core takes a task within ~50 ns; after a microsecond or two without that, it is not)
            if (st == 1 && idle >= 48) { int exp = 1; if (s.state.compare_exchange_strong(exp, 4, std::memory_order_acq_rel)) { s.fn(); break; } }
            relax();
        }
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
    unsigned cores_ = 1; bool fixed_ = false;
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
        cores_ = hc ? hc : 1;
        if (const char *e = getenv("OTTI_HOST_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 16) n = v - 1; }
        if (const char *e = getenv("OTTI_HOST_THREADS_FIXED")) fixed_ = e[0] == '1';      // keep every helper whatever the number of provers at work
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
        if (on) {
            // one prover at work in the process: its helpers go next to it.  Several (proofs in flight on their own threads): the
            // scheduler spreads the threads better than a rule that would put every prover's helpers on the first cores of its L3 group
            if (cpu_place::sessions().fetch_add(1) == 0) { place_helpers(); hold_caller(); } else release_helpers();
            active_.fetch_add(1); std::lock_guard<std::mutex> lk(mu_); cv_.notify_all();
        } else { active_.fetch_sub(1); cpu_place::sessions().fetch_sub(1); free_caller(); }
    }
    // for the length of the session the calling thread stays on the free cores of the group its helpers were placed in (it would
    // otherwise wander — onto a helper's core, or to another group and away from its helpers and its warm tables); its own mask comes back after
    void hold_caller() {
        if (placed_for_ < 0 || group_.empty() || caller_held_) return;
        if (sched_getaffinity(0, sizeof saved_mask_, &saved_mask_) != 0) return;
        cpu_set_t set; CPU_ZERO(&set); int n = 0;
        for (int x = 0; x < (int)group_.size() && x < CPU_SETSIZE; x++) if (group_[x] == 1 && CPU_ISSET(x, &saved_mask_)) { CPU_SET(x, &set); n++; }
        if (n && sched_setaffinity(0, sizeof set, &set) == 0) caller_held_ = true;
    }
    void free_caller() { if (caller_held_) { (void)sched_setaffinity(0, sizeof saved_mask_, &saved_mask_); caller_held_ = false; } }
    cpu_set_t saved_mask_; bool caller_held_ = false;
    void release_helpers() {
        if (placed_for_ < 0) return;
        placed_for_ = -1; group_.clear();
        cpu_set_t all; CPU_ZERO(&all);
        if (sched_getaffinity(0, sizeof all, &all) != 0) return;         // the calling thread's own mask: what the helpers were created with
        for (auto &t : th_) (void)pthread_setaffinity_np(t.native_handle(), sizeof all, &all);
    }
    // helper i on the i-th core that shares the L3 with the calling thread's current CPU; redone only when that thread has moved to another core
    void place_helpers() {
        if (th_.empty() || !cpu_place::enabled()) return;
        const int cpu = sched_getcpu();
        if (cpu < 0 || cpu == placed_for_) return;
        // still in the L3 group the helpers were placed for, and not on a core one of them holds: nothing to do (the calling thread is
        // not pinned and wanders between the cores of a group; re-reading sysfs per proof would cost more than it saves)
        if (placed_for_ >= 0 && cpu < (int)group_.size() && group_[cpu] == 1) { return; }
        placed_for_ = cpu;
        std::vector<std::vector<int>> cores = cpu_place::neighbour_cores(cpu);
        {   // only cores the calling thread itself may run on (a caller confined by taskset / a cpuset keeps its helpers inside that set)
            cpu_set_t mine; CPU_ZERO(&mine);
            if (sched_getaffinity(0, sizeof mine, &mine) == 0) {
                std::vector<std::vector<int>> ok;
                for (auto &c : cores) { bool in = !c.empty(); for (int x : c) in = in && x < CPU_SETSIZE && CPU_ISSET(x, &mine); if (in) ok.push_back(c); }
                cores.swap(ok);
            }
        }
        group_.assign(4096, 0);
        if (cores.size() < th_.size()) return;                   // fewer neighbour cores than helpers (or no topology information): leave them alone
        for (int x : cpu_place::siblings_of(cpu)) if (x >= 0 && x < 4096) group_[x] = 1;
        for (size_t i = 0; i < cores.size(); i++) for (int x : cores[i]) if (x >= 0 && x < 4096) group_[x] = i < th_.size() ? 2 : 1;   // 2: a helper's core
        for (size_t i = 0; i < th_.size(); i++) {
            cpu_set_t set; CPU_ZERO(&set);
            for (int x : cores[i]) if (x < CPU_SETSIZE) CPU_SET(x, &set);
            (void)pthread_setaffinity_np(th_[i].native_handle(), sizeof set, &set);
        }
    }
    int placed_for_ = -1;
    std::vector<char> group_;                                     // per logical CPU: 1 = in the L3 group the helpers were placed for and free, 2 = a helper's core
    void run(int i) {
        Slot &s = slots_[i];
        for (;;) {
            if (quit_.load()) return;
            if (s.state.load(std::memory_order_acquire) == 1) {
                int exp = 1;                                          // (the caller may take the task back: wait())
                if (s.state.compare_exchange_strong(exp, 3, std::memory_order_acq_rel)) { s.fn(); s.state.store(2, std::memory_order_release); }
                continue;
            }
            if (active_.load(std::memory_order_relaxed) > 0) {
                if (i < workers()) relax();
                else std::this_thread::sleep_for(std::chrono::microseconds(20));      // over the budget of a crowded process: leave the core to a prover thread
                continue;
            }
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait_for(lk, std::chrono::milliseconds(50), [&] { return quit_.load() || active_.load() > 0 || s.state.load() == 1; });
        }
    }
};

}  // namespace otti
