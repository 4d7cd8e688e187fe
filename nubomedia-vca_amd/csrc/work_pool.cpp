// work_pool.cpp -- a few helper threads for host work that is independent per job (nvca_internal.h: WorkPool).
// Pure C++ (no HIP): also built under ThreadSanitizer by tests/test_host_sanitizers.py.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace nvca {

struct WorkPool;
WorkPool *work_pool_create(int threads);
void work_pool_destroy(WorkPool *p);
void work_pool_run(WorkPool *p, int n, void (*fn)(void *arg, int i), void *arg);
int work_pool_threads(const WorkPool *p);

// A run is opened by the calling thread (gen goes odd), worked on by the caller and by whichever helpers get there, and closed
// (gen goes even) as soon as every index has been handled: the caller never waits for a helper to WAKE, only for the ones that
// joined the run to leave it.  A helper announces itself (inside++) before it looks at the run's fields and backs out if the
// run it saw has been closed meanwhile, so the caller may rewrite the fields once it has seen inside == 0.  Helpers spin for a
// short while after a run (rounds of a batched call follow each other within a few hundred microseconds) before they sleep.
struct WorkPool {
    std::vector<std::thread> th;
    std::mutex m; std::condition_variable cv;
    std::mutex run_mu;                                  // one run at a time (the pool is shared by the process's contexts)
    void (*fn)(void *, int) = nullptr; void *arg = nullptr; int n = 0;
    std::atomic<int> next{0}, finished{0}, inside{0}, sleepers{0};
    std::atomic<uint64_t> gen{0};                       // odd: a run is open
    std::atomic<bool> stop{false};
    int spin_us = 50;
    static void relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            uint64_t g = 0;
            const auto t0 = std::chrono::steady_clock::now();
            for (int spins = 0;; spins++) {
                if (stop.load()) return;
                g = gen.load();
                if ((g & 1) && g != seen) break;
                if ((spins & 63) != 63 || std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spin_us)) { relax(); continue; }
                std::unique_lock<std::mutex> lk(m);
                sleepers.fetch_add(1);
                cv.wait(lk, [&] { g = gen.load(); return stop.load() || ((g & 1) && g != seen); });
                sleepers.fetch_sub(1);
                if (stop.load()) return;
                break;
            }
            inside.fetch_add(1);
            if (gen.load() != g) { inside.fetch_sub(1); continue; }      // closed before this helper got here (a later run is picked up next time round)
            seen = g;
            for (int i; (i = next.fetch_add(1)) < n;) { fn(arg, i); finished.fetch_add(1); }
            inside.fetch_sub(1);
        }
    }
};
// One pool per PROCESS, shared by every context that asks for helpers (the GStreamer shim holds a context per GPU slot, a bench
// may hold several): the first caller sizes it, later callers get the same threads -- contexts do not multiply spinning helpers
// on an oversubscribed host.  A run is exclusive; a caller that finds the pool busy with another context's run waits for it
// briefly (work_pool_run) and otherwise does its own work alone.
static std::mutex g_pool_mu;
static WorkPool *g_pool = nullptr;
static int g_pool_refs = 0;
WorkPool *work_pool_create(int threads)
{
    if (threads <= 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!g_pool) {
        WorkPool *p = new (std::nothrow) WorkPool();
        if (!p) return nullptr;
        if (const char *e = getenv("NVCA_HOST_SPIN_US")) p->spin_us = std::max(0, atoi(e));
        try { for (int i = 0; i < threads; i++) p->th.emplace_back([p] { p->worker(); }); }
        catch (...) { }                                     // fewer threads than asked for (or none): the caller works anyway
        g_pool = p;
    }
    g_pool_refs++;
    return g_pool;
}
void work_pool_destroy(WorkPool *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (p != g_pool || --g_pool_refs > 0) return;
    { std::lock_guard<std::mutex> lk2(p->m); p->stop.store(true); }
    p->cv.notify_all();
    for (std::thread &t : p->th) t.join();
    delete p;
    g_pool = nullptr;
}
// the caller spins briefly for a helper that has taken an index (a run's items are tens of microseconds), then yields the core:
// a helper descheduled on an oversubscribed host is not waited for at full tilt
static void wait_until(const std::atomic<int> &v, int target, bool at_least)
{
    for (int spins = 0; at_least ? v.load() < target : v.load() != target; spins++) {
        if (spins < 4096) WorkPool::relax(); else std::this_thread::yield();
    }
}
int work_pool_threads(const WorkPool *p) { return p ? (int)p->th.size() : 0; }
void work_pool_run(WorkPool *p, int n, void (*fn)(void *, int), void *arg)
{
    if (!p || p->th.empty() || n < 4) { for (int i = 0; i < n; i++) fn(arg, i); return; }
    // another context's run may be open: it is over within tens of microseconds, and waiting that long for the helpers is cheaper than
    // doing a round's jobs alone (two contexts of one process on two serving threads -- the shim's NVCA_VIRTUAL_GPUS=2 -- otherwise
    // took turns at running their jobs serially); a run that stays busy for longer is not waited for
    std::unique_lock<std::mutex> run(p->run_mu, std::try_to_lock);
    if (!run.owns_lock()) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int spins = 0; !run.owns_lock(); spins++) {
            WorkPool::relax();
            if ((spins & 31) == 31) {
                if (run.try_lock()) break;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
            }
        }
        if (!run.owns_lock()) { for (int i = 0; i < n; i++) fn(arg, i); return; }
    }
    p->fn = fn; p->arg = arg; p->n = n; p->next.store(0); p->finished.store(0);          // (no helper is inside: the previous run waited for that)
    const uint64_t g = p->gen.load() + 1;
    p->gen.store(g);
    if (p->sleepers.load() > 0) { { std::lock_guard<std::mutex> lk(p->m); } p->cv.notify_all(); }      // (through the mutex: a helper between its check and its wait is not missed)
    for (int i; (i = p->next.fetch_add(1)) < n;) { fn(arg, i); p->finished.fetch_add(1); }              // the caller takes part
    wait_until(p->finished, n, true);
    p->gen.store(g + 1);
    wait_until(p->inside, 0, false);
}

} // namespace nvca
