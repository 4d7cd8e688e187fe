// work_pool.cpp -- a few helper threads for host work that is independent per job (nvca_internal.h: WorkPool).
// Pure C++ (no HIP): also built under ThreadSanitizer by tests/test_host_sanitizers.py.
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace nvca {

struct WorkPool;
WorkPool *work_pool_create(int threads);
void work_pool_destroy(WorkPool *p);
void work_pool_run(WorkPool *p, int n, void (*fn)(void *arg, int i), void *arg);

struct WorkPool {
    std::vector<std::thread> th;
    std::mutex m; std::condition_variable cv, done;
    void (*fn)(void *, int) = nullptr; void *arg = nullptr;
    int n = 0; std::atomic<int> next{0}; int busy = 0, acked = 0; uint64_t gen = 0; bool stop = false;
    void worker()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen; busy++; acked++;
            lk.unlock();
            for (int i; (i = next.fetch_add(1)) < n;) fn(arg, i);
            lk.lock();
            if (--busy == 0) done.notify_all();
        }
    }
};
WorkPool *work_pool_create(int threads)
{
    if (threads <= 0) return nullptr;
    WorkPool *p = new (std::nothrow) WorkPool();
    if (!p) return nullptr;
    try { for (int i = 0; i < threads; i++) p->th.emplace_back([p] { p->worker(); }); }
    catch (...) { }                                     // fewer threads than asked for (or none): the caller works anyway
    return p;
}
void work_pool_destroy(WorkPool *p)
{
    if (!p) return;
    { std::lock_guard<std::mutex> lk(p->m); p->stop = true; }
    p->cv.notify_all();
    for (std::thread &t : p->th) t.join();
    delete p;
}
void work_pool_run(WorkPool *p, int n, void (*fn)(void *, int), void *arg)
{
    if (!p || p->th.empty() || n < 4) { for (int i = 0; i < n; i++) fn(arg, i); return; }
    { std::lock_guard<std::mutex> lk(p->m); p->fn = fn; p->arg = arg; p->n = n; p->next.store(0); p->acked = 0; p->gen++; }
    p->cv.notify_all();
    for (int i; (i = p->next.fetch_add(1)) < n;) fn(arg, i);          // the caller takes part
    std::unique_lock<std::mutex> lk(p->m);
    // every helper has woken for this generation and left its loop: none can still be reading fn / arg / n when the next run sets them
    p->done.wait(lk, [&] { return p->busy == 0 && p->acked == (int)p->th.size(); });
}

} // namespace nvca
