// pool_driver.cpp -- test infrastructure: the product's WorkPool (csrc/work_pool.cpp) under ThreadSanitizer: many runs of varying
// size back to back (the hand-over between runs is where a late helper could pick up the next run's function), every index
// handled exactly once, destruction with helpers parked.
#include <atomic>
#include <cstdio>
#include <vector>
namespace nvca {
struct WorkPool;
WorkPool *work_pool_create(int threads);
void work_pool_destroy(WorkPool *p);
void work_pool_run(WorkPool *p, int n, void (*fn)(void *arg, int i), void *arg);
}
struct Arg { std::vector<std::atomic<int>> *hits; int tag; std::atomic<long long> *sum; };
int main()
{
    using namespace nvca;
    long long expect = 0;
    std::atomic<long long> sum{0};
    for (int threads : {0, 1, 3, 7}) {
        WorkPool *p = work_pool_create(threads);
        for (int run = 0; run < 3000; run++) {
            const int n = (run * 7) % 41;                 // 0 .. 40: below and above the serial threshold
            std::vector<std::atomic<int>> hits(n);
            for (auto &h : hits) h.store(0);
            Arg a{&hits, run, &sum};
            work_pool_run(p, n, [](void *v, int i) { Arg *g = (Arg *)v; (*g->hits)[i].fetch_add(1); g->sum->fetch_add(g->tag + i); }, &a);
            for (int i = 0; i < n; i++) { if (hits[i].load() != 1) { fprintf(stderr, "index %d of run %d handled %d times\n", i, run, hits[i].load()); return 1; } expect += run + i; }
        }
        work_pool_destroy(p);
    }
    if (sum.load() != expect) { fprintf(stderr, "sum mismatch\n"); return 1; }
    printf("pool ok %lld\n", expect);
    return 0;
}
