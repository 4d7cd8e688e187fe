// ranges_driver.cpp -- test infrastructure: the bookkeeping that decides which caller host memory may reach the HIP runtime as a raw
// pointer and which streams nvca_host_unregister drains before pages are released (csrc/host_ranges.h), under ASan + UBSan.
// The rule: direct copies only out of / into a range that is registered NOW; a range that was released is bounced like
// any other memory; unregister waits for every stream that carried a copy of the range, not only the context's own.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../nubomedia-vca_amd/csrc/host_ranges.h"
using nvca::HostRangeTable;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "ranges_driver: check failed at line %d: %s\n", __LINE__, #c); return 1; } } while (0)
int main()
{
    std::vector<unsigned char> heap(1 << 20);
    unsigned char *a = heap.data() + 4096, *b = heap.data() + 300000;
    HostRangeTable t;
    // nothing registered: every copy is bounced
    CHECK(!t.registered(a, 100) && !t.note_copy(a, 100, 0));
    CHECK(t.add(a, 200000));
    CHECK(!t.add(a + 1000, 10));                       // overlaps a live range
    CHECK(!t.add(a - 10, 20));
    CHECK(t.add(b, 50000));
    CHECK(!t.add(nullptr, 10) && !t.add(a, 0));
    // copies inside a live range are direct and remembered per stream; a copy that leaves the range is not registered memory
    CHECK(t.note_copy(a, 200000, 0));                  // the context's stream
    CHECK(t.note_copy(a + 6000, 6000, 10));            // the copy stream (host-frame chunks of a submitted batch)
    CHECK(t.note_copy(a + 12000, 6000, 8));            // the second batch's lane
    CHECK(!t.note_copy(a + 199999, 2, 3));             // straddles the end
    CHECK(!t.note_copy(a - 1, 2, 3));
    CHECK(!t.note_copy(a + 250000, 16, 3));            // the gap between the two ranges
    CHECK(t.note_copy(b + 49999, 1, 9));               // the trackers' lane
    // unregister: the pointer the range was registered with; every stream that carried a copy must be drained first
    bool found = true;
    CHECK(t.remove(a + 8, &found) == 0 && !found);     // not a registered base pointer
    const uint64_t s = t.remove(a, &found);
    CHECK(found && s == ((1ull << 0) | (1ull << 10) | (1ull << 8)));      // NOT only the context's stream
    // released memory is no longer registered memory: whatever lands there next (a small numpy image, two tests later) is bounced
    CHECK(!t.registered(a, 8051) && !t.note_copy(a + 50000, 8051, 0));
    CHECK(t.was_registered(a + 50000, 8051) && !t.was_registered(heap.data(), 16));
    // the other range is untouched; its streams are its own
    CHECK(t.registered(b, 50000));
    CHECK(t.remove(b, &found) == (1ull << 9) && found);
    CHECK(t.live.empty() && t.retired.size() == 2);
    // the same memory registered again starts with no streams to wait for
    CHECK(t.add(a, 4096) && t.remove(a, &found) == 0 && found);
    // the history is bounded
    for (int i = 0; i < 1000; i++) { CHECK(t.add(a, 64)); (void)t.remove(a, &found); }
    CHECK(t.retired.size() <= HostRangeTable::kRetiredMax);
    printf("ranges ok\n");
    return 0;
}
