// kernels_tracker.hip -- NuboTracker's per-frame pixel work on gfx950:
// replaces, inside gst_nubo_tracker_process (TRK/gstnubotracker.cpp:339-421),
//   cvtColor(BGRA->gray) :356, absdiff :361, threshold :364,
//   updateMotionHistory :368, segmentMotion :376
// (calcMotionGradient :372 has no observable effect: its outputs are never read).
//
// K7 k_trk_pixel   one streaming pass: gray, |gray-prev| > thr, MHI update, prev <- gray.
// K8 segmentMotion as connected-component labelling: cvSegmentMotion flood-fills, in
//    raster order, from every unlabelled pixel equal to (float)timestamp, joining
//    4-neighbours whose MHI values differ by at most seg_thresh.  The relation is
//    symmetric, so the filled regions are exactly the connected components of that
//    graph which contain a seed, ordered by their first seed pixel.  Zero pixels (which
//    OpenCV swaps for a huge sentinel) only ever connect to each other and hold no seed.
//    k_ccl_tile / k_ccl_border / k_ccl_flatten: lock-free union-find on pixel indices, first inside 256 x 16 tiles out of LDS;
//    k_ccl_reduce: bounding box + first seed per root (atomics once per horizontal run);
//    k_ccl_collect: roots that own a seed -> component list (host sorts by first seed).
// All HBM-bound streaming / atomic work; no MFMA.
#include "nvca_internal.h"

namespace nvca {

__device__ __forceinline__ int gray4(unsigned px)
{
    return (int)((px & 255) * 1868 + ((px >> 8) & 255) * 9617 + ((px >> 16) & 255) * 4899 + 8192) >> 14;
}

// 4 pixels per thread; w4 = ceil(w / 4); rows packed (prev / mhi pitch == w)
// Does a 256-pixel row segment (one wave of the pixel pass = one block of the component kernels) hold any motion history?  One
// byte per segment.  Live segments of every 16th row are also counted per slot (int counters behind the flag bytes, zeroed
// before the launch): a scene that moves everywhere is walked differently by k_ccl_reduce.
__device__ __forceinline__ int seg_per_row(int w) { return (w + 255) / 256; }
__device__ __forceinline__ int *segment_counts(const uint8_t *flags, int w, int h) { return (int *)(flags + (((size_t)seg_per_row(w) * h * gridDim.z + 63) & ~(size_t)63)); }
__device__ __forceinline__ void segment_flag(uint8_t *__restrict__ flags, int w, int h, bool any)
{
    const bool hit = __ballot(any) != 0ull;
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0 && seg < seg_per_row(w)) {
        flags[((size_t)blockIdx.z * h + blockIdx.y) * seg_per_row(w) + seg] = hit ? 1 : 0;
        if (hit && (blockIdx.y & 15) == 0) atomicAdd(segment_counts(flags, w, h) + blockIdx.z, 1);       // an estimate is all that is asked
    }
}

__global__ __launch_bounds__(256) void k_trk_pixel(const TrkSlot *__restrict__ slots, int w, int h, uint8_t *__restrict__ flags)
{
    const TrkSlot s = slots[blockIdx.z];
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    bool any = false;
    const uint8_t *row = s.src + (size_t)y * s.sstride + (size_t)x4 * 4;
    const size_t o = (size_t)y * w + x4;
    const int n = x4 >= w ? 0 : (w - x4 < 4 ? w - x4 : 4);
    for (int k = 0; k < n; k++) {
        const unsigned px = (unsigned)row[k * 4] | ((unsigned)row[k * 4 + 1] << 8) | ((unsigned)row[k * 4 + 2] << 16);
        const int g = gray4(px);
        if (s.has_prev) {
            const int d = g - (int)s.prev[o + k];
            const bool moved = (d < 0 ? -d : d) > s.threshold;           // absdiff + THRESH_BINARY
            const float m = s.mhi[o + k];
            const float v = moved ? s.ts : (m < s.delbound ? 0.f : m);   // cvUpdateMotionHistory
            s.mhi[o + k] = v;
            any = any || v != 0.f;
        }
        s.prev[o + k] = (uint8_t)g;
    }
    segment_flag(flags, w, h, any);
}

// vectorised variant: w % 4 == 0, 16-byte aligned frame rows
__global__ __launch_bounds__(256) void k_trk_pixel4(const TrkSlot *__restrict__ slots, int w, int h, uint8_t *__restrict__ flags)
{
    const TrkSlot s = slots[blockIdx.z];
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    bool any = false;
    if (x4 < w) {
        const uint4 px = *(const uint4 *)(s.src + (size_t)y * s.sstride + (size_t)x4 * 4);
        const size_t o = (size_t)y * w + x4;
        const int g0 = gray4(px.x), g1 = gray4(px.y), g2 = gray4(px.z), g3 = gray4(px.w);
        if (s.has_prev) {
            const unsigned pv = *(const unsigned *)(s.prev + o);
            float4 m = *(const float4 *)(s.mhi + o);
            const int d0 = g0 - (int)(pv & 255), d1 = g1 - (int)((pv >> 8) & 255), d2 = g2 - (int)((pv >> 16) & 255), d3 = g3 - (int)(pv >> 24);
            m.x = (d0 < 0 ? -d0 : d0) > s.threshold ? s.ts : (m.x < s.delbound ? 0.f : m.x);
            m.y = (d1 < 0 ? -d1 : d1) > s.threshold ? s.ts : (m.y < s.delbound ? 0.f : m.y);
            m.z = (d2 < 0 ? -d2 : d2) > s.threshold ? s.ts : (m.z < s.delbound ? 0.f : m.z);
            m.w = (d3 < 0 ? -d3 : d3) > s.threshold ? s.ts : (m.w < s.delbound ? 0.f : m.w);
            *(float4 *)(s.mhi + o) = m;
            any = m.x != 0.f || m.y != 0.f || m.z != 0.f || m.w != 0.f;
        }
        *(unsigned *)(s.prev + o) = (unsigned)g0 | ((unsigned)g1 << 8) | ((unsigned)g2 << 16) | ((unsigned)g3 << 24);
    }
    segment_flag(flags, w, h, any);
}

// ---- union-find on pixel indices (labels[i] = parent; roots are self-parented) ----
__device__ __forceinline__ int uf_find(int *labels, int i)
{
    int p = labels[i];
    while (p != i) { i = p; p = labels[i]; }
    return i;
}
__device__ __forceinline__ void uf_union(int *labels, int a, int b)
{
    for (;;) {
        a = uf_find(labels, a); b = uf_find(labels, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&labels[b], a);     // hang the larger root under the smaller
        if (old == b) return;
        b = old;                                      // someone re-parented b meanwhile: retry from there
    }
}
__device__ __forceinline__ bool joined(float a, float b, float seg)
{   // Diff32fC1 with lo = up = seg_thresh: -seg <= a - b <= seg
    const float d = a - b;
    return -seg <= d && d <= seg;
}

// the component kernels run 256 pixels of a row per block: one flagged segment
__device__ __forceinline__ bool segment_live(const uint8_t *__restrict__ flags, int w, int h, int y)
{
    return flags[((size_t)blockIdx.z * h + y) * seg_per_row(w) + blockIdx.x] != 0;
}
// which of the rows y0 .. y0 + n - 1 (n <= 16) of this block's 256-pixel column hold motion history: bit per row.  The flag bytes are
// requested TOGETHER, ahead of the walk (one round trip): asked for row by row, behind an `if (dead) continue`, they were a
// chain of dependent global loads that made up most of every component kernel on a mostly static scene (five kernels, 4 352
// to 69 k blocks each)
__device__ __forceinline__ unsigned live_rows(const uint8_t *__restrict__ flags, int w, int h, int y0, int n)
{
    const uint8_t *f = flags + ((size_t)blockIdx.z * h + y0) * seg_per_row(w) + blockIdx.x;
    const int nseg = seg_per_row(w);
    unsigned char b[16];
#pragma unroll
    for (int k = 0; k < 16; k++) b[k] = k < n ? f[(size_t)k * nseg] : 0;
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) m |= (b[k] ? 1u : 0u) << k;
    return m;
}
// A block of the component kernels walks kCclRows rows of its 256-pixel column: most segments of a frame hold no motion and
// are skipped on their flag byte, and a grid of one block per segment (69 k blocks for 8 x 1080p) cost 30 us per kernel in
// block dispatch alone -- five kernels a tick.
static constexpr int kCclRowsDefault = 8;
// the two union-find kernels walk fewer: a thread's unions / finds are chains of dependent global round trips, one row after the
// other (measured on 8 x 1080p: 1 / 2 / 4 / 8 rows -> trackers alone 0.261 / 0.250 / 0.259 / 0.274 ms per tick)
static constexpr int kCclRowsUf = 2;

// ---- labelling, first inside tiles of 256 x 16 pixels out of LDS, then across the tiles' borders ------------------------------
// A block takes a tile: the 16 rows' motion history goes to LDS, labels start as horizontal runs (within a wave every pixel points
// at the first pixel of its maximal run of joined neighbours: ballot + count-leading-zeros), the runs are united across the
// wave boundaries and from row to row by the same lock-free union-find as before -- but on LDS words, whose round trips cost a
// hundred cycles instead of a global atomic's thousands -- and every pixel leaves with the GLOBAL index of its tile-local root
// (the root of a set is its smallest index, and row-major order inside a tile is row-major order in the frame: the tile-local
// root is the smallest pixel of the tile's part of the component).  k_ccl_border then unites across the tiles' top rows and left
// columns only -- a sixteenth of the rows, a 256th of the columns -- on labels whose chains are as long as the number of tiles
// a component spans, not the number of rows.  (Round 3: run labels per wave, then every vertical link of the frame as a global
// union: k_ccl_merge + k_ccl_flatten were ~90 us of dependent global round trips per 8 x 1080p.)
static constexpr int kCclTileRows = 16;
__global__ __launch_bounds__(256) void k_ccl_tile(const TrkSlot *__restrict__ slots, int *__restrict__ labels, int w, int h, const uint8_t *__restrict__ flags)
{
    __shared__ float m[kCclTileRows][256];
    __shared__ int lab[kCclTileRows * 256];
    const TrkSlot s = slots[blockIdx.z];
    int *glab = labels + (size_t)blockIdx.z * w * h;
    const int tx = threadIdx.x, lane = tx & 63, x = blockIdx.x * 256 + tx, y0 = blockIdx.y * kCclTileRows;
    const bool in = x < w;
    const int rows = min(kCclTileRows, h - y0);
    const unsigned live = live_rows(flags, w, h, y0, rows);       // bit per row of the tile whose segment holds motion history (block-uniform)
    if (!live) return;                                // nothing but zeros here: no labels are written, and nobody will read any
    for (int ry = 0; ry < rows; ry++) m[ry][tx] = (((live >> ry) & 1u) && in) ? s.mhi[(size_t)(y0 + ry) * w + x] : 0.f;
    __syncthreads();
    for (int ry = 0; ry < rows; ry++) {
        const float v = m[ry][tx], l = tx > 0 ? m[ry][tx - 1] : 0.f;
        const bool link = v != 0.f && l != 0.f && joined(v, l, s.seg);
        const unsigned long long starts = ~__ballot(link) | 1ull;                   // lanes that begin a run inside this wave
        const unsigned long long upto = starts & (~0ull >> (63 - lane));            // ... at or left of this lane
        const int head = 63 - __clzll((long long)upto);
        lab[ry * 256 + tx] = v != 0.f ? ry * 256 + tx - (lane - head) : -1;
    }
    __syncthreads();
    for (int ry = 0; ry < rows; ry++) {
        const int idx = ry * 256 + tx;
        const float v = m[ry][tx];
        if (v == 0.f) continue;
        const float l = tx > 0 ? m[ry][tx - 1] : 0.f;
        const bool link_l = l != 0.f && joined(v, l, s.seg);
        if (lane == 0 && link_l) uf_union(lab, idx, idx - 1);                       // runs are cut at wave boundaries
        if (ry > 0) {
            const float u = m[ry - 1][tx];
            if (u != 0.f && joined(v, u, s.seg)) {
                // redundant when the left neighbour already ties the two rows together: v~l, l~ul, ul~u
                const float ul = tx > 0 ? m[ry - 1][tx - 1] : 0.f;
                const bool tied = link_l && ul != 0.f && joined(l, ul, s.seg) && joined(u, ul, s.seg);
                if (!tied) uf_union(lab, idx, idx - 256);
            }
        }
    }
    __syncthreads();
    if (!in) return;
    for (int ry = 0; ry < rows; ry++) {
        if (!((live >> ry) & 1u)) continue;
        const int idx = ry * 256 + tx;
        int g = -1;
        if (lab[idx] >= 0) { const int r = uf_find(lab, idx); g = (y0 + (r >> 8)) * w + blockIdx.x * 256 + (r & 255); }
        glab[(size_t)(y0 + ry) * w + x] = g;
    }
}

// the links that cross a tile's top row or its left column, as unions on the global labels (same redundancy rule: the links it
// leans on are made by the neighbouring thread of this kernel or inside a tile)
__global__ __launch_bounds__(256) void k_ccl_border(const TrkSlot *__restrict__ slots, int *__restrict__ labels, int w, int h, const uint8_t *__restrict__ flags)
{
    const TrkSlot s = slots[blockIdx.z];
    int *lab = labels + (size_t)blockIdx.z * w * h;
    const int tx = threadIdx.x, x = blockIdx.x * 256 + tx, y0 = blockIdx.y * kCclTileRows;
    if (y0 > 0 && x < w && segment_live(flags, w, h, y0)) {
        const int i = y0 * w + x;
        const float v = s.mhi[i];
        if (v != 0.f) {
            const float u = s.mhi[i - w];             // (a segment without motion history holds zeros: no link into it)
            if (u != 0.f && joined(v, u, s.seg)) {
                const float l = x > 0 ? s.mhi[i - 1] : 0.f, ul = x > 0 ? s.mhi[i - w - 1] : 0.f;
                const bool tied = l != 0.f && joined(v, l, s.seg) && ul != 0.f && joined(l, ul, s.seg) && joined(u, ul, s.seg);
                if (!tied) uf_union(lab, i, i - w);
            }
        }
    }
    if (blockIdx.x > 0 && tx < kCclTileRows && y0 + tx < h && segment_live(flags, w, h, y0 + tx)) {
        const int i = (y0 + tx) * w + blockIdx.x * 256;
        const float v = s.mhi[i], l = s.mhi[i - 1];
        if (v != 0.f && l != 0.f && joined(v, l, s.seg)) uf_union(lab, i, i - 1);
    }
}


__global__ __launch_bounds__(256) void k_ccl_flatten(int *__restrict__ labels, CompAcc *__restrict__ acc, int w, int h, const uint8_t *__restrict__ flags, int kCclRows)
{
    const int n = w * h;
    int *lab = labels + (size_t)blockIdx.z * n;
    CompAcc *ac = acc + (size_t)blockIdx.z * n;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const int yb = blockIdx.y * kCclRows;
    const unsigned live = live_rows(flags, w, h, yb, min(kCclRows, h - yb));
    for (int y = yb, y1 = min(y + kCclRows, h); y < y1; y++) {
        if (!((live >> (y - yb)) & 1u)) continue;
        const int i = y * w + x;
        if (lab[i] < 0) continue;
        const int r = uf_find(lab, i);
        lab[i] = r;
        if (r == i) { CompAcc c; c.minx = c.miny = 0x7fffffff; c.maxx = c.maxy = -1; c.seed = 0x7fffffff; c.pad = 0; ac[i] = c; }
    }
}

// bounding box / first seed of a root: min / max atomics, attempted only when the value read (at device scope, past the
// per-CU cache) would still be improved -- minima only fall and maxima only rise, so a stale read costs an atomic, never a
// result.  A moving scene is one component as large as the frame: without the test every run of every row queues up on the
// same five words (12 ms per 720p frame instead of 0.1).
// what one lane reports for a root: candidates for the box and the first seed (INT_MAX / -1: nothing to report for that word).
// The five words are read first -- independent loads, one round trip -- then only the atomics that still improve are issued.
__device__ __forceinline__ int acc_peek(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void acc_report(CompAcc *c, int minx, int maxx, int y, bool row, int seed)
{
    const int cur_minx = acc_peek(&c->minx), cur_miny = acc_peek(&c->miny), cur_maxx = acc_peek(&c->maxx), cur_maxy = acc_peek(&c->maxy), cur_seed = acc_peek(&c->seed);
    if (minx < cur_minx) atomicMin(&c->minx, minx);
    if (maxx > cur_maxx) atomicMax(&c->maxx, maxx);
    if (row && y < cur_miny) atomicMin(&c->miny, y);
    if (row && y > cur_maxy) atomicMax(&c->maxy, y);
    if (seed < cur_seed) atomicMin(&c->seed, seed);
}

__global__ __launch_bounds__(256) void k_ccl_reduce(const TrkSlot *__restrict__ slots, const int *__restrict__ labels,
                                                    CompAcc *__restrict__ acc, int w, int h, const uint8_t *__restrict__ flags, int order, int kCclRows)
{
    const TrkSlot s = slots[blockIdx.z];
    const int nseg = seg_per_row(w);
    if (order < 0) order = 2 * segment_counts(flags, w, h)[blockIdx.z] > nseg * ((h + 15) / 16) ? 1 : 0;      // most of the frame holds motion: outside in
    const int n = w * h;
    const int *lab = labels + (size_t)blockIdx.z * n;
    CompAcc *ac = acc + (size_t)blockIdx.z * n;
    // When most of the frame moves, rows are visited from the outside in (0, last, 1, last - 1, ...): the extreme rows of a
    // frame-sized component arrive first, and everything that follows fails the "would it still improve" test instead of
    // queueing up (0.76 -> 0.46 ms per 4 x 720p); otherwise top to bottom, which is kinder to memory (0.49 -> 0.42 ms per 8 x 1080p)
    const int bx = (int)blockIdx.x;
    const int vb = blockIdx.y * kCclRows;
    unsigned live = 0;
    {   // the rows' flag bytes in one round trip (see live_rows)
        unsigned char b[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int v = vb + k, by = (order & 1) ? ((v & 1) ? h - 1 - (v >> 1) : (v >> 1)) : v;
            b[k] = (k < kCclRows && v < h) ? flags[((size_t)blockIdx.z * h + by) * nseg + bx] : 0;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) live |= (b[k] ? 1u : 0u) << k;
    }
    for (int v = vb, v1 = min(v + kCclRows, h); v < v1; v++) {
    const int by = (order & 1) ? ((v & 1) ? h - 1 - (v >> 1) : (v >> 1)) : v;
    if (!((live >> (v - vb)) & 1u)) continue;
    const int x = bx * 256 + threadIdx.x, y = by, lane = threadIdx.x & 63;
    const int i = y * w + x;
    const int r = x < w ? lab[i] : -1;                // labels are final roots after k_ccl_flatten
    bool start_l = false, end_r = false, seed0 = false;
    if (r >= 0) {
        // a neighbour across a segment border may sit in a segment without labels: its motion history (0 there) is asked first
        const bool same_l = x > 0 && s.mhi[i - 1] != 0.f && lab[i - 1] == r;
        start_l = !same_l;                            // a horizontal run of the component starts / ends here
        end_r = !(x + 1 < w && s.mhi[i + 1] != 0.f && lab[i + 1] == r);
        if (__float_as_int(s.mhi[i]) == __float_as_int(s.ts))
            seed0 = !(same_l && __float_as_int(s.mhi[i - 1]) == __float_as_int(s.ts));
    }
    // A wave holds 64 consecutive pixels of one row.  The components that draw crowds are few and large (a moving scene is one
    // component as large as the frame), so the roots of the wave's first lanes with something to report -- two rounds -- are
    // handled by one lane each: the leftmost run start, the rightmost run end, the row and the leftmost seed of that root in
    // the wave, taken from ballots.  Whatever is left (small components: little company) reports lane by lane, in parallel.
    const unsigned long long m_l = __ballot(start_l), m_r = __ballot(end_r), m_s = __ballot(seed0);
    unsigned long long todo = m_l | m_r | m_s;
    const int rounds = order ? 2 : 1;                         // a crowded frame: two roots by proxy; otherwise the first one only
    for (int round = 0; round < rounds && todo; round++) {    // wave-uniform
        const int leader = __ffsll((long long)todo) - 1;
        const int r0 = __shfl(r, leader);
        const unsigned long long g = __ballot(r == r0);
        if (lane == leader) {
            const unsigned long long gl = g & m_l, gr = g & m_r, gs = g & m_s;
            const int xb = x - lane;
            acc_report(&ac[r0], gl ? xb + __ffsll((long long)gl) - 1 : 0x7fffffff, gr ? xb + 63 - __clzll((long long)gr) : -1, y, gl != 0,
                       gs ? i - lane + __ffsll((long long)gs) - 1 : 0x7fffffff);
        }
        todo &= ~g;
    }
    if ((todo >> lane) & 1ull) acc_report(&ac[r], start_l ? x : 0x7fffffff, end_r ? x : -1, y, start_l, seed0 ? i : 0x7fffffff);
    }
}

__global__ __launch_bounds__(256) void k_ccl_collect(const TrkSlot *__restrict__ slots, const int *__restrict__ labels,
                                                     const CompAcc *__restrict__ acc, int w, int h, const uint8_t *__restrict__ flags,
                                                     int *__restrict__ out /* [0]=count, then 6 ints per comp */, int cap, int kCclRows)
{
    const int slot = blockIdx.z, n = w * h;
    const int *lab = labels + (size_t)slot * n;
    const CompAcc *ac = acc + (size_t)slot * n;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const int yb = blockIdx.y * kCclRows;
    const unsigned live = live_rows(flags, w, h, yb, min(kCclRows, h - yb));
    for (int y = yb, y1 = min(y + kCclRows, h); y < y1; y++) {
        if (!((live >> (y - yb)) & 1u)) continue;
        const int i = y * w + x;
        if (lab[i] != i) continue;
        const CompAcc c = ac[i];
        if (c.seed == 0x7fffffff) continue;
        {   // TRK/gstnubotracker.cpp:171-200: boxes outside the area window are erased and never merged with anything
            const int area = (c.maxx - c.minx + 1) * (c.maxy - c.miny + 1);
            if (!(area > slots[slot].min_area && (long long)area < slots[slot].max_area)) continue;
        }
        const int k = atomicAdd(&out[0], 1);
        if (k < cap) {
            int *o = out + 2 + (size_t)k * 6;
            o[0] = slot; o[1] = c.seed; o[2] = c.minx; o[3] = c.miny; o[4] = c.maxx - c.minx + 1; o[5] = c.maxy - c.miny + 1;
        }
    }
}

void launch_tracker(hipStream_t st, const void *d_slots, int batch, int w, int h, bool vec4, int *labels, void *acc,
                    int *out, int cap, bool run_ccl, uint8_t *flags, int order)
{
    const TrkSlot *slots = (const TrkSlot *)d_slots;
    dim3 gp(((w + 3) / 4 + 255) / 256, h, batch);
    if (vec4) NVCA_LAUNCH(k_trk_pixel4, gp, dim3(256), 0, st, slots, w, h, flags);
    else NVCA_LAUNCH(k_trk_pixel, gp, dim3(256), 0, st, slots, w, h, flags);
    if (!run_ccl) return;
    static const int rows = [] { const char *e = getenv("NVCA_CCL_ROWS"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 16 ? v : kCclRowsDefault; }();
    static const int rows_uf = [] { const char *e = getenv("NVCA_CCL_ROWS_UF"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 16 ? v : kCclRowsUf; }();
    dim3 g2((w + 255) / 256, (h + rows - 1) / rows, batch), g3((w + 255) / 256, (h + rows_uf - 1) / rows_uf, batch);
    dim3 gt((w + 255) / 256, (h + kCclTileRows - 1) / kCclTileRows, batch);
    NVCA_LAUNCH(k_ccl_tile, gt, dim3(256), 0, st, slots, labels, w, h, (const uint8_t *)flags);
    NVCA_LAUNCH(k_ccl_border, gt, dim3(256), 0, st, slots, labels, w, h, (const uint8_t *)flags);
    NVCA_LAUNCH(k_ccl_flatten, g3, dim3(256), 0, st, labels, (CompAcc *)acc, w, h, (const uint8_t *)flags, rows_uf);
    // order (Switches::trk_order): -1: decided per frame on the device
    NVCA_LAUNCH(k_ccl_reduce, g2, dim3(256), 0, st, slots, (const int *)labels, (CompAcc *)acc, w, h, (const uint8_t *)flags, order, rows);
    NVCA_LAUNCH(k_ccl_collect, g2, dim3(256), 0, st, slots, (const int *)labels, (const CompAcc *)acc, w, h, (const uint8_t *)flags, out, cap, rows);
}


} // namespace nvca
