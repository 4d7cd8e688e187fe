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

static constexpr int kRootsHdr = 8;           // words of the tracker's root-list header: [0] entries, [1] overflowed, [2 .. 4] the label walks' "must never happen" record

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
// The tiles (256 x kCclTileRows pixels) that hold motion history are also LISTED, per slot, by the first of their segments that
// finds some: `tiles` = [mark per tile][list: a slot's tiles][count per slot, two sets].  A mark equal to this launch's tick
// means "listed already" (marks are never cleared: the tick moves on); the counts of tick t live in set t & 1, and the pixel
// pass of tick t clears set (t + 1) & 1, which nobody reads or writes during tick t.  The component kernels then walk the live
// tiles only, evenly spread over their blocks (k_ccl_tile was bound by the blocks that drew three live tiles out of eight).
struct TileList { int *mark, *list, *cnt_now, *cnt_next; int per_slot; };
__device__ __forceinline__ TileList tile_list(int *tiles, int w, int h, int batch, int tick)
{
    TileList t;
    t.per_slot = ((w + 255) / 256) * ((h + 7) / 8);
    t.mark = tiles; t.list = tiles + (size_t)t.per_slot * batch;
    int *cnt = tiles + 2 * (size_t)t.per_slot * batch;
    t.cnt_now = cnt + (tick & 1) * batch; t.cnt_next = cnt + ((tick + 1) & 1) * batch;
    return t;
}
__device__ __forceinline__ void segment_flag(uint8_t *__restrict__ flags, int w, int h, bool any, int *__restrict__ tiles, int tick)
{
    const bool hit = __ballot(any) != 0ull;
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0 && seg < seg_per_row(w)) {
        flags[((size_t)blockIdx.z * h + blockIdx.y) * seg_per_row(w) + seg] = hit ? 1 : 0;
        if (hit) {
            if ((blockIdx.y & 15) == 0) atomicAdd(segment_counts(flags, w, h) + blockIdx.z, 1);       // an estimate is all that is asked
            const TileList t = tile_list(tiles, w, h, gridDim.z, tick);
            const int local = (blockIdx.y / 8) * seg_per_row(w) + seg;
            if (atomicExch(&t.mark[(size_t)blockIdx.z * t.per_slot + local], tick) != tick)
                t.list[(size_t)blockIdx.z * t.per_slot + atomicAdd(&t.cnt_now[blockIdx.z], 1)] = local;
        }
    }
}
__device__ __forceinline__ void pixel_pass_clears(int *__restrict__ out, int *__restrict__ roots, int *__restrict__ tiles, int w, int h, int tick)
{
    // the component kernels' counters start at zero: cleared here, by the kernel in front of them, instead of by memset launches of their own
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < kRootsHdr) {
        if (blockIdx.z == 0) { roots[threadIdx.x] = 0; if (threadIdx.x < 2) out[threadIdx.x] = 0; }
        if (threadIdx.x == 0) tile_list(tiles, w, h, gridDim.z, tick).cnt_next[blockIdx.z] = 0;
    }
}

__global__ __launch_bounds__(256) void k_trk_pixel(const TrkSlot *__restrict__ slots, int w, int h, uint8_t *__restrict__ flags, int *__restrict__ out, int *__restrict__ roots, int *__restrict__ tiles, int tick)
{
    pixel_pass_clears(out, roots, tiles, w, h, tick);
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
    segment_flag(flags, w, h, any, tiles, tick);
}

// vectorised variant: w % 4 == 0, 16-byte aligned frame rows
__global__ __launch_bounds__(256) void k_trk_pixel4(const TrkSlot *__restrict__ slots, int w, int h, uint8_t *__restrict__ flags, int *__restrict__ out, int *__restrict__ roots, int *__restrict__ tiles, int tick)
{
    pixel_pass_clears(out, roots, tiles, w, h, tick);
    const TrkSlot s = slots[blockIdx.z];
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    bool any = false;
    if (x4 < w) {
        const uint4 px = *(const uint4 *)(s.src + (size_t)y * s.sstride + (size_t)x4 * 4);
        const size_t o = (size_t)y * w + x4;
        const int g0 = gray4(px.x), g1 = gray4(px.y), g2 = gray4(px.z), g3 = gray4(px.w);
        if (s.has_prev) {
            const unsigned pv = *(const unsigned *)(s.prev + o);
            const float4 m0 = *(const float4 *)(s.mhi + o);
            float4 m = m0;
            const int d0 = g0 - (int)(pv & 255), d1 = g1 - (int)((pv >> 8) & 255), d2 = g2 - (int)((pv >> 16) & 255), d3 = g3 - (int)(pv >> 24);
            m.x = (d0 < 0 ? -d0 : d0) > s.threshold ? s.ts : (m.x < s.delbound ? 0.f : m.x);
            m.y = (d1 < 0 ? -d1 : d1) > s.threshold ? s.ts : (m.y < s.delbound ? 0.f : m.y);
            m.z = (d2 < 0 ? -d2 : d2) > s.threshold ? s.ts : (m.z < s.delbound ? 0.f : m.z);
            m.w = (d3 < 0 ? -d3 : d3) > s.threshold ? s.ts : (m.w < s.delbound ? 0.f : m.w);
            // (stores only of what changed: on a mostly static scene the history stays zero and the gray values stay what they were --
            // 14 bytes a pixel become 9)
            if (__float_as_uint(m.x) != __float_as_uint(m0.x) || __float_as_uint(m.y) != __float_as_uint(m0.y) ||
                __float_as_uint(m.z) != __float_as_uint(m0.z) || __float_as_uint(m.w) != __float_as_uint(m0.w)) *(float4 *)(s.mhi + o) = m;
            any = m.x != 0.f || m.y != 0.f || m.z != 0.f || m.w != 0.f;
            const unsigned gv = (unsigned)g0 | ((unsigned)g1 << 8) | ((unsigned)g2 << 16) | ((unsigned)g3 << 24);
            if (gv != pv) *(unsigned *)(s.prev + o) = gv;
        } else
            *(unsigned *)(s.prev + o) = (unsigned)g0 | ((unsigned)g1 << 8) | ((unsigned)g2 << 16) | ((unsigned)g3 << 24);
    }
    segment_flag(flags, w, h, any, tiles, tick);
}

// ---- union-find on pixel indices (labels[i] = parent; roots are self-parented) ----
__device__ __forceinline__ int uf_find(int *labels, int i)
{
    int p = labels[i];
    while (p != i) { i = p; p = labels[i]; }
    return i;
}
// find that leaves its starting node pointing at the root it found (LDS labels of k_ccl_tile).  The store may race with an atomicMin
// of a union on the same word: both values are smaller members of the node's own set, and a union whose atomicMin did not
// find a root goes on to unite with what it displaced -- no link of the set is ever lost, and parents stay below their children.
// Without it the chains of a textured tile (salt-and-pepper timestamps: hundreds of runs in one component) were tens of hops,
// every hop an LDS round trip: k_ccl_tile 59 -> NN us per 8 x 1080p
__device__ __forceinline__ int uf_find_c(int *labels, int i)
{
    const int i0 = i;
    int p = labels[i], hops = 0;
    while (p != i) { i = p; p = labels[i]; hops++; }
    if (hops > 1) labels[i0] = i;
    return i;
}
__device__ __forceinline__ void uf_union_c(int *labels, int a, int b)
{
    for (;;) {
        a = uf_find_c(labels, a); b = uf_find_c(labels, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&labels[b], a);
        if (old == b) return;
        b = old;
    }
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
// the same on the frame-wide labels, with the one thing that must never happen made harmless and visible: a parent word below
// zero (a pixel without motion history, or a word nobody wrote this frame) ends the walk and is counted in dbg[0] (dbg[1]: where,
// dbg[2]: the kernel) -- the host reports it as an internal error instead of the GPU faulting on labels[-1]
__device__ __forceinline__ int uf_find_g(int *labels, int i, int *dbg, int tag)
{
    int p = labels[i];
    while (p != i) {
        if (p < 0) { if (atomicAdd(&dbg[0], 1) == 0) { dbg[1] = i; dbg[2] = tag; } return i; }
        i = p; p = labels[i];
    }
    return i;
}
__device__ __forceinline__ void uf_union_g(int *labels, int a, int b, int *dbg, int tag)
{
    for (;;) {
        a = uf_find_g(labels, a, dbg, tag); b = uf_find_g(labels, b, dbg, tag);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        if (labels[b] < 0) return;                    // (reported by the find above)
        const int old = atomicMin(&labels[b], a);
        if (old == b) return;
        b = old;
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
__device__ __forceinline__ unsigned live_rows(const uint8_t *__restrict__ flags, int w, int h, int y0, int n, int bx = blockIdx.x, int bz = blockIdx.z)
{
    const uint8_t *f = flags + ((size_t)bz * h + y0) * seg_per_row(w) + bx;
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
static constexpr int kCclTileRows = 8;
static constexpr int kCclTileWaves = 4;
// FOLD: the tile also reduces -- bounding box and first seed of every tile-local component, accumulated per local root in LDS (the
// runs are right here; LDS atomics) -- and leaves one record per tile root: its CompAcc at the root's pixel index and an entry
// in the list of tile roots.  Behind the border unions k_ccl_fold merges the records of tile roots that are no longer roots into
// their roots' and k_ccl_emit reports the roots: two kernels over a few thousand list entries instead of three more walks over
// every live pixel (k_ccl_flatten, _reduce, _collect: 87 of the trackers' 180 us of kernels per 8 x 1080p tick).  Labels are then
// only written where the border unions read them (a tile's first / last row and column).
template <bool FOLD>
__global__ __launch_bounds__(256) void k_ccl_tile(const TrkSlot *__restrict__ slots, int *__restrict__ labels, CompAcc *__restrict__ acc, int *__restrict__ roots, int roots_cap,
                                                  int w, int h, const uint8_t *__restrict__ flags, int ntx, int batch, int *__restrict__ tiles, int tick)
{
    __shared__ float m[kCclTileRows][256];
    __shared__ int lab[kCclTileRows * 256];
    __shared__ int a_maxx[FOLD ? kCclTileRows * 256 : 1];
    __shared__ unsigned a_rows[FOLD ? kCclTileRows * 64 : 1];         // the rows a root's component reaches: a byte per root (26 KB of LDS in all: six blocks per CU)
    __shared__ int rowcnt[kCclTileWaves * kCclTileRows + 1];
    const int n = w * h;
    const int tx = threadIdx.x, lane = tx & 63, wave = tx >> 6;
    // a block walks the LIVE tiles blockIdx.x, + gridDim.x, ... of the list the pixel pass left (slot after slot): a block per tile
    // (8 640 for 8 x 1080p, 15 % of them live) spent most of the kernel starting and ending blocks, and a fixed share of all tiles
    // per block was bound by the blocks that drew three live ones
    const TileList tl = tile_list(tiles, w, h, batch, tick);
    int total = 0;
    for (int z = 0; z < batch; z++) total += tl.cnt_now[z];
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
    int tbz = 0, vbase = 0;
    for (int c = tl.cnt_now[0]; v >= vbase + c; c = tl.cnt_now[tbz]) { vbase += c; tbz++; }
    const int local = tl.list[(size_t)tbz * tl.per_slot + (v - vbase)];
    const int tbx = local % ntx, tby = local / ntx;
    {
    const int x = tbx * 256 + tx, y0 = tby * kCclTileRows;
    const bool in = x < w;
    const int rows = min(kCclTileRows, h - y0);
    const TrkSlot s = slots[tbz];
    int *glab = labels + (size_t)tbz * n;
    const unsigned live = live_rows(flags, w, h, y0, rows, tbx, tbz);      // bit per row whose segment holds motion history (asked for together with the values)
    float mv[kCclTileRows];
#pragma unroll
    for (int ry = 0; ry < kCclTileRows; ry++) mv[ry] = (ry < rows && in) ? s.mhi[(size_t)(y0 + ry) * w + x] : 0.f;       // (a segment without motion history holds zeros)
    __syncthreads();                                  // the previous tile's last LDS reads are over
    unsigned seedbits = 0;
#pragma unroll
    for (int ry = 0; ry < kCclTileRows; ry++) {
        if (ry >= rows) break;
        const float v = mv[ry];
        m[ry][tx] = v;
        if (__float_as_int(v) == __float_as_int(s.ts) && v != 0.f) seedbits |= 1u << ry;
    }
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
        if (lane == 0 && link_l) uf_union_c(lab, idx, idx - 1);                       // runs are cut at wave boundaries
        if (ry > 0) {
            const float u = m[ry - 1][tx];
            if (u != 0.f && joined(v, u, s.seg)) {
                // redundant when the left neighbour already ties the two rows together: v~l, l~ul, ul~u
                const float ul = tx > 0 ? m[ry - 1][tx - 1] : 0.f;
                const bool tied = link_l && ul != 0.f && joined(l, ul, s.seg) && joined(u, ul, s.seg);
                if (!tied) uf_union_c(lab, idx, idx - 256);
            }
        }
    }
    __syncthreads();
    int root[kCclTileRows];                           // this column's pixels: their tile-local roots (-1: no motion history)
#pragma unroll
    for (int ry = 0; ry < kCclTileRows; ry++) root[ry] = (ry < rows && lab[ry * 256 + tx] >= 0) ? uf_find_c(lab, ry * 256 + tx) : -1;
    if (!FOLD) {
#pragma unroll
        for (int ry = 0; ry < kCclTileRows; ry++) {
            if (ry >= rows || !((live >> ry) & 1u) || !in) continue;
            const int r = root[ry];
            glab[(size_t)(y0 + ry) * w + x] = r < 0 ? -1 : (y0 + (r >> 8)) * w + tbx * 256 + (r & 255);
        }
        continue;
    }
    __syncthreads();                                  // every find is done: the parent words and the values become accumulators
    int *a_minx = lab, *a_seed = (int *)&m[0][0];
#pragma unroll
    for (int ry = 0; ry < kCclTileRows; ry++) { const int i = ry * 256 + tx; a_minx[i] = 0x7fffffff; a_seed[i] = 0x7fffffff; a_maxx[i] = -1; }
    a_rows[tx] = 0; a_rows[256 + tx] = 0;
    __syncthreads();
#pragma unroll
    for (int ry = 0; ry < kCclTileRows; ry++) {
        const int r = root[ry];                       // (wave-wide shuffles: every lane takes part)
        const int rl = __shfl_up(r, 1), rr = __shfl_down(r, 1);
        const unsigned sl = (unsigned)__shfl_up((int)seedbits, 1);
        if (r < 0) continue;
        const bool start = lane == 0 || rl != r, end = lane == 63 || rr != r;       // a run of the component inside this wave starts / ends here
        if (start) { atomicMin(&a_minx[r], tx); atomicOr(&a_rows[r >> 2], (1u << ry) << ((r & 3) * 8)); }
        if (end) atomicMax(&a_maxx[r], tx);
        if (((seedbits >> ry) & 1u) && (start || !((sl >> ry) & 1u))) atomicMin(&a_seed[r], ry * 256 + tx);     // the first seed of a run of seeds
    }
    __syncthreads();
    // the tile's roots: a list entry and a record each
    unsigned long long rm[kCclTileRows];
#pragma unroll
    for (int ry = 0; ry < kCclTileRows; ry++) {
        rm[ry] = __ballot(root[ry] == ry * 256 + tx);
        if (lane == 0) rowcnt[wave * kCclTileRows + ry] = __popcll(rm[ry]);
    }
    __syncthreads();
    if (tx < 64) {                                    // exclusive prefix over the 32 (wave, row) counts by wave 0's lanes
        const int c = tx < kCclTileWaves * kCclTileRows ? rowcnt[tx] : 0;
        int incl = c;
        for (int d = 1; d < 32; d <<= 1) { const int o = __shfl_up(incl, d); if ((tx & 63) >= d) incl += o; }
        const int tot = __shfl(incl, kCclTileWaves * kCclTileRows - 1);
        int base = 0;
        if (tx == 0) {
            base = tot ? atomicAdd(&roots[0], tot) : 0;
            if (base + tot > roots_cap) { roots[1] = 1; base = -1; }        // the list is full: the host re-runs the frame set on the per-pixel kernels
            rowcnt[kCclTileWaves * kCclTileRows] = base;
        }
        if (tx < kCclTileWaves * kCclTileRows) rowcnt[tx] = incl - c;
    }
    __syncthreads();
    const int base = rowcnt[kCclTileWaves * kCclTileRows];
#pragma unroll
    for (int ry = 0; ry < kCclTileRows; ry++) {
        if (ry >= rows || !((live >> ry) & 1u) || !in) continue;
        const int r = root[ry], i = ry * 256 + tx;
        const int g = r < 0 ? -1 : (y0 + (r >> 8)) * w + tbx * 256 + (r & 255);
        if (ry == 0 || ry == rows - 1 || tx == 0 || tx == 255 || x == w - 1) glab[(size_t)(y0 + ry) * w + x] = g;      // where k_ccl_border reads
        if (r == i) {
            glab[(size_t)(y0 + ry) * w + x] = g;
            CompAcc c;
            c.minx = tbx * 256 + a_minx[i]; c.maxx = tbx * 256 + a_maxx[i]; c.miny = y0 + ry; c.maxy = y0 + 31 - __clz((a_rows[i >> 2] >> ((i & 3) * 8)) & 0xffu);
            c.seed = a_seed[i] == 0x7fffffff ? 0x7fffffff : (y0 + (a_seed[i] >> 8)) * w + tbx * 256 + (a_seed[i] & 255);
            c.pad = 0;
            acc[(size_t)tbz * n + g] = c;
            if (base >= 0) roots[kRootsHdr + base + rowcnt[wave * kCclTileRows + ry] + __popcll(rm[ry] & ((1ull << lane) - 1ull))] = tbz * n + g;
        }
    }
    }
    }   // live tiles
}

// tile roots that the border unions hung under another root hand their boxes and seeds on to it
__global__ __launch_bounds__(256) void k_ccl_fold(int *__restrict__ labels, CompAcc *__restrict__ acc, const int *__restrict__ roots, int roots_cap, int n, int *dbg, int batch)
{
    if (roots[1]) return;                             // the list overflowed (entries are missing): the host re-runs the frames on the per-pixel kernels
    int cnt = roots[0];
    if (cnt > roots_cap) cnt = roots_cap;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < cnt; e += gridDim.x * 256) {
        const int ent = roots[kRootsHdr + e], slot = ent / n, i = ent - slot * n;
        int *lab = labels + (size_t)slot * n;
        if (ent < 0 || slot >= batch) { if (atomicAdd(&dbg[0], 1) == 0) { dbg[1] = ent; dbg[2] = 3; } continue; }
        const int R = uf_find_g(lab, i, dbg, 4);
        if (R == i) continue;
        const CompAcc c = acc[(size_t)slot * n + i];
        CompAcc *t = &acc[(size_t)slot * n + R];
        atomicMin(&t->minx, c.minx); atomicMax(&t->maxx, c.maxx); atomicMin(&t->miny, c.miny); atomicMax(&t->maxy, c.maxy); atomicMin(&t->seed, c.seed);
    }
}
__global__ __launch_bounds__(256) void k_ccl_emit(const TrkSlot *__restrict__ slots, const int *__restrict__ labels, const CompAcc *__restrict__ acc, const int *__restrict__ roots, int roots_cap,
                                                  int n, int *__restrict__ out, int cap, int batch)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && roots[2]) atomicOr(&out[1], 2);      // a label walk met a word it must never meet: the host reports an internal error
    if (roots[1]) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&out[1], 1); return; }      // list overflow: the host falls back
    int cnt = roots[0];
    if (cnt > roots_cap) cnt = roots_cap;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < cnt; e += gridDim.x * 256) {
        const int ent = roots[kRootsHdr + e], slot = ent / n, i = ent - slot * n;
        if (ent < 0 || slot >= batch) { atomicOr(&out[1], 2); continue; }       // (an entry nobody wrote: never an index -- the host reports an internal error)
        if (labels[(size_t)slot * n + i] != i) continue;
        const CompAcc c = acc[(size_t)slot * n + i];
        if (c.seed == 0x7fffffff) continue;
        const int area = (c.maxx - c.minx + 1) * (c.maxy - c.miny + 1);      // TRK/gstnubotracker.cpp:171-200 (see k_ccl_collect)
        if (!(area > slots[slot].min_area && (long long)area < slots[slot].max_area)) continue;
        const int k = atomicAdd(&out[0], 1);
        if (k < cap) {
            int *o = out + 2 + (size_t)k * 6;
            o[0] = slot; o[1] = c.seed; o[2] = c.minx; o[3] = c.miny; o[4] = c.maxx - c.minx + 1; o[5] = c.maxy - c.miny + 1;
        }
    }
}

// the links that cross a tile's top row or its left column, as unions on the global labels (same redundancy rule: the links it
// leans on are made by the neighbouring thread of this kernel or inside a tile)
__global__ __launch_bounds__(256) void k_ccl_border(const TrkSlot *__restrict__ slots, int *__restrict__ labels, int w, int h, int batch, int *__restrict__ tiles, int tick, int *dbg)
{
    const TileList tl = tile_list(tiles, w, h, batch, tick);
    const int ntx = seg_per_row(w), tx = threadIdx.x;
    int total = 0;
    for (int z = 0; z < batch; z++) total += tl.cnt_now[z];
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        int bz = 0, vbase = 0;
        for (int c = tl.cnt_now[0]; v >= vbase + c; c = tl.cnt_now[bz]) { vbase += c; bz++; }
        const int local = tl.list[(size_t)bz * tl.per_slot + (v - vbase)];
        const int bx = local % ntx, by = local / ntx;
        const TrkSlot s = slots[bz];
        int *lab = labels + (size_t)bz * w * h;
        const int x = bx * 256 + tx, y0 = by * kCclTileRows;
        // (a tile that is not listed holds zeros only: no link out of it, and none into it either -- the other side's values are read here)
        if (y0 > 0 && x < w) {
            const int i = y0 * w + x;
            const float v0 = s.mhi[i];
            if (v0 != 0.f) {
                const float u = s.mhi[i - w];
                if (u != 0.f && joined(v0, u, s.seg)) {
                    const float l = x > 0 ? s.mhi[i - 1] : 0.f, ul = x > 0 ? s.mhi[i - w - 1] : 0.f;
                    const bool tied = l != 0.f && joined(v0, l, s.seg) && ul != 0.f && joined(l, ul, s.seg) && joined(u, ul, s.seg);
                    if (!tied) uf_union_g(lab, i, i - w, dbg, 1);
                }
            }
        }
        if (bx > 0 && tx < kCclTileRows && y0 + tx < h) {
            const int i = (y0 + tx) * w + bx * 256;
            const float v0 = s.mhi[i], l = s.mhi[i - 1];
            if (v0 != 0.f && l != 0.f && joined(v0, l, s.seg)) uf_union_g(lab, i, i - 1, dbg, 2);
        }
    }
}


__global__ __launch_bounds__(256) void k_ccl_flatten(int *__restrict__ labels, CompAcc *__restrict__ acc, int w, int h, const uint8_t *__restrict__ flags, int kCclRows, int *dbg)
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
        const int r = uf_find_g(lab, i, dbg, 5);
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
                                                     int *__restrict__ out /* [0]=count, [1]: flags, then 6 ints per comp */, int cap, int kCclRows, const int *dbg)
{
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && dbg[0]) atomicOr(&out[1], 2);
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
                    int *out, int cap, bool run_ccl, uint8_t *flags, int order, int *roots, int roots_cap, int mode, int *tiles, int tick)
{
    // mode 0: pixel pass + the folded component path; 1: the per-pixel component kernels (NVCA_TRK_FOLD=0); 2: those alone, on the
    // motion history the pixel pass of an earlier launch left (the root list of mode 0 overflowed)
    const TrkSlot *slots = (const TrkSlot *)d_slots;
    dim3 gp(((w + 3) / 4 + 255) / 256, h, batch);
    if (mode != 2) {
        if (vec4) NVCA_LAUNCH(k_trk_pixel4, gp, dim3(256), 0, st, slots, w, h, flags, out, roots, tiles, tick);
        else NVCA_LAUNCH(k_trk_pixel, gp, dim3(256), 0, st, slots, w, h, flags, out, roots, tiles, tick);
    }
    if (!run_ccl) return;
    static const int rows = [] { const char *e = getenv("NVCA_CCL_ROWS"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 16 ? v : kCclRowsDefault; }();
    static const int rows_uf = [] { const char *e = getenv("NVCA_CCL_ROWS_UF"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 16 ? v : kCclRowsUf; }();
    dim3 g2((w + 255) / 256, (h + rows - 1) / rows, batch), g3((w + 255) / 256, (h + rows_uf - 1) / rows_uf, batch);
    dim3 gt((w + 255) / 256, (h + kCclTileRows - 1) / kCclTileRows, batch);
    const int ntiles = (int)(gt.x * gt.y * gt.z);
    const int tile_blocks = ntiles < 2048 ? ntiles : 2048;          // 256 CUs x 4 resident blocks (32 KB of LDS each) x 2: blocks beyond the live tiles' number end at once
    if (mode == 0) {
        NVCA_LAUNCH(k_ccl_tile<true>, dim3(tile_blocks), dim3(256), 0, st, slots, labels, (CompAcc *)acc, roots, roots_cap, w, h, (const uint8_t *)flags, (int)gt.x, batch, tiles, tick);
        NVCA_LAUNCH(k_ccl_border, dim3(tile_blocks), dim3(256), 0, st, slots, labels, w, h, batch, tiles, tick, roots + 2);
        NVCA_LAUNCH(k_ccl_fold, dim3(128), dim3(256), 0, st, labels, (CompAcc *)acc, (const int *)roots, roots_cap, w * h, roots + 2, batch);
        NVCA_LAUNCH(k_ccl_emit, dim3(128), dim3(256), 0, st, slots, (const int *)labels, (const CompAcc *)acc, (const int *)roots, roots_cap, w * h, out, cap, batch);
        return;
    }
    NVCA_LAUNCH(k_ccl_tile<false>, dim3(tile_blocks), dim3(256), 0, st, slots, labels, (CompAcc *)acc, roots, roots_cap, w, h, (const uint8_t *)flags, (int)gt.x, batch, tiles, tick);
    NVCA_LAUNCH(k_ccl_border, dim3(tile_blocks), dim3(256), 0, st, slots, labels, w, h, batch, tiles, tick, roots + 2);
    NVCA_LAUNCH(k_ccl_flatten, g3, dim3(256), 0, st, labels, (CompAcc *)acc, w, h, (const uint8_t *)flags, rows_uf, roots + 2);
    // order (Switches::trk_order): -1: decided per frame on the device
    NVCA_LAUNCH(k_ccl_reduce, g2, dim3(256), 0, st, slots, (const int *)labels, (CompAcc *)acc, w, h, (const uint8_t *)flags, order, rows);
    NVCA_LAUNCH(k_ccl_collect, g2, dim3(256), 0, st, slots, (const int *)labels, (const CompAcc *)acc, w, h, (const uint8_t *)flags, out, cap, rows, (const int *)(roots + 2));
}


} // namespace nvca
