// kernels_cascade.hip -- the Haar cascade evaluator (scale-cascade variant,
// flags without CV_HAAR_SCALE_IMAGE): replaces the per-window loop of
// HaarDetectObjects_ScaleCascade_Invoker + cvRunHaarClassifierCascadeSum
// (OpenCV 2.4 haar.cpp) behind cascade->detectMultiScale at
// FACE/kmsfacedetect.cpp:809-811.
//
// Launches per batch of frames:
//  K5  k_band        (batches with >= 540 bands in flight, api.cpp) one workgroup per row of tiles of a
//                    scale, walking it left to right: per tile (<= 32 x 24 windows, a window per thread) the integral
//                    samples the windows touch are staged, compacted, in LDS; window variance and
//                    stage 0 for every window; OpenCV's adaptive x step (ix += result != 0 ? 1 : 2)
//                    in closed form -- window j is visited iff the run of stage-0 rejects
//                    immediately left of it in its row has even length (the parity is carried from
//                    tile to tile); visited survivors are queued in LDS and run through the next
//                    stages, re-compacted after every stage; whoever survives stage deep_stage-1
//                    is appended to the deep list.
//  K5a k_stage0 +    (smaller batches) the same as a pre-pass over all windows with global reads
//  K5b k_tile        (reject bits + variance normaliser per window) followed by one workgroup per
//                    tile.  k_strip: the older variant with global gathers (NVCA_TILES=0; the fallback when a plan has no tiles).
//  K5c k_deep        one workgroup per surviving window, one stump per thread (the long stages have
//                    33..213 stumps); after the first late stage the window's samples are staged in LDS.
//  K6  k_group       cv::groupRectangles per frame.
// Stump records are geometry-independent tables per (cascade, factor), wave-uniform in K5/K5a/K5b
// (scalar loads) and per-lane in K5c.  No MFMA: integer rect sums, f32 products, f64 stage sums --
// exactly the reference's arithmetic (compiled with -ffp-contract=off).
#include "nvca_internal.h"

namespace nvca {

// Wave-uniform table records are read through the constant address space: the compiler then issues scalar loads
// (s_load) for them even though the kernels also store to global memory.  The tables are never written by a kernel.
typedef const __attribute__((address_space(4))) TStumpRec CTStumpRec;
// a small plain record at a wave-uniform address, read dword by dword through the constant address space (s_load)
template <class T> __device__ __forceinline__ T load_const(const T *p)
{
    static_assert(sizeof(T) % 4 == 0, "dword records only");
    typedef const __attribute__((address_space(4))) int CInt;
    CInt *q = (CInt *)p;
    T r;
    int *d = (int *)&r;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); i++) d[i] = q[i];
    return r;
}

__device__ __forceinline__ int ldsum(const int *__restrict__ sum, unsigned idx) { return sum[idx]; }

// Squared-pixel sum of the variance window from the squared integral, as the f64 OpenCV computes (every operand and every
// partial result is an integer below 2^53, so the f64 chain is exact and equals the integer result).  The plane pair is a
// u32 low-word plane and a u8 high-byte plane (the values stay below 2^40 whenever the i32 sum plane is valid); when the
// window's sum is known to be below 2^32 the low words alone give it, modulo 2^32.
__device__ __forceinline__ double window_sqsum(const unsigned *__restrict__ sql, const uint8_t *__restrict__ sqh, bool lo_only,
                                               unsigned e0, unsigned e1, unsigned e2, unsigned e3)
{
    if (lo_only) return (double)(unsigned)(sql[e0] - sql[e1] - sql[e2] + sql[e3]);
    const unsigned long long q0 = ((unsigned long long)sqh[e0] << 32) | sql[e0], q1 = ((unsigned long long)sqh[e1] << 32) | sql[e1];
    const unsigned long long q2 = ((unsigned long long)sqh[e2] << 32) | sql[e2], q3 = ((unsigned long long)sqh[e3] << 32) | sql[e3];
    return (double)q0 - (double)q1 - (double)q2 + (double)q3;
}

// feature value of one stump on one window (v) against its threshold: returns the vote.  Records hold corner columns /
// rows relative to the window (geometry-independent tables); the plane offset is row * pitch + column.
template <bool PAIR, class Rec, bool UNI = false>
__device__ __forceinline__ double stump_vote(const int *__restrict__ sum, unsigned off, int pitch, double vnf, Rec &f)
{
    auto rs = [&](int q) {
        const unsigned r0 = off + (unsigned)(f.y0[q] * pitch), r1 = off + (unsigned)(f.y1[q] * pitch);
        return ldsum(sum, r0 + (unsigned)f.x0[q]) - ldsum(sum, r0 + (unsigned)f.x1[q]) - ldsum(sum, r1 + (unsigned)f.x0[q]) +
               ldsum(sum, r1 + (unsigned)f.x1[q]);
    };
    const int s0 = rs(0);
    const int s1 = rs(1);
    const double t = f.thr * vnf;                       // node->threshold * variance_norm_factor
    double v;
    if (PAIR) {
        const float fs = (float)s0 * f.w[0] + (float)s1 * f.w[1];     // SSE2 path: f32 add
        v = (double)fs;
    } else {
        v = (double)((float)s0 * f.w[0]);
        v += (double)((float)s1 * f.w[1]);
        if ((f.nrect & 255) == 3) {
            const int s2 = rs(2);
            v += (double)((float)s2 * f.w[2]);
        }
    }
    double a0 = f.a0, a1 = f.a1;
    if (UNI) asm("" : "+s"(a0), "+s"(a1));      // wave-uniform record: both votes stay in scalar registers
    return v >= t ? a1 : a0;
}

// one stage on one window per lane; recs are wave-uniform (scalar loads)
template <bool PAIR>
__device__ __forceinline__ bool eval_stage(const int *__restrict__ sum, unsigned off, int pitch, double vnf,
                                           CTStumpRec *recs, int count, float stage_thr)
{
    double stage_sum = 0.0;
    for (int j = 0; j < count; j++) stage_sum += stump_vote<PAIR, CTStumpRec, true>(sum, off, pitch, vnf, recs[j]);
    return !(stage_sum < (double)stage_thr);
}

__device__ __forceinline__ bool run_stage(const int *__restrict__ sum, unsigned off, int pitch, double vnf,
                                          CTStumpRec *recs, const StageRec &st, int pair_policy)
{
    if (pair_policy && (st.flags & 1)) return eval_stage<true>(sum, off, pitch, vnf, recs + st.first, st.count, st.thr);
    return eval_stage<false>(sum, off, pitch, vnf, recs + st.first, st.count, st.thr);
}

// block -> (slot, local index): 1-D grid, frame-major (few integral planes live at a time); within a
// frame consecutive local indices alternate over 8 contiguous chunks, i.e. blocks that share an XCD
// (b % 8) walk one contiguous part of the scan (speed only)
__device__ __forceinline__ bool xcd_chunk_index(int nlocal, int &slot, int &idx)
{
    const int per_frame = ((nlocal + 7) / 8) * 8;
    slot = blockIdx.x / per_frame;
    const int l = blockIdx.x - slot * per_frame;
    const int chunk = per_frame / 8;
    idx = (l & 7) * chunk + (l >> 3);
    return idx < nlocal;
}

// ---- K5a: variance + stage 0 for every window --------------------------------
__global__ __launch_bounds__(256) void k_stage0(CascadeArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int slot, bidx;
    if (!xcd_chunk_index((a.ntasks + 3) / 4, slot, bidx)) return;
    const int t = __builtin_amdgcn_readfirstlane(bidx * 4 + wave);
    if (t >= a.ntasks) return;
    const unsigned task = a.tasks[t];
    const int s = task >> 20, iy = (task >> 7) & 8191, k = task & 127;
    const ScaleRec &sc = a.scales[s];
    const int ix = k * 64 + lane;
    const bool active = ix < sc.endX;
    const int *__restrict__ sum = a.sum + (size_t)slot * a.sum_slot + sc.plane_off;
    // squared integral: a u32 low-word plane and a u8 high-byte plane per slot, see k_integral
    const unsigned *__restrict__ sql = (const unsigned *)a.sqsum + (size_t)slot * 2 * a.sum_slot + sc.plane_off;
    const uint8_t *__restrict__ sqh = (const uint8_t *)((const unsigned *)a.sqsum + (size_t)slot * 2 * a.sum_slot + a.sum_slot) + sc.plane_off;
    bool pass0 = false;
    double vnf = 1.;
    if (active) {
        const unsigned off = (unsigned)(a.pos[sc.ypos_off + iy] * sc.pitch + a.pos[sc.xpos_off + ix]);
        const unsigned e0 = off + sc.eq[0], e1 = off + sc.eq[1], e2 = off + sc.eq[2], e3 = off + sc.eq[3];
        const int ws = sum[e0] - sum[e1] - sum[e2] + sum[e3];
        const double mean = (double)ws * sc.inv_area;
        vnf = window_sqsum(sql, sqh, sc.sq32 != 0, e0, e1, e2, e3);
        vnf = vnf * sc.inv_area - mean * mean;
        vnf = vnf >= 0. ? sqrt(vnf) : 1.;
        pass0 = run_stage(sum, off, sc.pitch, vnf, (CTStumpRec *)sc.trecs, a.stages[0], a.pair_policy);
    }
    const unsigned long long fb = __ballot(active && !pass0);
    const size_t o = (size_t)slot * a.ntasks + t;
    if (lane == 0) a.failbits[o] = fb;
    a.vnf[o * 64 + lane] = vnf;
}

// visited by OpenCV's adaptive scan?  row_bits: the row's stage-0 reject words
__device__ __forceinline__ bool visited(const unsigned long long *__restrict__ row_bits, int ix)
{
    int d = 0, pos = ix;
    while (pos > 0) {
        const int p = pos - 1, b = p & 63;
        const unsigned long long m = row_bits[p >> 6] << (63 - b);       // bit p at the MSB
        const int ones = (~m == 0ull) ? 64 : __clzll((long long)~m);
        const int lim = b + 1;
        d += ones < lim ? ones : lim;
        if (ones < lim) break;
        pos -= lim;
    }
    return !(d & 1);
}

// ---- general cascades: tree-structured weak classifiers and / or tilted features ----------------------------------------
// cvRunHaarClassifierCascadeSum's general branch: per weak classifier a walk idx = sum < t ? left : right from the root to
// a leaf, whose value is the vote; a feature's rectangles read the integral image or, for a tilted feature, the tilted
// integral.  Window per lane, global reads (these cascades run on the part detectors' small working images; the LDS tile
// machinery above is built around upright stumps).  The stage loop is wave-uniform, so the root of every weak classifier is
// a scalar record; only the nodes below it are per-lane.
typedef const __attribute__((address_space(4))) GNodeRec CGNodeRec;
template <class Rec>
__device__ __forceinline__ double gen_node_sum(const int *__restrict__ pl, unsigned off, int pitch, const Rec &n, bool pair)
{
    auto rs = [&](int q) {
        return pl[off + (unsigned)(n.dy[q][0] * pitch + n.dx[q][0])] - pl[off + (unsigned)(n.dy[q][1] * pitch + n.dx[q][1])] -
               pl[off + (unsigned)(n.dy[q][2] * pitch + n.dx[q][2])] + pl[off + (unsigned)(n.dy[q][3] * pitch + n.dx[q][3])];
    };
    const int s0 = rs(0), s1 = rs(1);
    if (pair) return (double)((float)s0 * n.w[0] + (float)s1 * n.w[1]);       // SSE2 path of two-rectangle stump stages
    double v = (double)((float)s0 * n.w[0]);
    v += (double)((float)s1 * n.w[1]);
    if ((n.flags & 255) == 3) v += (double)((float)rs(2) * n.w[2]);
    return v;
}
// The weak classifiers of a stage are independent of one another up to their votes: four roots are evaluated per step -- their
// 32 to 48 gathers are in flight together instead of one classifier's at a time -- then each walk is finished and the votes are
// added in stage order (OpenCV's order: the f64 sum is the same).
__device__ __forceinline__ bool gen_stage(const CascadeArgs &a, const int *__restrict__ sum, const int *__restrict__ tilt, unsigned off, int pitch,
                                          double vnf, const GNodeRec *recs, const StageRec &st)
{
    const bool pair = a.pair_policy && a.stump_based && (st.flags & 1);
    double stage_sum = 0.0;
    auto finish = [&](int base, int idx) {                                  // below the root the lanes of a wave part ways
        while (idx > 0) {
            const GNodeRec &n = recs[base + idx];
            const double sn = gen_node_sum((n.flags & 256) ? tilt : sum, off, pitch, n, false);
            idx = sn < (double)n.thr * vnf ? n.left : n.right;
        }
        return (double)a.galpha[-idx];
    };
    int j = 0;
    for (; j + 4 <= st.count; j += 4) {
        int base[4], idx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            base[u] = a.gcls_first[st.first + j + u];
            CGNodeRec &root = ((CGNodeRec *)recs)[base[u]];
            const double s = gen_node_sum((root.flags & 256) ? tilt : sum, off, pitch, root, pair);
            idx[u] = s < (double)root.thr * vnf ? root.left : root.right;   // node->threshold * variance_norm_factor
        }
#pragma unroll
        for (int u = 0; u < 4; u++) stage_sum += finish(base[u], idx[u]);
    }
    for (; j < st.count; j++) {
        const int base = a.gcls_first[st.first + j];
        CGNodeRec &root = ((CGNodeRec *)recs)[base];
        const double s = gen_node_sum((root.flags & 256) ? tilt : sum, off, pitch, root, pair);
        stage_sum += finish(base, s < (double)root.thr * vnf ? root.left : root.right);
    }
    return !(stage_sum < (double)st.thr);
}

__global__ __launch_bounds__(256) void k_gen_stage0(CascadeArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int slot, bidx;
    if (!xcd_chunk_index((a.ntasks + 3) / 4, slot, bidx)) return;
    const int t = __builtin_amdgcn_readfirstlane(bidx * 4 + wave);
    if (t >= a.ntasks) return;
    const unsigned task = a.tasks[t];
    const int s = task >> 20, iy = (task >> 7) & 8191, k = task & 127;
    const ScaleRec &sc = a.scales[s];
    const int ix = k * 64 + lane;
    const bool active = ix < sc.endX;
    const int *__restrict__ sum = a.sum + (size_t)slot * a.sum_slot + sc.plane_off;
    const int *__restrict__ tilt = a.tilted ? a.tilted + (size_t)slot * a.sum_slot + sc.plane_off : sum;
    const unsigned *__restrict__ sql = (const unsigned *)a.sqsum + (size_t)slot * 2 * a.sum_slot + sc.plane_off;
    const uint8_t *__restrict__ sqh = (const uint8_t *)((const unsigned *)a.sqsum + (size_t)slot * 2 * a.sum_slot + a.sum_slot) + sc.plane_off;
    bool pass0 = false;
    double vnf = 1.;
    if (active) {
        const unsigned off = (unsigned)(a.pos[sc.ypos_off + iy] * sc.pitch + a.pos[sc.xpos_off + ix]);
        const unsigned e0 = off + sc.eq[0], e1 = off + sc.eq[1], e2 = off + sc.eq[2], e3 = off + sc.eq[3];
        const int ws = sum[e0] - sum[e1] - sum[e2] + sum[e3];
        const double mean = (double)ws * sc.inv_area;
        vnf = window_sqsum(sql, sqh, sc.sq32 != 0, e0, e1, e2, e3);
        vnf = vnf * sc.inv_area - mean * mean;
        vnf = vnf >= 0. ? sqrt(vnf) : 1.;
        pass0 = gen_stage(a, sum, tilt, off, sc.pitch, vnf, sc.grecs, a.stages[0]);
    }
    const unsigned long long fb = __ballot(active && !pass0);
    const size_t o = (size_t)slot * a.ntasks + t;
    if (lane == 0) a.failbits[o] = fb;
    a.vnf[o * 64 + lane] = vnf;
}

__global__ __launch_bounds__(256) void k_gen_rest(CascadeArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int slot, bidx;
    if (!xcd_chunk_index((a.ntasks + 3) / 4, slot, bidx)) return;
    const int t = __builtin_amdgcn_readfirstlane(bidx * 4 + wave);
    if (t >= a.ntasks) return;
    const unsigned task = a.tasks[t];
    const int s = task >> 20, iy = (task >> 7) & 8191, k = task & 127;
    const ScaleRec &sc = a.scales[s];
    const int ix = k * 64 + lane;
    const unsigned long long *rb = a.failbits + (size_t)slot * a.ntasks + sc.task_off + (size_t)iy * sc.wpr;
    bool alive = false;
    if (ix < sc.endX && !((rb[k] >> lane) & 1ull)) alive = sc.adaptive ? visited(rb, ix) : true;
    if (!__any(alive)) return;
    const int *__restrict__ sum = a.sum + (size_t)slot * a.sum_slot + sc.plane_off;
    const int *__restrict__ tilt = a.tilted ? a.tilted + (size_t)slot * a.sum_slot + sc.plane_off : sum;
    unsigned off = 0; double vnf = 1.;
    if (alive) {
        off = (unsigned)(a.pos[sc.ypos_off + iy] * sc.pitch + a.pos[sc.xpos_off + ix]);
        vnf = a.vnf[((size_t)slot * a.ntasks + t) * 64 + lane];
    }
    for (int st_i = 1; st_i < a.nstages; st_i++) {           // wave-uniform stage loop: lanes that fell out idle
        if (!__any(alive)) return;
        if (alive) alive = gen_stage(a, sum, tilt, off, sc.pitch, vnf, sc.grecs, a.stages[st_i]);
    }
    const unsigned long long hm = __ballot(alive);
    if (!hm) return;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(a.hits, (unsigned long long)__popcll(hm));
    base = __shfl(base, 0);
    if (alive) {
        const unsigned long long pos = base + __popcll(hm & ((1ull << lane) - 1ull));
        const unsigned key = ((unsigned)s << a.key_ss) | ((unsigned)iy << a.key_sy) | (unsigned)ix;
        if (pos < a.hit_cap) a.hits[1 + pos] = ((unsigned long long)slot << 32) | key;
    }
}

void launch_generic(hipStream_t st, const CascadeArgs &a, int batch, int which)
{
    if (batch <= 0 || a.ntasks <= 0) return;
    const int blocks = (((a.ntasks + 3) / 4 + 7) / 8) * 8;
    if (which == 0) NVCA_LAUNCH(k_gen_stage0, dim3((unsigned)blocks * (unsigned)batch), dim3(256), 0, st, a);
    else NVCA_LAUNCH(k_gen_rest, dim3((unsigned)blocks * (unsigned)batch), dim3(256), 0, st, a);
}

// ---- K5b: stages 1 .. deep_stage-1 on strips ---------------------------------
__global__ __launch_bounds__(256) void k_strip(CascadeArgs a)
{
    __shared__ unsigned short q[2][kStripMaxWin];
    __shared__ double psum[256];
    __shared__ int qn[2];
    __shared__ unsigned gbase_s;

    const int tid = threadIdx.x, lane = tid & 63;
    const int slot = blockIdx.x / a.blocks_per_frame;
    const int sidx = a.order[blockIdx.x - slot * a.blocks_per_frame];
    if (sidx < 0) return;
    const StripRec strip = a.strips[sidx];
    const ScaleRec &sc = a.scales[strip.scale];
    // a strip covers columns [ix0, ix0 + ncols) of nrows scan rows (rows longer than a strip are cut into segments)
    const int endX = strip.ncols, ix0 = strip.ix0, nwin = strip.nrows * endX;
    const int *__restrict__ sum = a.sum + (size_t)slot * a.sum_slot + sc.plane_off;
    CTStumpRec *recs = (CTStumpRec *)sc.trecs;
    const int *__restrict__ xpos = a.pos + sc.xpos_off;
    const int *__restrict__ ypos = a.pos + sc.ypos_off + strip.iy0;
    const unsigned long long *__restrict__ bits = a.failbits + (size_t)slot * a.ntasks + sc.task_off + (size_t)strip.iy0 * sc.wpr;
    const double *__restrict__ vnfp = a.vnf + ((size_t)slot * a.ntasks + sc.task_off + (size_t)strip.iy0 * sc.wpr) * 64;

    if (tid < 2) qn[tid] = 0;
    __syncthreads();

    // adaptive-step reachability + compaction of visited stage-0 survivors
    for (int base = 0; base < nwin; base += 256) {
        const int w = base + tid;
        bool keep = false;
        if (w < nwin) {
            const int r = w / endX, ix = ix0 + (w - r * endX);
            const unsigned long long *rb = bits + (size_t)r * sc.wpr;
            if (!((rb[ix >> 6] >> (ix & 63)) & 1ull)) keep = sc.adaptive ? visited(rb, ix) : true;
        }
        const unsigned long long km = __ballot(keep);
        if (km) {
            int wbase = 0;
            if (lane == 0) wbase = atomicAdd(&qn[0], __popcll(km));
            wbase = __shfl(wbase, 0);
            if (keep) q[0][wbase + __popcll(km & ((1ull << lane) - 1ull))] = (unsigned short)w;
        }
    }

    int cur = 0;
    int last = a.deep_stage < a.nstages ? a.deep_stage : a.nstages;
    for (int s = 1; s < last; s++) {
        __syncthreads();
        const int n = qn[cur];
        if (n == 0) break;
        if (tid == 0) qn[cur ^ 1] = 0;
        __syncthreads();
        const StageRec st = a.stages[s];
        if ((st.flags & 2) && n <= 128) {
            // Few survivors: most lanes would idle while one wave walks the whole stage.  The votes of this stage may be
            // summed in any order (flag bit 1), so spread its stumps over the idle lanes: thread = (window slot i,
            // stump partition p); partition p takes stumps p, p+P, ...; partial sums meet in LDS.
            int lg = 0;
            while ((1 << lg) < n) lg++;
            const int npad = 1 << lg;
            int P = 256 >> lg;
            if (P > st.count) P = st.count;
            const int i = tid & (npad - 1), p = tid >> lg;
            double part = 0.0;
            int w = 0;
            if (i < n && p < P) {
                w = q[cur][i];
                const int r = w / endX, ix = ix0 + (w - r * endX);
                const unsigned off = (unsigned)(ypos[r] * sc.pitch + xpos[ix]);
                const double vnf = vnfp[((size_t)r * sc.wpr + (ix >> 6)) * 64 + (ix & 63)];
                const bool pair = a.pair_policy && (st.flags & 1);
                if (lg >= 6) {               // a wave holds one partition: records stay wave-uniform (scalar loads)
                    const int pu = __builtin_amdgcn_readfirstlane(p);
                    for (int j = pu; j < st.count; j += P)
                        part += pair ? stump_vote<true>(sum, off, sc.pitch, vnf, recs[st.first + j]) : stump_vote<false>(sum, off, sc.pitch, vnf, recs[st.first + j]);
                } else {
                    for (int j = p; j < st.count; j += P)
                        part += pair ? stump_vote<true>(sum, off, sc.pitch, vnf, recs[st.first + j]) : stump_vote<false>(sum, off, sc.pitch, vnf, recs[st.first + j]);
                }
            }
            psum[tid] = part;
            __syncthreads();
            bool pass = false;
            if (tid < n) {
                double tot = 0.0;
                for (int pp = 0; pp < P; pp++) tot += psum[(pp << lg) + tid];
                pass = !(tot < (double)st.thr);
                w = q[cur][tid];
            }
            const unsigned long long pm = __ballot(pass);
            if (pm) {
                int wbase = 0;
                if (lane == 0) wbase = atomicAdd(&qn[cur ^ 1], __popcll(pm));
                wbase = __shfl(wbase, 0);
                if (pass) q[cur ^ 1][wbase + __popcll(pm & ((1ull << lane) - 1ull))] = (unsigned short)w;
            }
        } else
        for (int base = 0; base < n; base += 256) {
            const int i = base + tid;
            bool pass = false; int w = 0;
            if (i < n) {
                w = q[cur][i];
                const int r = w / endX, ix = ix0 + (w - r * endX);
                const unsigned off = (unsigned)(ypos[r] * sc.pitch + xpos[ix]);
                const double vnf = vnfp[((size_t)r * sc.wpr + (ix >> 6)) * 64 + (ix & 63)];
                pass = run_stage(sum, off, sc.pitch, vnf, recs, st, a.pair_policy);
            }
            const unsigned long long pm = __ballot(pass);
            if (pm) {
                int wbase = 0;
                if (lane == 0) wbase = atomicAdd(&qn[cur ^ 1], __popcll(pm));
                wbase = __shfl(wbase, 0);
                if (pass) q[cur ^ 1][wbase + __popcll(pm & ((1ull << lane) - 1ull))] = (unsigned short)w;
            }
        }
        cur ^= 1;
    }
    __syncthreads();
    const int nh = qn[cur];
    if (nh == 0) return;
    // survivors: final candidates if the cascade ends here, otherwise work for k_deep
    unsigned long long *list = last == a.nstages ? a.hits : a.deep;
    const unsigned cap = last == a.nstages ? a.hit_cap : a.deep_cap;
    if (tid == 0) gbase_s = (unsigned)atomicAdd(list, (unsigned long long)nh);
    __syncthreads();
    const unsigned gb = gbase_s;
    for (int i = tid; i < nh; i += 256) {
        const int w = q[cur][i];
        const int r = w / endX, ix = ix0 + (w - r * endX);
        const unsigned key = ((unsigned)strip.scale << a.key_ss) | ((unsigned)(strip.iy0 + r) << a.key_sy) | (unsigned)ix;
        if (gb + i < cap) list[1 + gb + i] = ((unsigned long long)slot << 32) | key;
    }
}

// ---- K5b: stages 1 .. deep_stage-1 on LDS lattice tiles -------------------------------------
// A tile is nx x ny (<= 32 x 24) windows of one scale.  Window origins and scaled rectangle corners of a scale
// fall on a near-lattice, so the tile's windows touch only ~2.7 (n + 20) distinct columns and rows of the sum
// plane whatever the scale.  Those rows x columns are copied, compacted, into LDS once; every rectangle corner
// is then two u16 map look-ups (column index, row offset) and one LDS read, instead of a global gather whose 64
// lanes touch up to 64 different cache lines.  Values and arithmetic are unchanged.
template <bool PAIR, bool UNI = true>
__device__ __forceinline__ double tile_vote(const int *T, const unsigned short *cmap, const unsigned short *rmap,
                                            int xw, int yw, double vnf, CTStumpRec &f)
{
    // the row map holds WORD offsets of the row starts from T, the column map BYTE offsets: a corner address is one shift-add
    auto at = [&](int rw, int cb) { return *(const int *)((const char *)T + ((rw << 2) + cb)); };
    auto rs = [&](int q) {
        const int c0 = cmap[xw + f.x0[q]], c1 = cmap[xw + f.x1[q]];
        const int r0 = rmap[yw + f.y0[q]], r1 = rmap[yw + f.y1[q]];
        return at(r0, c0) - at(r0, c1) - at(r1, c0) + at(r1, c1);
    };
    const int s0 = rs(0);
    const int s1 = rs(1);
    const double t = f.thr * vnf;
    double v;
    if (PAIR) {
        const float fs = (float)s0 * f.w[0] + (float)s1 * f.w[1];
        v = (double)fs;
    } else {
        v = (double)((float)s0 * f.w[0]);
        v += (double)((float)s1 * f.w[1]);
        if ((f.nrect & 255) == 3) {
            const int s2 = rs(2);
            v += (double)((float)s2 * f.w[2]);
        }
    }
    double a0 = f.a0, a1 = f.a1;
    if (UNI) asm("" : "+s"(a0), "+s"(a1));      // wave-uniform record: both votes stay in scalar registers (no dependent load of the selected one)
    return v >= t ? a1 : a0;
}

// A wave-uniform record held whole in scalar registers.  Left to itself the compiler fetches a record piecemeal, each
// piece right before its use (corners; weights; threshold; votes): four dependent scalar-memory round trips per stump on
// the critical path of a latency-bound loop.  One 16-dword and one 8-dword load bring the record in at once.
typedef int int16v __attribute__((ext_vector_type(16)));
typedef int int8v __attribute__((ext_vector_type(8)));
struct SRec { int16v a; int8v b; };
__device__ __forceinline__ SRec load_srec(const TStumpRec *p)
{
    SRec r;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(r.a), "=&s"(r.b) : "s"(p) : "memory");
    return r;
}
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
// LDS by absolute byte address (the tile kernels keep map / sample addresses as integers: no generic-pointer base in the way)
typedef __attribute__((address_space(3))) const unsigned short lds_cu16;
typedef __attribute__((address_space(3))) const int lds_ci32;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"      // the host pass sees 64-bit pointers; device LDS pointers are 32 bits
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(unsigned long long)(lptr_t)p; }
__device__ __forceinline__ int lds_u16(unsigned a) { return *(lds_cu16 *)a; }
// sample at (row start word address r, column byte offset c): one shift-add, one read
__device__ __forceinline__ int lds_sample(int r, int c)
{
    unsigned a;
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a) : "v"(r), "v"(c));
    return *(lds_ci32 *)a;
}
#pragma clang diagnostic pop
// cm / rm: LDS byte addresses of this window's entries in the column / row map (map base + 2 * window offset); the column
// map holds byte offsets of the compacted columns, the row map absolute word addresses of the staged rows
// Sum = double: the vote as OpenCV adds it (f64).  Sum = int: the vote as an integer multiple of 2^vote_exp (StageRec flag bit 2).
template <bool PAIR, class Sum>
__device__ __forceinline__ Sum tile_vote_s(unsigned cm, unsigned rm, double vnf, const SRec &f)
{
    // dwords: x0[3] 0-2 | x1[3] 3-5 | y0[3] 6-8 | y1[3] 9-11 | w[3] 12-14 | nrect + share 15 || thr 0-1 | a0 2-3 | a1 4-5 | a0i 6 | a1i 7
    // share (bits 8.. of dword 15): which pairs a later rectangle shares with rectangle 0 (wave-uniform: scalar branches)
    const int share = f.a[15] >> 8;
    const int ca = lds_u16(cm + 2 * f.a[0]), cb = lds_u16(cm + 2 * f.a[3]);
    const int ra = lds_u16(rm + 2 * f.a[6]), rb = lds_u16(rm + 2 * f.a[9]);
    auto rs = [&](int x0, int x1, int y0, int y1, int sh) {
        int c0 = ca, c1 = cb, r0 = ra, r1 = rb;
        if (!(sh & 2)) { c0 = lds_u16(cm + 2 * x0); c1 = lds_u16(cm + 2 * x1); }
        if (!(sh & 1)) { r0 = lds_u16(rm + 2 * y0); r1 = lds_u16(rm + 2 * y1); }
        return lds_sample(r0, c0) - lds_sample(r0, c1) - lds_sample(r1, c0) + lds_sample(r1, c1);
    };
    const int s0 = lds_sample(ra, ca) - lds_sample(ra, cb) - lds_sample(rb, ca) + lds_sample(rb, cb);
    const int s1 = rs(f.a[1], f.a[4], f.a[7], f.a[10], share);
    const double t = __hiloint2double(f.b[1], f.b[0]) * vnf;
    const float w0 = __int_as_float(f.a[12]), w1 = __int_as_float(f.a[13]);
    double v;
    if (PAIR) {
        const float fs = (float)s0 * w0 + (float)s1 * w1;
        v = (double)fs;
    } else {
        v = (double)((float)s0 * w0);
        v += (double)((float)s1 * w1);
        if ((f.a[15] & 255) == 3) {
            const int s2 = rs(f.a[2], f.a[5], f.a[8], f.a[11], share >> 2);
            v += (double)((float)s2 * __int_as_float(f.a[14]));
        }
    }
    if (sizeof(Sum) == sizeof(int)) return (Sum)(v >= t ? f.b[7] : f.b[6]);
    const double a0 = __hiloint2double(f.b[3], f.b[2]), a1 = __hiloint2double(f.b[5], f.b[4]);
    return (Sum)(v >= t ? a1 : a0);
}
// one stage's sum over stumps j0, j0 + step, ... < count for one window, records wave-uniform
template <bool PAIR, class Sum = double>
__device__ __forceinline__ Sum tile_stage_sum(unsigned cm, unsigned rm, double vnf, const TStumpRec *recs, int j0, int count, int step)
{
    // the table pointer is wave-uniform but comes out of a vector load: hand the asm a scalar copy
    const unsigned long long u = (unsigned long long)recs;
    const unsigned long long ub = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)u);
    const TStumpRec *base = (const TStumpRec *)ub;
    asm("" : "+v"(cm), "+v"(rm));        // keep the two per-window addresses whole: a look-up address is then one shift-add of a scalar
    Sum sum = 0;
    for (int j = j0; j < count; j += step) sum += tile_vote_s<PAIR, Sum>(cm, rm, vnf, load_srec(base + j));
    return sum;
}
// does a window pass the stage?  (all of the stage's stumps on this thread)
__device__ __forceinline__ bool tile_stage_pass(unsigned cm, unsigned rm, double vnf, const TStumpRec *recs, const StageRec &st, bool pair)
{
    if (st.flags & 4) {
        const int sum = pair ? tile_stage_sum<true, int>(cm, rm, vnf, recs + st.first, 0, st.count, 1) : tile_stage_sum<false, int>(cm, rm, vnf, recs + st.first, 0, st.count, 1);
        return sum >= st.thr_i;
    }
    const double sum = pair ? tile_stage_sum<true>(cm, rm, vnf, recs + st.first, 0, st.count, 1) : tile_stage_sum<false>(cm, rm, vnf, recs + st.first, 0, st.count, 1);
    return !(sum < (double)st.thr);
}

// ---- a stump per LANE: the compact record (LStumpRec) arrives as three 16-byte vector loads ------------------------------
struct LRec { int4 a, b, c; };      // a: xx0 yy0 xx1 yy1 | b: xx2 yy2 w0 w1 | c: w2 thr a0 a1
__device__ __forceinline__ LRec load_lrec(const LStumpRec *p)
{
    const int4 *q = (const int4 *)p;
    LRec r; r.a = q[0]; r.b = q[1]; r.c = q[2];
    return r;
}
// one rectangle through the tile's maps: xx / yy hold the two corner columns / rows as byte offsets into the u16 maps
__device__ __forceinline__ int lane_rect(unsigned cm, unsigned rm, unsigned xx, unsigned yy)
{
    const int c0 = lds_u16(cm + (xx & 0xffffu)), c1 = lds_u16(cm + (xx >> 16));
    const int r0 = lds_u16(rm + (yy & 0xffffu)), r1 = lds_u16(rm + (yy >> 16));
    return lds_sample(r0, c0) - lds_sample(r0, c1) - lds_sample(r1, c0) + lds_sample(r1, c1);
}
// The vote of the lane's stump on the lane's window, as the f64 OpenCV adds to the stage sum.  Lanes of one wave may hold
// stumps of different stages: whether the SSE2 pair form applies (a stage of two-rectangle stumps under NVCA_SUM_F32PAIR) is
// a mark in the record (xx2 == 0, yy2 == 1) and selected per lane.
__device__ __forceinline__ double lane_vote(unsigned cm, unsigned rm, double vnf, const LRec &f, int pair_policy)
{
    const int s0 = lane_rect(cm, rm, (unsigned)f.a.x, (unsigned)f.a.y);
    const int s1 = lane_rect(cm, rm, (unsigned)f.a.z, (unsigned)f.a.w);
    const double t = (double)__int_as_float(f.c.y) * vnf;          // node->threshold * variance_norm_factor
    const float p0 = (float)s0 * __int_as_float(f.b.z), p1 = (float)s1 * __int_as_float(f.b.w);
    double v = (double)p0;
    v += (double)p1;
    if (f.b.x) v += (double)((float)lane_rect(cm, rm, (unsigned)f.b.x, (unsigned)f.b.y) * __int_as_float(f.c.x));
    else if (pair_policy && f.b.y == 1) v = (double)(p0 + p1);
    return (double)__int_as_float(v >= t ? f.c.w : f.c.z);
}

#ifdef NVCA_STAMPS
// diagnostic build: thread 0 of the first 64 workgroups leaves s_memtime stamps per tile and phase (64 words per tile, 16 tiles)
#define NVCA_STAMP(a, tile, id) do { if (threadIdx.x == 0 && blockIdx.x < 64 && (tile) < 16 && (id) < 64 && (a).dbg) { unsigned long long t__; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); (a).dbg[((size_t)blockIdx.x * 16 + (tile)) * 64 + (id)] = t__; } } while (0)
#define NVCA_STAMP_VAL(a, tile, id, v) do { if (threadIdx.x == 0 && blockIdx.x < 64 && (tile) < 16 && (id) < 64 && (a).dbg) (a).dbg[((size_t)blockIdx.x * 16 + (tile)) * 64 + (id)] = (unsigned long long)(v); } while (0)
#else
#define NVCA_STAMP(a, tile, id) do { } while (0)
#define NVCA_STAMP_VAL(a, tile, id, v) do { } while (0)
#endif

// LDS carve-up of a tile (tile_lds_bytes() on the host sizes exactly this)
struct TileLds {
    double *acc;                  // [kTileSlots] stage accumulators (viewed as int or double per stage), all zero outside a stage
    unsigned short *q0, *winx, *winy; int *qn; double *vnf_s;
    unsigned short *cmap, *rmap; int *T; int pitchT;
    int tword;                    // absolute LDS word address of T: the row map holds tword + r * pitchT
    unsigned cmA, rmA;            // LDS byte addresses of the maps
};
__device__ __forceinline__ TileLds carve_tile(unsigned char *lds, const TileRec &t)
{
    TileLds L;
    L.acc = (double *)lds;
    L.q0 = (unsigned short *)(lds + kTileSlots * 8);
    L.winx = L.q0 + 2 * kTileSlots; L.winy = L.winx + kTileWin;
    L.qn = (int *)(L.winy + kTileWin);                       // qn[0..2] rotating queue counters, qn[3] = list base, qn[4 ..] stage statistics
    L.vnf_s = (double *)((unsigned char *)L.qn + 64);
    L.cmap = (unsigned short *)(L.vnf_s + kTileSlots);
    L.rmap = L.cmap + ((t.span_x + 3) & ~3);
    L.T = (int *)(L.rmap + ((t.span_y + 3) & ~3));
    L.pitchT = tile_pitch(t.ncol);
    L.tword = (int)(lds_addr(L.T) >> 2);          // < 2^16: the whole of LDS is 160 KiB = 40 Ki words
    L.cmA = lds_addr(L.cmap); L.rmA = lds_addr(L.rmap);
    return L;
}

// Staging a tile = two rounds of global reads.  Round 1 (tile_coords): every coordinate a thread needs -- its entries of
// the tile's column / row lists for the maps, the window origins, the wave's sample-row offsets and the lane's sample
// columns.  Round 2 (tile_commit): the maps are scattered into LDS and the sample rows x columns are copied global -> LDS by
// LDS-DMA (global_load_lds_dword: the source address is per lane -- a gather of the tile's lattice columns -- the
// destination is 64 consecutive words of the row, no registers in between).  A wave takes rows wave, wave + 16, ...; every
// transfer of the tile is in flight before the first one is waited for (register staging kept two rows per wave in
// flight: the copy was a chain of dependent round trips).  The caller waits: s_waitcnt vmcnt(0) + barrier before T is read.
struct TileCoords { int mapc, mapr, wx, wy; unsigned rowb, xcb[4]; };
__device__ __forceinline__ TileCoords tile_coords(const CascadeArgs &a, const TileRec &t, const ScaleRec &sc)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = kTileThreads / 64;
    const unsigned short *__restrict__ cl = a.tcoords + t.col_off, *__restrict__ rl = a.tcoords + t.row_off;
    TileCoords c;
    c.mapc = tid < t.ncol ? (int)cl[tid] : -1;               // ncol, nrow <= 256 < kTileThreads: one list entry per thread
    c.mapr = tid < t.nrow ? (int)rl[tid] : -1;
    c.wx = tid < t.nx ? a.pos[sc.xpos_off + t.ix0 + tid] : 0;
    c.wy = (tid >= 64 && tid < 64 + t.ny) ? a.pos[sc.ypos_off + t.iy0 + tid - 64] : 0;
    const int nmine = t.nrow > wave ? (t.nrow - wave + NW - 1) / NW : 0;
    c.rowb = lane < nmine ? (unsigned)rl[wave + NW * lane] * (unsigned)sc.pitch * 4u : 0u;     // lane j: the wave's j-th sample row
#pragma unroll
    for (int k = 0; k < 4; k++) { const int q = lane + 64 * k; c.xcb[k] = 4u * (unsigned)cl[q < t.ncol ? q : t.ncol - 1]; }
    return c;
}
__device__ __forceinline__ void tile_commit(const CascadeArgs &a, const TileRec &t, const ScaleRec &sc, int slot, const TileLds &L, const TileCoords &c)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = kTileThreads / 64;
    const char *__restrict__ src = (const char *)(a.sum + (size_t)slot * a.sum_slot + sc.plane_off);
    const int nmine = t.nrow > wave ? (t.nrow - wave + NW - 1) / NW : 0;
    const int nk = (t.ncol + 63) >> 6;      // 64-column groups that hold a staged column (wave-uniform)
    for (int j = 0; j < nmine; j++) {
        const unsigned rb = (unsigned)__builtin_amdgcn_readlane((int)c.rowb, j);
        const char *rowp = src + rb;
        int *dst = L.T + (wave + NW * j) * L.pitchT;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k < nk && lane + 64 * k < t.ncol)
                __builtin_amdgcn_global_load_lds((gptr_t)(rowp + c.xcb[k]), (lptr_t)(dst + 64 * k), 4, 0, 0);
    }
    if (tid < 3) L.qn[tid] = 0;          // the three rotating queue counters (qn[3]: list base scratch)
    if (tid < t.nx) L.winx[tid] = (unsigned short)(c.wx - t.x0);
    if (tid >= 64 && tid < 64 + t.ny) L.winy[tid - 64] = (unsigned short)(c.wy - t.y0);
    if (c.mapc >= 0) L.cmap[c.mapc - t.x0] = (unsigned short)(tid * 4);
    if (c.mapr >= 0) L.rmap[c.mapr - t.y0] = (unsigned short)(L.tword + tid * L.pitchT);     // absolute word address of the row: a corner address is one shift-add
}

// the same when every lane that may keep a window sits in ONE wave (a round of at most 64 windows: wave 0 decides them all) and
// *count is zero and nobody else's: positions and count come straight from the ballot -- no returning atomic, one LDS round
// trip less in a round that is a chain of such round trips
__device__ __forceinline__ void queue_push_wave0(bool keep, int w, unsigned short *q, int *count)
{
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    const unsigned long long km = __ballot(keep);
    if (keep) q[__popcll(km & ((1ull << lane) - 1ull))] = (unsigned short)w;
    if (lane == 0) *count = __popcll(km);
}
__device__ __forceinline__ void queue_push(bool keep, int w, unsigned short *q, int *count)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long km = __ballot(keep);
    if (km) {
        int wbase = 0;
        if (lane == 0) wbase = atomicAdd(count, __popcll(km));
        wbase = __shfl(wbase, 0);
        if (keep) q[wbase + __popcll(km & ((1ull << lane) - 1ull))] = (unsigned short)w;
    }
}

// ---- the stages behind stage 0 on the windows queued in q0[0 .. qn[0]) (window id = ry * 32 + rx) -----------------------------
// One round per stage (several stages per round once few windows are left), the queue re-compacted after every round; whoever is
// left after stage last-1 goes to the candidate list (last == nstages: the usual plan -- every stage runs out of the tile, there
// is no late-stage kernel) or to k_deep's list (plans whose tiles cannot hold the late stages' samples).
// A round's work is the grid (windows of the queue) x (stumps of the stage); three ways of dealing it to the workgroup's waves:
//   * more than 32 windows, votes exact in any order (StageRec flag bit 1; bit 2: as 32-bit integers of a common unit): the grid of
//     (64-window groups) x (stumps) goes to the waves as equal contiguous runs, a window per lane, the stump records wave-uniform
//     in scalar registers; a wave adds its run's sum to the window's accumulator in LDS (ds_add_u32 / ds_add_f64 -- every partial
//     sum is exactly representable, so any order and grouping gives the f64 sum OpenCV forms, bit for bit).
//   * at most 32 windows (n): every wave holds ALL n windows, G = 64 / n stumps at a time -- lane = (window l % n, stump l / n) -- and
//     takes every NW-th block of G stumps, records per lane (LStumpRec: three 16-byte loads from L1 / L2).  Lanes stay busy
//     whatever n is: a late stage of 50 .. 213 stumps on a handful of windows is ONE step of the workgroup, not a chain of stumps
//     walked by a handful of lanes.  With few windows left a round takes several stages at once (as many as fit one step of
//     the workgroup: 768 window-stump pairs): the stumps of stages s+1, s+2 .. are evaluated for every window of the queue before
//     it is known whether it passes stage s -- no side effects, and a window goes on iff it passes them all, exactly as if they
//     had run one after the other; the rounds a surviving window waits for (a barrier pair and a round trip of records each)
//     shrink from one per stage to one per 2 .. 7 stages.
//   * votes that may not be re-ordered (neither flag bit): a window per thread, the stage's stumps in OpenCV's order.
// The early stages 1 .. 5 may be walked in the order the PREVIOUS tile of the band found cheapest (stump count per window
// killed, from that tile's entered / passed counts: stat words behind the queue counters, two sets used alternately): a window
// goes on iff it passes all of them, and which of them it fails first is seen by nobody.  Switch "stage_order".
static constexpr int kStatStages = 6;         // stages 0 .. 5 take part in the adaptive order
static constexpr int kPairMax = 32;           // windows up to which a round runs lane = (window, stump)
template <bool VNF_LDS>
__device__ __forceinline__ void tile_stages(const CascadeArgs &a, const TileRec &t, const ScaleRec &sc, int slot, const TileLds &L, int ti = 0, int par = 0)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int *stat_r = L.qn + 4 + par * kStatStages;       // what the previous tile saw
    int *stat_w = L.qn + 4 + (par ^ 1) * kStatStages;       // what this tile sees
    constexpr int NW = kTileThreads / 64;
    const TStumpRec *urecs = sc.trecs;
    const size_t vbase = ((size_t)slot * a.ntasks + sc.task_off + (size_t)t.iy0 * sc.wpr) * 64;
    const double *__restrict__ vnfp = a.vnf + vbase;
    auto vnf_of = [&](int w) {
        if (VNF_LDS) return L.vnf_s[w];
        const int ix = t.ix0 + (w & 31);
        return vnfp[((size_t)(w >> 5) * sc.wpr + (ix >> 6)) * 64 + (ix & 63)];
    };
    int *acci = (int *)L.acc;
    int cur = 0;
    // queue counters rotate over three words: a round reads qn[cin], appends to qn[cout] and clears the third one, which
    // nobody touches during this round and which the next round appends to
    int cin = 0;
    const int last = a.deep_stage < a.nstages ? a.deep_stage : a.nstages;
    const int m = (last < kStatStages ? last : kStatStages) - 1;          // stages 1 .. m: the adaptive prefix
    const bool adapt = a.stage_order && m >= 2;
    unsigned ordpack = 0x54321u;
    if (adapt) {
        // (the word may come from the plan's hint words, which other workgroups write as they go: it is used only if its first m
        // positions name each of the stages 1 .. m once)
        const unsigned o = (unsigned)__builtin_amdgcn_readfirstlane(stat_r[0]);
        unsigned seen = 0;
        for (int q = 0; q < m; q++) seen |= 1u << ((o >> (4 * q)) & 15u);
        if (seen == (2u << m) - 2u) ordpack = o;
    }
    int kpos = 0, prev = 0; unsigned entered = 0;          // wave-uniform; kpos: stages behind stage 0 already walked
    auto stage_at = [&](int kp) { return __builtin_amdgcn_readfirstlane((adapt && kp < m) ? (int)((ordpack >> (4 * kp)) & 15u) : kp + 1); };
    // the record of the round's first stage is requested one round ahead (behind the previous round's barrier it would stand
    // at the head of a chain of dependent round trips that IS the round when few windows are left)
    struct I8 { int v[8]; };
    struct F8 { float v[8]; };
    const int s_first = last > 1 ? stage_at(0) : 0;
    StageRec st = load_const(a.stages + s_first);
    I8 F = load_const((const I8 *)(a.stage_first + s_first));      // first stump of stages s .. s + 7 (the table is padded with INT_MAX); constant indices only: the words stay in scalar registers
    F8 TH = load_const((const F8 *)(a.stage_thr + s_first));       // the thresholds of stages s .. s + 7
    while (kpos < last - 1) {
        const bool in_prefix = adapt && kpos < m;
        const int s = stage_at(kpos);
        __syncthreads();             // queue complete (first pass: tile and maps staged as well)
        NVCA_STAMP(a, ti, 8 + 8 * s);
        const int n = __builtin_amdgcn_readfirstlane(L.qn[cin]);
        if (adapt && tid == 0 && prev) stat_w[prev] |= n;        // what the stage before let through
        if (n == 0) break;
        NVCA_STAMP_VAL(a, ti, 8 + 8 * s + 7, n);
        const int cout = cin == 2 ? 0 : cin + 1;
        if (tid == 0) L.qn[cout == 2 ? 0 : cout + 1] = 0;
        const unsigned short *qi = L.q0 + cur * kTileSlots;
        unsigned short *qo = L.q0 + (cur ^ 1) * kTileSlots;
        const bool pair = a.pair_policy && (st.flags & 1);
        int adv = 1;
        if (n <= a.pair_max && (st.flags & 2)) {
            // ---- lane = (window, stump): all n windows in every wave, G stumps at a time; several stages per round when few windows are left
            const int G = 64 / n;
            int mm = 1;
            if (!in_prefix) {
                int lim = st.spec_run < last - s ? st.spec_run : last - s;        // consecutive stages from s on whose votes are exact in any order
                if (lim > 7) lim = 7;
#pragma unroll
                for (int q = 1; q < 7; q++)
                    if (mm == q && q < lim && n * (F.v[q + 1] - F.v[0]) <= a.spec_pairs) mm = q + 1;
            }
            const int g0 = F.v[0];
            int g1 = F.v[1];
#pragma unroll
            for (int q = 2; q < 8; q++) if (mm >= q) g1 = F.v[q];
            const unsigned inv = 65535u / (unsigned)n + 1u;              // floor(l / n) = l * inv >> 16 for l < 64, n <= 64
            const int cc = (int)(((unsigned)lane * inv) >> 16), j = lane - cc * n;
            if (cc < G) {
                const int w = qi[j];
                const unsigned cm = L.cmA + 2 * L.winx[w & 31], rm = L.rmA + 2 * L.winy[w >> 5];
                const double vnf = vnf_of(w);
                for (int g = g0 + wave * G + cc; g < g1; g += NW * G) {
                    const LRec f = load_lrec(sc.lrecs + g);
                    const int r = (g >= F.v[1]) + (g >= F.v[2]) + (g >= F.v[3]) + (g >= F.v[4]) + (g >= F.v[5]) + (g >= F.v[6]);      // which of the round's stages (entries behind the round's last stage are larger than any g of the round)
                    atomicAdd(&L.acc[r * kPairMax + j], lane_vote(cm, rm, vnf, f, a.pair_policy));
                }
            }
            NVCA_STAMP(a, ti, 8 + 8 * s + 1);
            __syncthreads();
            NVCA_STAMP(a, ti, 8 + 8 * s + 2);
            bool pass = false; int w = 0;
            if (tid < n) {
                pass = true;
#pragma unroll
                for (int r = 0; r < 7; r++)
                    if (r < mm) {
                        const double sum = L.acc[r * kPairMax + tid];
                        L.acc[r * kPairMax + tid] = 0.;
                        pass = pass && !(sum < (double)TH.v[r]);
                    }
                w = qi[tid];
            }
            if (s < kStatStages && tid == 0) stat_w[s] = adapt ? n << 16 : n;
            queue_push_wave0(pass, w, qo, &L.qn[cout]);
            adv = mm;
            NVCA_STAMP(a, ti, 8 + 8 * s + 3);
        } else if (st.flags & 6) {
            // ---- a window per lane: balanced runs over (window group, stump), accumulators in LDS (integers where flag bit 2 says so, f64 otherwise).
            // The queue's last, partial group of 64 does not idle its empty lanes: with rem <= 32 windows in it, lane = (window l % rem,
            // stump l / rem) as in the rounds above -- Gr = 64 / rem stumps of the stage per item instead of one (a third to a half of a
            // round's items at 60 .. 300 windows were such mostly empty groups)
            const bool ints = (st.flags & 4) != 0;
            const int C = st.count;
            const int rem = n & 63, nfull = n >> 6;
            const bool rem_pair = rem > 0 && rem <= a.pair_max;
            const int Gr = rem_pair ? 64 / rem : 1;
            const int lead = (rem_pair ? nfull : (n + 63) >> 6) * C;          // items of the groups that run a window per lane
            const int items = lead + (rem_pair ? (C + Gr - 1) / Gr : 0), K = (items + NW - 1) / NW;
            const double vscale = ints ? __hiloint2double((1023 - st.vote_exp) << 20, 0) : 1.;      // 2^-vote_exp: a vote times it is the integer the record's a0i / a1i hold
            int e = wave * K;
            const int e1 = e + K < items ? e + K : items;
            while (e < e1 && e < lead) {           // at most two window groups per wave (K <= C), wave-uniform
                const int top = e1 < lead ? e1 : lead;
                const int wg = e / C, c0 = e - wg * C, c1 = (C - c0 < top - e) ? C : c0 + (top - e);
                const int i = wg * 64 + lane;
                if (i < n) {
                    const int w = qi[i];
                    const int xw = L.winx[w & 31], yw = L.winy[w >> 5];
                    const double vnf = vnf_of(w);
                    const unsigned cm = L.cmA + 2 * xw, rm = L.rmA + 2 * yw;
                    if (ints) {
                        const int v = pair ? tile_stage_sum<true, int>(cm, rm, vnf, urecs + st.first, c0, c1, 1) : tile_stage_sum<false, int>(cm, rm, vnf, urecs + st.first, c0, c1, 1);
                        atomicAdd(&acci[i], v);
                    } else {
                        const double v = pair ? tile_stage_sum<true, double>(cm, rm, vnf, urecs + st.first, c0, c1, 1) : tile_stage_sum<false, double>(cm, rm, vnf, urecs + st.first, c0, c1, 1);
                        atomicAdd(&L.acc[i], v);
                    }
                }
                e += c1 - c0;
            }
            if (e < e1) {                          // blocks of Gr stumps on the partial group
                const unsigned inv = 65535u / (unsigned)rem + 1u;
                const int cc = (int)(((unsigned)lane * inv) >> 16), j = lane - cc * rem;
                if (cc < Gr) {
                    const int i = nfull * 64 + j;
                    const int w = qi[i];
                    const unsigned cm = L.cmA + 2 * L.winx[w & 31], rm = L.rmA + 2 * L.winy[w >> 5];
                    const double vnf = vnf_of(w);
                    double sum = 0.;
                    for (int c = (e - lead) * Gr + cc; c < (e1 - lead) * Gr && c < C; c += Gr)
                        sum += lane_vote(cm, rm, vnf, load_lrec(sc.lrecs + st.first + c), a.pair_policy);        // exact in any order (flag bit 1)
                    if (ints) atomicAdd(&acci[i], (int)(sum * vscale)); else atomicAdd(&L.acc[i], sum);
                }
            }
            NVCA_STAMP(a, ti, 8 + 8 * s + 1);
            __syncthreads();
            NVCA_STAMP(a, ti, 8 + 8 * s + 2);
            bool pass = false; int w = 0;
            if (tid < n) {
                if (ints) { const int sa = acci[tid]; acci[tid] = 0; pass = sa >= st.thr_i; }
                else { const double sa = L.acc[tid]; L.acc[tid] = 0.; pass = !(sa < (double)st.thr); }
                w = qi[tid];
            }
            if (s < kStatStages && tid == 0) stat_w[s] = adapt ? n << 16 : n;
            if (n <= 64) queue_push_wave0(pass, w, qo, &L.qn[cout]); else queue_push(pass, w, qo, &L.qn[cout]);
            NVCA_STAMP(a, ti, 8 + 8 * s + 3);
        } else {
            // votes whose sums depend on the order (neither flag bit: not seen with f32 votes): window per thread, the stage's stumps in OpenCV's order
            for (int base = 0; base < n; base += kTileThreads) {
                const int i = base + tid;
                bool pass = false; int w = 0;
                if (i < n) {
                    w = qi[i];
                    const int xw = L.winx[w & 31], yw = L.winy[w >> 5];
                    const double vnf = vnf_of(w);
                    pass = tile_stage_pass(L.cmA + 2 * xw, L.rmA + 2 * yw, vnf, urecs, st, pair);
                }
                queue_push(pass, w, qo, &L.qn[cout]);
            }
            if (tid == 0 && s < kStatStages) stat_w[s] = adapt ? n << 16 : n;
        }
        cur ^= 1; cin = cout;
        if (in_prefix) { prev = s; entered |= 1u << s; } else prev = 0;
        kpos += adv;
        if (kpos < last - 1) {
            const int sn = stage_at(kpos);
            st = load_const(a.stages + sn); F = load_const((const I8 *)(a.stage_first + sn)); TH = load_const((const F8 *)(a.stage_thr + sn));
        }
    }
    __syncthreads();
    NVCA_STAMP(a, ti, 6);
    const int nh = L.qn[cin];
    if (adapt && tid == 0) {
        if (prev) stat_w[prev] |= nh;           // the prefix was the whole walk: its last stage's survivors
        // the order for the next tile.  A stage this tile did not reach, or met with a handful of windows only, keeps what was
        // known about it; cost of a stage = its stumps per window killed (x 64: integer arithmetic)
        int key[kStatStages - 1], id[kStatStages - 1];
        bool known = true;
#pragma unroll
        for (int q = 1; q < kStatStages; q++) {
            int v = stat_w[q];
            const int old = stat_r[q];
            if (q <= m && (!((entered >> q) & 1u) || ((v >> 16) < 16 && old != 0))) { v = old; stat_w[q] = v; }
            const int ent = v >> 16, pas = v & 0xffff, cnt = L.qn[4 + 2 * kStatStages - 1 + q] /* stumps of stage q, left by the kernel's prologue */;
            if (q <= m && ent == 0) known = false;
            const int killed = ent - pas > 0 ? ent - pas : 0;
            key[q - 1] = q > m ? 0x7fffffff : (killed ? (cnt * ent * 64) / killed : 0x7ffffff0);
            id[q - 1] = q;
        }
#pragma unroll
        for (int pass_i = 0; pass_i < kStatStages - 2; pass_i++)
#pragma unroll
            for (int q = 0; q + 1 < kStatStages - 1 - pass_i; q++)
                if (key[q] / 2 > key[q + 1]) { const int tk = key[q];       /* a stage moves ahead of its predecessor only where it is at least twice as cheap per window killed: the counts are taken at whatever position a stage ran, a near-tie re-ordered on them only stirs the walk */ key[q] = key[q + 1]; key[q + 1] = tk; const int ti2 = id[q]; id[q] = id[q + 1]; id[q + 1] = ti2; }
        unsigned pack = 0;
#pragma unroll
        for (int q = 0; q < kStatStages - 1; q++) pack |= (unsigned)id[q] << (4 * q);
        stat_w[0] = known ? (int)pack : 0;
        if (known && a.stage_hint) {            // plain stores of an intentionally racy hint: any mixture of finished tiles' words is a usable start (the order word is validated before use)
#pragma unroll
            for (int q = 1; q < kStatStages; q++) a.stage_hint[q] = stat_w[q];
            a.stage_hint[0] = (int)pack;
        }
    }
    if (nh == 0) return;
    // survivors: final candidates if the cascade ends here, otherwise work for k_deep
    unsigned long long *list = last == a.nstages ? a.hits : a.deep;
    const unsigned cap = last == a.nstages ? a.hit_cap : a.deep_cap;
    if (tid == 0) L.qn[3] = (int)(unsigned)atomicAdd(list, (unsigned long long)nh);
    __syncthreads();
    const unsigned gb = (unsigned)L.qn[3];
    const unsigned short *qi = L.q0 + cur * kTileSlots;
    for (int i = tid; i < nh; i += kTileThreads) {
        const int w = qi[i], ix = t.ix0 + (w & 31);
        const unsigned key = ((unsigned)t.scale << a.key_ss) | ((unsigned)(t.iy0 + (w >> 5)) << a.key_sy) | (unsigned)ix;
        if (gb + i < cap) list[1 + gb + i] = ((unsigned long long)slot << 32) | key;
        if (VNF_LDS && last != a.nstages) a.vnf[vbase + ((size_t)(w >> 5) * sc.wpr + (ix >> 6)) * 64 + (ix & 63)] = L.vnf_s[w];
    }
    NVCA_STAMP(a, ti, 7);
}

// the per-workgroup words tile_stages expects before a band's / tile's first round: zero accumulators, the plan's stage
// statistics (the adaptive order's starting point), the stump counts of stages 1 .. 5
__device__ __forceinline__ void tile_prologue(const CascadeArgs &a, const TileLds &L)
{
    const int tid = threadIdx.x;
    L.acc[tid] = 0.;
    if (tid < 2 * kStatStages) L.qn[4 + tid] = (tid < kStatStages && a.stage_hint) ? a.stage_hint[tid] : 0;
    if (tid >= 1 && tid < kStatStages) L.qn[4 + 2 * kStatStages - 1 + tid] = tid < a.nstages ? a.stages[tid].count : 0;      // qn[16 .. 20]: behind the two sets of stat words
}

__global__ __launch_bounds__(kTileThreads, 2 * kTileThreads / 256) void k_tile(CascadeArgs a)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x;
    const int slot = blockIdx.x / a.tile_blocks_per_frame;
    const int tidx = a.tile_order[blockIdx.x - slot * a.tile_blocks_per_frame];
    if (tidx < 0) return;
    const TileRec t = a.tiles[tidx];
    const ScaleRec &sc = a.scales[t.scale];
    const TileLds L = carve_tile(lds, t);
    const TileCoords tc = tile_coords(a, t, sc);
    tile_commit(a, t, sc, slot, L, tc);
    tile_prologue(a, L);
    const unsigned long long *__restrict__ bits = a.failbits + (size_t)slot * a.ntasks + sc.task_off + (size_t)t.iy0 * sc.wpr;
    // this thread's window (window id = ry * 32 + rx = tid): the pre-pass's verdict on it, under the tile's transfers
    const int w = tid, ry = w >> 5, rx = w & 31;
    bool keep = false;
    if (ry < t.ny && rx < t.nx) {
        const int ix = t.ix0 + rx;
        const unsigned long long *rb = bits + (size_t)ry * sc.wpr;
        if (!((rb[ix >> 6] >> (ix & 63)) & 1ull)) keep = sc.adaptive ? visited(rb, ix) : true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's transfers have landed
    __syncthreads();                 // qn zeroed, maps and samples staged
    queue_push(keep, w, L.q0, &L.qn[0]);
    tile_stages<false>(a, t, sc, slot, L);
}

// ---- K5: the whole cascade of a band of window rows in one workgroup ---------------------------------------
// The workgroup walks the band's tiles left to right.  Per tile: stage the samples, evaluate the window variance and
// stage 0 for every window from LDS (only the squared-integral corners are global reads), resolve the adaptive x step
// inside the wave -- a wave holds two whole window rows of the tile, so the reject bits it needs are its own ballot, and the
// parity of the reject run that reaches the tile's left edge is carried from tile to tile in a register -- then run the
// remaining stages as k_tile does.  No stage-0 pre-pass, no per-window global intermediates.
__global__ __launch_bounds__(kTileThreads, 2 * kTileThreads / 256) void k_band(CascadeArgs a)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // longest bands first across the whole batch (a band is a serial walk; the short ones level the tail)
    int bi = blockIdx.x / a.batch, slot = blockIdx.x - bi * a.batch;
    if (a.band_map) {
        // EXPERIMENT: workgroups go round-robin over the 8 XCDs; keep a frame on one XCD and run the frames of an XCD one
        // after the other, so that the bands of a frame (all scales) share that XCD's L2
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int g = a.band_map;                                  // frames of an XCD walked together (1, 2, ...)
        const int fi = k / (a.band_blocks_per_frame * g), r = k - fi * a.band_blocks_per_frame * g;
        bi = r / g; slot = xcd + 8 * (fi * g + (r - bi * g));
    }
    const BandRec b = load_const(a.bands + a.band_order[bi]);
    const ScaleRec &sc = a.scales[b.scale];
    const unsigned *__restrict__ sql = (const unsigned *)a.sqsum + (size_t)slot * 2 * a.sum_slot + sc.plane_off;
    const uint8_t *__restrict__ sqh = (const uint8_t *)((const unsigned *)a.sqsum + (size_t)slot * 2 * a.sum_slot + a.sum_slot) + sc.plane_off;
    const bool sq_lo_only = sc.sq32 != 0;
    const int ex0 = sc.eq[0] % sc.pitch, ey0 = sc.eq[0] / sc.pitch, ex1 = sc.eq[3] % sc.pitch, ey1 = sc.eq[3] / sc.pitch;
    const StageRec st0 = load_const(a.stages);
    const bool pair0 = a.pair_policy && (st0.flags & 1);
    static_assert(kTileSlots == kTileThreads, "one window per thread and tile");
    const int w = tid, ry = w >> 5, rx = w & 31;          // this thread's window in every tile of the band
    unsigned carry = 0;                                    // wave-uniform: parity of the stage-0 reject run that ends at the previous tile's right edge, bit 0 / 1: the wave's first / second row
    // The coordinates of a tile (its record, this thread's list entries and window origin) are requested one tile ahead, right
    // behind the previous tile's sample transfers: their round trip passes under those transfers instead of standing at the
    // head of every tile (768 threads leave the registers for it)
    TileRec t = load_const(a.tiles + b.first_tile);
    TileCoords tc = tile_coords(a, t, sc);
    tile_prologue(a, carve_tile(lds, t));                  // the fixed part of the carve-up does not depend on the tile
    int oxw = 0, oyw = 0;
    if (ry < t.ny && rx < t.nx) { oxw = a.pos[sc.xpos_off + t.ix0 + rx]; oyw = a.pos[sc.ypos_off + t.iy0 + ry]; }
    for (int ti = 0; ti < b.ntiles; ti++) {
        __syncthreads();             // previous tile completely done with LDS
        NVCA_STAMP(a, ti, 0);
        const TileLds L = carve_tile(lds, t);
        // this thread's window: the four (eight) squared-integral corners -- uncoalesced global reads -- are requested before the
        // tile's samples and arrive under their transfer
        const bool active = ry < t.ny && rx < t.nx;
        int xw = oxw, yw = oyw;
        unsigned q0 = 0, q1 = 0, q2 = 0, q3 = 0, h0 = 0, h1 = 0, h2 = 0, h3 = 0;
        if (active) {
            xw -= t.x0; yw -= t.y0;
            const unsigned off = (unsigned)((t.y0 + yw) * sc.pitch + t.x0 + xw);
            const unsigned e0 = off + sc.eq[0], e1 = off + sc.eq[1], e2 = off + sc.eq[2], e3 = off + sc.eq[3];
            q0 = sql[e0]; q1 = sql[e1]; q2 = sql[e2]; q3 = sql[e3];
            if (!sq_lo_only) { h0 = sqh[e0]; h1 = sqh[e1]; h2 = sqh[e2]; h3 = sqh[e3]; }
        }
        NVCA_STAMP(a, ti, 1);
        tile_commit(a, t, sc, slot, L, tc);
        if (tid < kStatStages) L.qn[4 + ((ti + 1) & 1) * kStatStages + tid] = 0;       // the survivor counts this tile will write
        TileRec tn = t; TileCoords tcn = tc; int nxw = 0, nyw = 0;
        if (ti + 1 < b.ntiles) {                                 // the next tile's coordinates
            tn = load_const(a.tiles + b.first_tile + ti + 1);
            tcn = tile_coords(a, tn, sc);
            if (ry < tn.ny && rx < tn.nx) { nxw = a.pos[sc.xpos_off + tn.ix0 + rx]; nyw = a.pos[sc.ypos_off + tn.iy0 + ry]; }
        }
        NVCA_STAMP(a, ti, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's transfers have landed
        __syncthreads();
        NVCA_STAMP(a, ti, 3);
        // variance + stage 0 for every window of the tile; a wave covers two window rows
        bool pass0 = false;
        if (active) {
            const int c0 = L.cmap[xw + ex0], c1 = L.cmap[xw + ex1], r0 = L.rmap[yw + ey0], r1 = L.rmap[yw + ey1];
            const int ws = lds_sample(r0, c0) - lds_sample(r0, c1) - lds_sample(r1, c0) + lds_sample(r1, c1);
            const double mean = (double)ws * sc.inv_area;
            // squared-pixel sum of the variance window: exact integers below 2^53 (see window_sqsum)
            double vnf;
            if (sq_lo_only) vnf = (double)(unsigned)(q0 - q1 - q2 + q3);
            else vnf = (double)(((unsigned long long)h0 << 32) | q0) - (double)(((unsigned long long)h1 << 32) | q1) -
                       (double)(((unsigned long long)h2 << 32) | q2) + (double)(((unsigned long long)h3 << 32) | q3);
            vnf = vnf * sc.inv_area - mean * mean;
            vnf = vnf >= 0. ? sqrt(vnf) : 1.;
            L.vnf_s[w] = vnf;
            pass0 = tile_stage_pass(L.cmA + 2 * xw, L.rmA + 2 * yw, vnf, sc.trecs, st0, pair0);
        }
        // OpenCV's adaptive x step: a window is visited iff the run of stage-0 rejects immediately left of it in its
        // row has even length; a run that reaches the tile's left edge continues with the carried parity.  The wave's own
        // ballot holds the reject bits of its two rows.
        const unsigned long long fb = __ballot(active && !pass0);
        const unsigned R = lane < 32 ? (unsigned)fb : (unsigned)(fb >> 32);
        bool keep = false;
        if (active && !((R >> rx) & 1u)) {
            if (!sc.adaptive) keep = true;
            else {
                int ones = 0;
                if (rx > 0) {
                    const unsigned mk = ~(R << (32 - rx));                // window rx-1 at the MSB, rejects are 0 now
                    ones = mk ? __clz((int)mk) : 32;
                    if (ones > rx) ones = rx;
                }
                const int cbit = (int)((carry >> (lane >> 5)) & 1u);
                const int parity = ones == rx ? ((rx + cbit) & 1) : (ones & 1);
                keep = !parity;
            }
        }
        {   // parity of the reject run that ends at this tile's right edge, for both rows of the wave
            unsigned nc = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const unsigned Rh = h ? (unsigned)(fb >> 32) : (unsigned)fb;
                const unsigned mk = ~(Rh << (32 - t.nx));
                int ones = mk ? __clz((int)mk) : 32;
                if (ones > t.nx) ones = t.nx;
                const unsigned p = ones == t.nx ? ((unsigned)t.nx + ((carry >> h) & 1u)) & 1u : (unsigned)ones & 1u;
                nc |= p << h;
            }
            carry = (unsigned)__builtin_amdgcn_readfirstlane((int)nc);
        }
        queue_push(keep, w, L.q0, &L.qn[0]);
        NVCA_STAMP(a, ti, 5);
        tile_stages<true>(a, t, sc, slot, L, ti, ti & 1);
        t = tn; tc = tcn; oxw = nxw; oyw = nyw;
    }
}

// ---- K5c: one wave per surviving window, one stump per lane -------------------
__device__ __forceinline__ double wave_sum_exact(double v)
{   // only used where every partial sum is exactly representable (StageRec flag bit 1)
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__global__ __launch_bounds__(256) void k_deep(CascadeArgs a)
{
    // one workgroup per surviving window: every late stage (33..213 stumps) is one step, stump per thread.  The first late
    // stage reads the sum plane directly; a window that passes it gets the ncol x nrow samples all later stumps can touch
    // staged in LDS (DeepRec), and the remaining ~1900 stumps x 8-12 corners become LDS reads.
    __shared__ double part[2][4];
    __shared__ double votes[256];
    extern __shared__ int T[];              // the patch: sized by the plan's largest (a.deep_lds) -- the kernel lives on the windows in flight per CU
    __shared__ unsigned short cmap[kDeepMaxSpan], rmap[kDeepMaxSpan];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long cnt = a.deep[0];
    if (cnt > a.deep_cap) {                              // list overflowed: poison the hit count (host reports it)
        if (blockIdx.x == 0 && tid == 0) atomicAdd(a.hits, 1ull << 40);
        cnt = a.deep_cap;
    }
    int flip = 0;                                        // partial-sum buffers alternate across stages AND windows
    for (unsigned long long i = blockIdx.x; i < cnt; i += gridDim.x) {
        const unsigned long long e = a.deep[1 + i];
        const int slot = (int)(e >> 32);
        const unsigned key = (unsigned)e;
        const int s = key >> a.key_ss, iy = (key >> a.key_sy) & ((1u << (a.key_ss - a.key_sy)) - 1u), ix = key & ((1u << a.key_sy) - 1u);
        const ScaleRec &sc = a.scales[s];
        const int *__restrict__ sum = a.sum + (size_t)slot * a.sum_slot + sc.plane_off;
        const unsigned off = (unsigned)(a.pos[sc.ypos_off + iy] * sc.pitch + a.pos[sc.xpos_off + ix]);
        const double vnf = a.vnf[((size_t)slot * a.ntasks + sc.task_off + (size_t)iy * sc.wpr + (ix >> 6)) * 64 + (ix & 63)];
        DeepRec d; d.ncol = 0;
        if (a.deeprecs) d = a.deeprecs[s];
        CTStumpRec *trecs = (CTStumpRec *)sc.trecs;
        bool alive = true, patch = false;
        for (int st_i = a.deep_stage; st_i < a.nstages; st_i++) {
            const StageRec st = a.stages[st_i];
            const bool pair = a.pair_policy && (st.flags & 1);
            double stage_sum = 0.0;
            if (st_i == a.deep_stage + 1 && d.ncol > 0) {           // passed the first late stage: stage the patch
                const unsigned short *__restrict__ cl = a.tcoords + d.col_off, *__restrict__ rl = a.tcoords + d.row_off;
                const int pitchP = d.ncol | 1;
                __syncthreads();                                  // previous window's patch reads are over
                for (int c = tid; c < d.ncol; c += 256) cmap[cl[c]] = (unsigned short)(c * 4);
                for (int r = tid; r < d.nrow; r += 256) rmap[rl[r]] = (unsigned short)(r * pitchP);
                for (int q = tid; q < d.ncol * d.nrow; q += 256) {
                    const int r = q / d.ncol, c = q - r * d.ncol;
                    T[r * pitchP + c] = sum[off + (unsigned)rl[r] * (unsigned)sc.pitch + cl[c]];
                }
                __syncthreads();
                patch = true;
            }
            auto vote = [&](int j) {
                if (patch) return pair ? tile_vote<true, false>(T, cmap, rmap, 0, 0, vnf, trecs[st.first + j])
                                       : tile_vote<false, false>(T, cmap, rmap, 0, 0, vnf, trecs[st.first + j]);
                return pair ? stump_vote<true>(sum, off, sc.pitch, vnf, trecs[st.first + j]) : stump_vote<false>(sum, off, sc.pitch, vnf, trecs[st.first + j]);
            };
            if (st.flags & 2) {                         // any summation order is exact
                double p = 0.0;
                for (int j = tid; j < st.count; j += 256) p += vote(j);
                p = wave_sum_exact(p);
                if (lane == 0) part[flip][wave] = p;
                __syncthreads();
                stage_sum = (part[flip][0] + part[flip][1]) + (part[flip][2] + part[flip][3]);
                flip ^= 1;
            } else {                                    // keep OpenCV's left-to-right order
                for (int c = 0; c < st.count; c += 256) {
                    const int j = c + tid;
                    const double v = j < st.count ? vote(j) : 0.0;
                    __syncthreads();
                    votes[tid] = v;
                    __syncthreads();
                    const int m = st.count - c < 256 ? st.count - c : 256;
                    for (int l = 0; l < m; l++) stage_sum += votes[l];      // every thread walks the same order
                }
            }
            if (stage_sum < (double)st.thr) { alive = false; break; }
        }
        if (alive && tid == 0) {
            const unsigned long long h = atomicAdd(a.hits, 1ull);
            if (h < a.hit_cap) a.hits[1 + h] = e;
        }
    }
}

// ---- K6: groupRectangles on the device, one workgroup per frame ----------------------------------
// cv::groupRectangles(rects, groupThreshold, 0.2) as called at the end of cvHaarDetectObjectsForROC
// (cascadedetect.cpp / operations.hpp partition()).  The raw candidates of a frame are pulled out of the
// batch-wide list, sorted by key (= OpenCV's serial scan order: scale, y, x), clustered with a min-index
// union-find in LDS (class numbering by first member, as cv::partition assigns it), averaged with the same
// float rounding, filtered, and written as final boxes: the host only receives boxes.
static constexpr int kGroupMax = 2048;       // raw candidates per frame handled on the device

__device__ __forceinline__ bool similar_rects(const int4 &a, const int4 &b, double eps)
{
    const double delta = eps * ((a.z < b.z ? a.z : b.z) + (a.w < b.w ? a.w : b.w)) * 0.5;
    return abs(a.x - b.x) <= delta && abs(a.y - b.y) <= delta && abs(a.x + a.z - b.x - b.z) <= delta &&
           abs(a.y + a.w - b.y - b.w) <= delta;
}

__global__ __launch_bounds__(256) void k_group(CascadeArgs a, const int *__restrict__ group_thr, int *__restrict__ out, int out_cap)
{
    __shared__ unsigned keys[kGroupMax];
    __shared__ int parent[kGroupMax];
    __shared__ int cls_of[kGroupMax];          // class id of a root; then reused
    __shared__ int4 rects[kGroupMax];
    __shared__ int csum[4][256], ccnt[256];    // per-class sums (classes beyond 256 -> fallback)
    __shared__ int n_s, ncls_s, fallback_s;
    const int tid = threadIdx.x, slot = blockIdx.x;
    int *o = out + (size_t)slot * (2 + 4 * out_cap);       // [0] = count (-1: host must group), [1] = raw count
    if (tid == 0) { n_s = 0; ncls_s = 0; fallback_s = 0; }
    __syncthreads();
    unsigned long long total = a.hits[0];
    if (slot == 0 && tid == 0) {               // the raw candidate count travels with the box table (two words after the last record)
        int *tail = out + (size_t)gridDim.x * (2 + 4 * out_cap);
        tail[0] = (int)(unsigned)(total & 0xffffffffull); tail[1] = (int)(unsigned)(total >> 32);
    }
    if (total > a.hit_cap) total = a.hit_cap;
    for (unsigned long long i = tid; i < total; i += 256) {
        const unsigned long long e = a.hits[1 + i];
        if ((int)(e >> 32) == slot) {
            const int k = atomicAdd(&n_s, 1);
            if (k < kGroupMax) keys[k] = (unsigned)e;
        }
    }
    __syncthreads();
    const int n = n_s;
    const int thr = group_thr[slot];
    if (tid == 0) o[1] = n;
    if (n > kGroupMax || thr <= 0) { if (tid == 0) o[0] = -1; return; }       // host path (rare / ungrouped)
    if (n == 0) { if (tid == 0) o[0] = 0; return; }
    // ---- keys in ascending order.  Up to 256 of them (the usual case: a few dozen candidates per face): every thread counts
    // the keys below its own (keys are distinct windows; equal keys would be told apart by position) and stores it at that
    // rank -- two barriers instead of the ~30 of a bitonic network, which is what longer lists go through
    if (n <= 256) {
        const unsigned mine = tid < n ? keys[tid] : 0u;
        int rank = 0;
        if (tid < n)
            for (int j = 0; j < n; j++) { const unsigned o = keys[j]; rank += (o < mine || (o == mine && j < tid)) ? 1 : 0; }
        __syncthreads();
        if (tid < n) keys[rank] = mine;
        __syncthreads();
    } else {
        int np = 1;
        while (np < n) np <<= 1;
        for (int i = n + tid; i < np; i += 256) keys[i] = 0xffffffffu;
        __syncthreads();
        for (int k = 2; k <= np; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < np; i += 256) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned x = keys[i], y = keys[l];
                        const bool up = (i & k) == 0;
                        if ((x > y) == up) { keys[i] = y; keys[l] = x; }
                    }
                }
                __syncthreads();
            }
    }
    // ---- rectangles + singleton sets
    for (int i = tid; i < n; i += 256) {
        const unsigned key = keys[i];
        const int s = key >> a.key_ss, iy = (key >> a.key_sy) & ((1u << (a.key_ss - a.key_sy)) - 1u), ix = key & ((1u << a.key_sy) - 1u);
        const ScaleRec &sc = a.scales[s];
        rects[i] = make_int4(a.pos[sc.xpos_off + ix], a.pos[sc.ypos_off + iy], sc.winw, sc.winh);
        parent[i] = i;
    }
    __syncthreads();
    // ---- union of similar pairs (root = smallest member index)
    const long long npairs = (long long)n * (n - 1) / 2;
    for (long long p = tid; p < npairs; p += 256) {
        // unrank p -> (i < j)
        int j = (int)((1.0 + sqrt(1.0 + 8.0 * (double)p)) * 0.5);
        while ((long long)j * (j - 1) / 2 > p) j--;
        while ((long long)(j + 1) * j / 2 <= p) j++;
        const int i = (int)(p - (long long)j * (j - 1) / 2);
        if (!similar_rects(rects[i], rects[j], 0.2)) continue;
        int x = i, y = j;
        for (;;) {
            while (parent[x] != x) x = parent[x];
            while (parent[y] != y) y = parent[y];
            if (x == y) break;
            if (x > y) { const int t = x; x = y; y = t; }
            const int old = atomicMin(&parent[y], x);
            if (old == y) break;
            y = old;
        }
    }
    __syncthreads();
    // ---- class ids in order of first member; per-class sums
    for (int i = tid; i < n; i += 256) {
        int r = i;
        while (parent[r] != r) r = parent[r];
        cls_of[i] = r;                         // root index for now
    }
    __syncthreads();
    if (tid == 0) {                             // number the roots in ascending order (n <= 2048: a short serial pass)
        int c = 0;
        for (int i = 0; i < n; i++) if (cls_of[i] == i) { parent[i] = c < 256 ? c : 255; c++; }
        ncls_s = c;
        if (c > 256) fallback_s = 1;
    }
    for (int i = tid; i < 256; i += 256) { csum[0][i] = csum[1][i] = csum[2][i] = csum[3][i] = 0; ccnt[i] = 0; }
    __syncthreads();
    if (fallback_s) { if (tid == 0) o[0] = -1; return; }
    const int ncls = ncls_s;
    for (int i = tid; i < n; i += 256) {
        const int c = parent[cls_of[i]];       // class id of my root
        atomicAdd(&csum[0][c], rects[i].x); atomicAdd(&csum[1][c], rects[i].y);
        atomicAdd(&csum[2][c], rects[i].z); atomicAdd(&csum[3][c], rects[i].w);
        atomicAdd(&ccnt[c], 1);
    }
    __syncthreads();
    // ---- class averages: float s = 1.f / n; saturate_cast<int>(sum * s) (round half to even)
    int4 avg = make_int4(0, 0, 0, 0); int cnt = 0;
    if (tid < ncls) {
        cnt = ccnt[tid];
        const float sc = 1.f / (float)cnt;
        avg = make_int4(__float2int_rn((float)csum[0][tid] * sc), __float2int_rn((float)csum[1][tid] * sc),
                        __float2int_rn((float)csum[2][tid] * sc), __float2int_rn((float)csum[3][tid] * sc));
    }
    __syncthreads();
    if (tid < ncls) { rects[tid] = avg; cls_of[tid] = cnt; }      // reuse LDS: class rects and weights
    __syncthreads();
    // ---- keep class i unless too weak or inside a stronger one
    bool keepc = false;
    if (tid < ncls && cnt > thr) {
        keepc = true;
        for (int j = 0; j < ncls; j++) {
            const int n2 = cls_of[j];
            if (j == tid || n2 <= thr) continue;
            const int4 r2 = rects[j];
            const int dx = __double2int_rn(r2.z * 0.2), dy = __double2int_rn(r2.w * 0.2);
            if (avg.x >= r2.x - dx && avg.y >= r2.y - dy && avg.x + avg.z <= r2.x + r2.z + dx && avg.y + avg.w <= r2.y + r2.w + dy &&
                (n2 > (3 > cnt ? 3 : cnt) || cnt < 3)) { keepc = false; break; }
        }
    }
    // ncls <= 256: one ballot-ordered compaction over the 4 waves, in class order
    __shared__ int wcount[4];
    const unsigned long long km = __ballot(keepc);
    if ((tid & 63) == 0) wcount[tid >> 6] = __popcll(km);
    __syncthreads();
    int base = 0;
    for (int wv = 0; wv < (tid >> 6); wv++) base += wcount[wv];
    if (keepc) {
        const int pos = base + __popcll(km & ((1ull << (tid & 63)) - 1ull));
        if (pos < out_cap) { int *q = o + 2 + 4 * pos; q[0] = avg.x; q[1] = avg.y; q[2] = avg.z; q[3] = avg.w; }
    }
    if (tid == 0) o[0] = wcount[0] + wcount[1] + wcount[2] + wcount[3];
}

void launch_group(hipStream_t st, const CascadeArgs &a, const int *group_thr, int *out, int out_cap, int batch)
{
    NVCA_LAUNCH(k_group, dim3(batch), dim3(256), 0, st, a, group_thr, out, out_cap);
}

// Dynamic LDS above 64 KiB has to be granted per function and device (hipFuncSetAttribute SETS the limit, it does not raise
// it): the limit only ever grows, under one process-wide lock, so a context that needs less can never lower what another
// context on the same device was granted.  `granted` is the calling context's own record (skips the lock once it is covered).
static int grant_lds(const void *fn, int which, int bytes, int *granted)
{
    if (bytes <= *granted) return 0;
    static std::mutex mu;
    static int dev_max[2][64] = {{0}};
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0; (void)hipGetDevice(&dev); dev &= 63;
    if (bytes > dev_max[which][dev]) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
        dev_max[which][dev] = bytes;
    }
    *granted = dev_max[which][dev];
    return 0;
}

int launch_cascade_sc(hipStream_t st, const CascadeArgs &a, int batch, int which, int *lds_grant)
{
    if (batch <= 0 || a.ntasks <= 0) return 0;
    if (which == 0) {
        const int s0_blocks = (((a.ntasks + 3) / 4 + 7) / 8) * 8;
        NVCA_LAUNCH(k_stage0, dim3((unsigned)s0_blocks * (unsigned)batch), dim3(256), 0, st, a);
    } else if (which == 3) {
        if (a.tile_blocks_per_frame > 0) {
            if (int e = grant_lds(reinterpret_cast<const void *>(k_tile), 0, a.tile_lds, &lds_grant[0])) return e;
            NVCA_LAUNCH(k_tile, dim3((unsigned)a.tile_blocks_per_frame * (unsigned)batch), dim3(kTileThreads), (size_t)a.tile_lds, st, a);
        }
    } else if (which == 5) {
        if (a.band_blocks_per_frame > 0) {
            if (int e = grant_lds(reinterpret_cast<const void *>(k_band), 1, a.tile_lds, &lds_grant[1])) return e;
            NVCA_LAUNCH(k_band, dim3((unsigned)a.band_blocks_per_frame * (unsigned)batch), dim3(kTileThreads), (size_t)a.tile_lds, st, a);
        }
    } else if (which == 1) {
        if (a.blocks_per_frame > 0)
            NVCA_LAUNCH(k_strip, dim3((unsigned)a.blocks_per_frame * (unsigned)batch), dim3(256), 0, st, a);
    } else if (a.deep_stage < a.nstages) {
        // grid-stride over the list, one window per workgroup at a time; small jobs (the part detectors' ROI searches) do not
        // need eight thousand workgroups to find a handful of windows
        long long wg = (long long)a.ntasks * batch / 2;
        if (wg < 128) wg = 128;
        if (wg > 8192) wg = 8192;
        NVCA_LAUNCH(k_deep, dim3((unsigned)wg), dim3(256), (size_t)(a.deeprecs ? a.deep_lds : 0), st, a);
    }
    return 0;
}

} // namespace nvca
