// kernels_roi.hip -- detectMultiScale on a SMALL image in one workgroup: the part detectors' searches inside a face region
// (EYE/kmseyedetect.cpp:1002-1030 CV_HAAR_SCALE_IMAGE; NOSE/kmsnosedetect.cpp:870-873, MOUTH/kmsmouthdetect.cpp:870-873,
// EAR/kmseardetect.cpp:712-715 CV_HAAR_FIND_BIGGEST_OBJECT) and any other call on an image whose integral pair fits LDS.
//
// A face region is a few thousand pixels and a few thousand windows.  The large-image path (plan per geometry, integral
// kernel, stage-0 pre-pass, tile kernel, late-stage kernel: four launches and a table build per region) is built for
// megapixel frames; here ONE launch serves every region of every stream of a round, one workgroup per (region, ladder step / level):
//   * the region's integral and squared integral are built in LDS (row scans inside waves, then a column pass in place);
//     a CV_HAAR_SCALE_IMAGE job first resizes the region to each pyramid level (cv::resize's fixed-point bilinear, tables from
//     the host) and integrates the level;
//   * every window of every ladder step / level is evaluated from LDS with plain corner arithmetic -- row * pitch + column,
//     no lattice maps, no table of positions: grid coordinates are cvRound(i * ystep) computed here;
//   * OpenCV's adaptive x step (ix += result != 0 ? 1 : 2) is resolved per row from the stage-0 reject bits of the 64-window
//     chunks a wave walks left to right (the parity of a reject run is carried from chunk to chunk);
//   * candidates go to one list for the whole launch as (job << 32) | step << 26 | iy << 13 | ix; the host orders them
//     (= OpenCV's serial order), groups them and, for FIND_BIGGEST, replays the serial search on them (api.cpp).
// Same arithmetic as the large-image kernels: i32 rectangle sums, f32 products, f64 variance / thresholds / stage sums in
// OpenCV's order (one lane walks a window's stumps in order; a stage whose partial sums are exact in any order may be split over
// lanes and summed in LDS), contraction off.  Stump cascades with upright features only.
#include "nvca_internal.h"

namespace nvca {

typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) uint8_t lds_u8;

__device__ __forceinline__ unsigned roi_wave_scan(unsigned v, int lane)
{   // wave64 inclusive add-scan on the VALU (DPP row shifts, then the three row totals through readlane)
    unsigned x = v;
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xe, true);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xc, true);
    const unsigned t0 = (unsigned)__builtin_amdgcn_readlane((int)x, 15), t1 = (unsigned)__builtin_amdgcn_readlane((int)x, 31),
                   t2 = (unsigned)__builtin_amdgcn_readlane((int)x, 47);
    const int row = lane >> 4;
    return x + (row >= 1 ? t0 : 0u) + (row >= 2 ? t1 : 0u) + (row >= 3 ? t2 : 0u);
}

// integral (i32) and squared integral (u32: exact modulo 2^32, which is all a window sum below 2^32 needs) of a w x h image,
// pitch P = w + 1, in place in LDS: row prefixes first, then every column accumulates downwards
template <class Pix>
__device__ __forceinline__ void roi_integral(Pix pix, int w, int h, lds_i32 *s, lds_u32 *q, int P)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nwaves = nthreads >> 6;
    // row 0, and one spare row below the last: a scaled rectangle may end a pixel past its window (cvRound(x f) + cvRound(w f)
    // against cvRound((x + w) f)); OpenCV reads whatever lies there, here it is a defined zero
    for (int X = tid; X <= w; X += nthreads) { s[X] = 0; q[X] = 0; s[(h + 1) * P + X] = 0; q[(h + 1) * P + X] = 0; }
    for (int y = wave; y < h; y += nwaves) {
        unsigned cs = 0, cq = 0;
        for (int x0 = 0; x0 < w; x0 += 64) {
            const int x = x0 + lane;
            const unsigned v = x < w ? (unsigned)pix(x, y) : 0u;
            const unsigned is = roi_wave_scan(v, lane) + cs, iq = roi_wave_scan(v * v, lane) + cq;
            if (x < w) { s[(y + 1) * P + x + 1] = (int)is; q[(y + 1) * P + x + 1] = iq; }
            cs = (unsigned)__builtin_amdgcn_readlane((int)is, 63); cq = (unsigned)__builtin_amdgcn_readlane((int)iq, 63);
        }
        if (lane == 0) { s[(y + 1) * P] = 0; q[(y + 1) * P] = 0; }
    }
    __syncthreads();
    for (int x = tid; x < w; x += nthreads) {
        unsigned as = 0, aq = 0;
        for (int y = 1; y <= h; y++) { as += (unsigned)s[y * P + x + 1]; aq += q[y * P + x + 1]; s[y * P + x + 1] = (int)as; q[y * P + x + 1] = aq; }
    }
    __syncthreads();
}

// cvRunHaarClassifierCascadeSum for one window at plane offset `off` (stump branch), in two halves: the variance normaliser,
// and stages [i0, i1) -- returns the index of the first stage that rejects, or i1 if none does.  The stump records and stage
// records are wave-uniform addresses (scalar loads).
__device__ __forceinline__ double roi_window_vnf(const lds_i32 *s, const lds_u32 *q, int off, int P, const RoiStep &st)
{
    const int e0 = off + st.ey * P + st.ex, e1 = e0 + st.ew, e2 = e0 + st.eh * P, e3 = e2 + st.ew;
    const int ws = s[e0] - s[e1] - s[e2] + s[e3];
    const double mean = (double)ws * st.inv_area;
    double vnf = (double)(unsigned)(q[e0] - q[e1] - q[e2] + q[e3]);
    vnf = vnf * st.inv_area - mean * mean;
    return vnf >= 0. ? sqrt(vnf) : 1.;
}
__device__ __forceinline__ int roi_run_stages(const lds_i32 *s, int off, int P, double vnf, const TStumpRec *recs, const StageRec *stages, int i0, int i1,
                                              int pair_policy)
{
    for (int i = i0; i < i1; i++) {
        const StageRec sr = stages[i];
        const bool pair = pair_policy && (sr.flags & 1);
        double stage_sum = 0.0;
        for (int j = 0; j < sr.count; j++) {
            const TStumpRec &f = recs[sr.first + j];
            auto rs = [&](int k) {
                const int r0 = off + f.y0[k] * P, r1 = off + f.y1[k] * P;
                return s[r0 + f.x0[k]] - s[r0 + f.x1[k]] - s[r1 + f.x0[k]] + s[r1 + f.x1[k]];
            };
            const int s0 = rs(0), s1 = rs(1);
            const double t = f.thr * vnf;
            double v;
            if (pair) v = (double)((float)s0 * f.w[0] + (float)s1 * f.w[1]);
            else {
                v = (double)((float)s0 * f.w[0]);
                v += (double)((float)s1 * f.w[1]);
                if ((f.nrect & 255) == 3) v += (double)((float)rs(2) * f.w[2]);
            }
            stage_sum += v >= t ? f.a1 : f.a0;
        }
        if (stage_sum < (double)sr.thr) return i;
    }
    return i1;
}

__device__ __forceinline__ int roi_cvround(double v) { return (int)rint(v); }        // cvRound: round half to even (default rounding mode)

// one stump's vote on one window (the same expression roi_run_stages walks in order)
__device__ __forceinline__ double roi_vote(const lds_i32 *s, int off, int P, double vnf, const TStumpRec &f, bool pair)
{
    auto rs = [&](int k) {
        const int r0 = off + f.y0[k] * P, r1 = off + f.y1[k] * P;
        return s[r0 + f.x0[k]] - s[r0 + f.x1[k]] - s[r1 + f.x0[k]] + s[r1 + f.x1[k]];
    };
    const int s0 = rs(0), s1 = rs(1);
    const double t = f.thr * vnf;
    double v;
    if (pair) v = (double)((float)s0 * f.w[0] + (float)s1 * f.w[1]);
    else {
        v = (double)((float)s0 * f.w[0]);
        v += (double)((float)s1 * f.w[1]);
        if ((f.nrect & 255) == 3) v += (double)((float)rs(2) * f.w[2]);
    }
    return v >= t ? f.a1 : f.a0;
}

// The windows of ONE ladder step / pyramid level, by the whole workgroup (16 waves), in three phases:
//   A  dense: variance normaliser + stage 0 for every window (a wave per grid row, 64 windows at a time, the adaptive x step's
//      reject-run parity carried from chunk to chunk); the windows the serial walk visits and stage 0 passes are queued
//   B  every further stage on the compacted queue, re-queued after every stage: a window per lane while many are left (or where a
//      stage's votes may not be re-ordered), lane = (window, group of the stage's stumps) with f64 accumulators in LDS once at most
//      kRoiPairWin are (votes exact in any order -- StageRec flag bit 1): all 16 waves stay busy on a handful of windows
//   C  whoever is left after the last stage goes to the launch's candidate list
// The queues and the per-window normalisers hold kRoiMaxWin windows: a larger grid goes through in bands of whole rows.
static constexpr int kRoiThreads = 1024, kRoiWaves = kRoiThreads / 64;
#ifdef NVCA_STAMPS
// diagnostic build: thread 0 of every workgroup adds the s_memtime ticks of each phase to a device-global table
// ([0..7] scale-cascade steps, [8..15] scale-image levels: image + integral, A, B, C, workgroups, windows in C)
__device__ unsigned long long g_roi_phase[16];
#define ROI_STAMP(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_roi_phase[sbase + (k)], now_ - tprev); tprev = now_; } } while (0)
#define ROI_STAMP_ARGS , unsigned long long &tprev, int sbase
#define ROI_STAMP_PASS , tprev, sbase
#else
#define ROI_STAMP(k) do { } while (0)
#define ROI_STAMP_ARGS
#define ROI_STAMP_PASS
#endif
#ifdef NVCA_STAMPS
static constexpr int kRoiDeep = 4;            // (diagnostic build: where the phase stamps split the stage walk)
#endif
static constexpr int kRoiPairWin = 384;        // windows up to which a stage runs lane = (window, stump group)
struct RoiLds { lds_i32 *s; lds_u32 *q; double *vnf; unsigned short *qa, *qb; int *cnt; lds_u8 *lev; };
template <class Pos>
__device__ __forceinline__ void roi_band_windows(const RoiJobDev &job, const RoiStep &st, int li, int nx, int gy0, int gy1, int P, const RoiLds &L, Pos pos,
                                                 unsigned long long *__restrict__ hits, unsigned hit_cap, unsigned long long *__restrict__ rejbuf ROI_STAMP_ARGS)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nst = job.nstages;
#ifdef NVCA_STAMPS
    const int deep = nst < kRoiDeep ? nst : kRoiDeep;
#endif
    const unsigned long long slot = (unsigned long long)job.slot << 32;
    if (tid < 4) L.cnt[tid] = 0;
    __syncthreads();
    // ---- A
    for (int gy = gy0 + wave; gy < gy1; gy += kRoiWaves) {
        int carry = 0;
        for (int c0 = 0; c0 < nx; c0 += 64) {
            const int gx = c0 + lane;
            const bool in = gx < nx;
            double vnf = 1.; bool pass0 = false;
            if (in) {
                int x, y; pos(gx, gy, x, y);
                const int off = y * P + x;
                vnf = roi_window_vnf(L.s, L.q, off, P, st);
                L.vnf[(gy - gy0) * nx + gx] = vnf;
                pass0 = roi_run_stages(L.s, off, P, vnf, st.trecs, job.stages, 0, 1, job.pair_policy) == 1;
            }
            const unsigned long long rej = __ballot(in && !pass0);            // rejected by stage 0: the serial walk steps by 2 behind it
            if (st.adaptive == 2 && lane == 0) rejbuf[(size_t)st.rej_off + (size_t)(st.startY + gy) * st.rej_wpr + ((st.startX + c0) >> 6)] = rej;
            bool visited = in;
            if (in && st.adaptive == 1) {
                // window gx is visited iff the run of stage-0 rejects immediately to its left has even length
                const unsigned long long below = lane ? (rej << (64 - lane)) : 0ull;       // bit 63 = lane - 1
                int ones = lane ? __clzll((long long)~below) : 0;
                if (ones > lane) ones = lane;
                visited = !(ones == lane ? ((lane + carry) & 1) : (ones & 1));
            }
            const bool keep = visited && pass0;
            const unsigned long long km = __ballot(keep);
            if (km) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&L.cnt[0], (int)__popcll(km));
                base = __shfl(base, 0);
                if (keep) L.qa[base + __popcll(km & ((1ull << lane) - 1ull))] = (unsigned short)((gy - gy0) * nx + gx);
            }
            const int nin = nx - c0 < 64 ? nx - c0 : 64;                      // the reject run that ends at this chunk's right edge
            const unsigned long long top = nin < 64 ? (rej << (64 - nin)) : rej;
            int tail = ~top ? __clzll((long long)~top) : 64;
            if (tail > nin) tail = nin;
            carry = tail == nin ? ((nin + carry) & 1) : (tail & 1);
        }
    }
#ifdef NVCA_STAMPS
    __syncthreads();
    ROI_STAMP(1);
#endif
    // ---- B: stage after stage on the compacted queue.  The counters rotate over three words (read cin, append to cout, clear the
    // third): one barrier per stage.  Two ways through a stage:
    //   * many windows (or votes that may not be re-ordered): a window per lane, the stage's stumps in OpenCV's order;
    //   * at most kRoiPairWin windows and votes that are exact in any order (StageRec flag bit 1): lane = (window i, group g of the
    //     stage's stumps), G = 1024 / n groups -- every wave of the workgroup is busy whatever n is, a lane walks count / G stumps
    //     instead of the whole stage, and the partial sums meet in an f64 accumulator in LDS (ds_add_f64; exact, so the total
    //     is the one OpenCV forms).  Round 3 ran the late stages a wave per window, a stump per lane: with a calibrated cascade
    //     60-90 windows reach them and leave one by one, each stage of each window a chain of three global round trips (stage
    //     record, stump records, wave reduction) on one wave -- 67-132 k cycles of a workgroup's 145-230 k.
    // The accumulators live in the upper part of the first queue (entries 512 ..: a queue of <= 384 windows never reaches them).
    int cin = 0, cur = 0;
    bool acc_ready = false;
    double *acc = (double *)(L.qa + 512);
    for (int sidx = 1; sidx < nst; sidx++) {
        __syncthreads();
#ifdef NVCA_STAMPS
        if (sidx == deep) ROI_STAMP(2);
#endif
        const int n = L.cnt[cin];
        if (n == 0) break;
        const int cout = cin == 2 ? 0 : cin + 1;
        if (tid == 0) L.cnt[cout == 2 ? 0 : cout + 1] = 0;
        // the two queues lie back to back (qb == qa + kRoiMaxWin): chosen by arithmetic -- a select between the struct's pointer
        // members made the compiler keep the struct in scratch memory and index it (56 bytes of private segment per thread, the
        // only kernel of the library with any)
        const unsigned short *qi = L.qa + cur * kRoiMaxWin;
        unsigned short *qo = L.qa + (cur ^ 1) * kRoiMaxWin;
        const StageRec sr = job.stages[sidx];
        if (n <= kRoiPairWin && (sr.flags & 2)) {
            if (!acc_ready) {                          // (whatever earlier, longer queues left there)
                if (tid < kRoiPairWin) acc[tid] = 0.;
                acc_ready = true;
                __syncthreads();
            }
            const bool pair = job.pair_policy && (sr.flags & 1);
            int G = kRoiThreads / n;
            if (G > sr.count) G = sr.count;
            const int g = tid / n, i = tid - g * n;
            if (g < G) {
                const int wi = qi[i];
                const int ry = wi / nx, gx = wi - ry * nx;
                int x, y; pos(gx, gy0 + ry, x, y);
                const int off = y * P + x;
                const double vnf = L.vnf[wi];
                double p = 0.0;
                for (int j = g; j < sr.count; j += G) p += roi_vote(L.s, off, P, vnf, st.trecs[sr.first + j], pair);
                atomicAdd(&acc[i], p);
            }
            __syncthreads();
            bool pass = false; int wi = 0;
            if (tid < n) { const double sum = acc[tid]; acc[tid] = 0.; pass = !(sum < (double)sr.thr); wi = qi[tid]; }
            if (tid < ((n + 63) & ~63)) {              // the waves that hold a window of the queue (wave-uniform)
                const unsigned long long pm = __ballot(pass);
                if (pm) {
                    int b2 = 0;
                    if (lane == 0) b2 = atomicAdd(&L.cnt[cout], (int)__popcll(pm));
                    b2 = __shfl(b2, 0);
                    if (pass) qo[b2 + __popcll(pm & ((1ull << lane) - 1ull))] = (unsigned short)wi;
                }
            }
        } else {
            for (int base = 0; base < n; base += kRoiThreads) {
                const int i = base + tid;
                bool pass = false; int wi = 0;
                if (i < n) {
                    wi = qi[i];
                    const int ry = wi / nx, gx = wi - ry * nx;
                    int x, y; pos(gx, gy0 + ry, x, y);
                    pass = roi_run_stages(L.s, y * P + x, P, L.vnf[wi], st.trecs, job.stages, sidx, sidx + 1, job.pair_policy) == sidx + 1;
                }
                const unsigned long long pm = __ballot(pass);
                if (pm) {
                    int b2 = 0;
                    if (lane == 0) b2 = atomicAdd(&L.cnt[cout], (int)__popcll(pm));
                    b2 = __shfl(b2, 0);
                    if (pass) qo[b2 + __popcll(pm & ((1ull << lane) - 1ull))] = (unsigned short)wi;
                }
            }
        }
        cur ^= 1; cin = cout;
    }
    __syncthreads();
#ifdef NVCA_STAMPS
    if (nst <= deep) ROI_STAMP(2);
#endif
    // ---- C: whoever is left passed every stage
    const int nleft = nst > 1 ? L.cnt[cin] : L.cnt[0];
#ifdef NVCA_STAMPS
    if (tid == 0) atomicAdd(&g_roi_phase[sbase + 5], (unsigned long long)nleft);
#endif
    if (nleft > 0) {
        if (tid == 0) L.cnt[3] = (int)(unsigned)atomicAdd(hits, (unsigned long long)nleft);
        __syncthreads();
        const unsigned long long hb = (unsigned long long)(unsigned)L.cnt[3];
        const unsigned short *qi = L.qa + cur * kRoiMaxWin;
        for (int k = tid; k < nleft; k += kRoiThreads) {
            const int wi = qi[k], gy = gy0 + wi / nx, gx = wi - (wi / nx) * nx;
            const unsigned key = ((unsigned)li << 26) | ((unsigned)(st.key_y0 + gy * st.key_dy) << 13) | (unsigned)(st.key_x0 + gx * st.key_dx);
            if (hb + k < hit_cap) hits[1 + hb + k] = slot | key;
        }
    }
    __syncthreads();                                  // the queues are reused by the next band
    ROI_STAMP(3);
}
// a step's grid in bands of whole rows that fit the queues (the adaptive x step works row by row: bands are independent)
template <class Pos>
__device__ __forceinline__ void roi_step_windows(const RoiJobDev &job, const RoiStep &st, int li, int nx, int ny, int P, const RoiLds &L, Pos pos,
                                                 unsigned long long *__restrict__ hits, unsigned hit_cap, unsigned long long *__restrict__ rejbuf ROI_STAMP_ARGS)
{
    if (nx <= 0 || ny <= 0) return;
    const int rows_per = nx >= kRoiMaxWin ? 1 : kRoiMaxWin / nx;
    for (int gy0 = 0; gy0 < ny; gy0 += rows_per) roi_band_windows(job, st, li, nx, gy0, gy0 + rows_per < ny ? gy0 + rows_per : ny, P, L, pos, hits, hit_cap, rejbuf ROI_STAMP_PASS);
}

// One workgroup per (job, step): the steps of a job are independent of one another once the image is there, so each takes the
// image's integral pair (a few microseconds for a face region) for itself -- hundreds of workgroups per launch instead of one
// long chain per job.  Dynamic LDS: sum plane | squared plane | per-window normalisers | two queues | counters | (scale-image)
// the level's gray image.
__global__ __launch_bounds__(kRoiThreads) void k_roi(const RoiJobDev *__restrict__ jobs, const RoiStep *__restrict__ steps, const unsigned char *__restrict__ tabs,
                                                     unsigned long long *__restrict__ hits, unsigned hit_cap, int plane_words, unsigned long long *__restrict__ rejbuf)
{
    extern __shared__ __align__(16) unsigned char roi_lds[];
    const RoiStep st = steps[blockIdx.x];
    const RoiJobDev job = jobs[st.job];
    const int li = st.key_step;               // the step's number inside its job (several records -- bands of rows -- may share it)
    RoiLds L;
    L.s = (lds_i32 *)roi_lds;
    L.q = (lds_u32 *)(L.s + plane_words);
    L.vnf = (double *)(L.q + plane_words);
    L.qa = (unsigned short *)(L.vnf + kRoiMaxWin);
    L.qb = L.qa + kRoiMaxWin;
    L.cnt = (int *)(L.qb + kRoiMaxWin);
    L.lev = (lds_u8 *)(L.cnt + 4);
    const int tid = threadIdx.x;
    const uint8_t *__restrict__ img = job.img;
#ifdef NVCA_STAMPS
    unsigned long long tprev = __builtin_amdgcn_s_memtime(); const int sbase = job.scale_image ? 8 : 0;
    if (tid == 0) atomicAdd(&g_roi_phase[sbase + 4], 1ull);
#endif
    if (job.scale_image) {
        // cvHaarDetectObjectsForROC, CV_HAAR_SCALE_IMAGE branch, one factor: resize, integrate, scan the unscaled window on a fixed grid
        const int szw = st.szw, szh = st.szh, P = szw + 1;
        const int *xofs = (const int *)(tabs + st.xofs_off), *yofs = (const int *)(tabs + st.yofs_off);
        const short *ialpha = (const short *)(tabs + st.ialpha_off), *ibeta = (const short *)(tabs + st.ibeta_off);
        for (int i = tid; i < szw * szh; i += kRoiThreads) {
            const int y = i / szw, x = i - y * szw;
            L.lev[i] = (uint8_t)resize_sample_cn(img, job.h, job.stride, 1, st.mode, xofs, ialpha, yofs, ibeta, st.xmax, x, y, 0);
        }
        __syncthreads();
        roi_integral([&](int x, int y) { return L.lev[y * szw + x]; }, szw, szh, L.s, L.q, P);
        const int nx = (st.endX - st.startX + st.step - 1) / st.step, ny = (st.endY - st.startY + st.step - 1) / st.step;   // origins 0, step, 2 step, ...
        ROI_STAMP(0);
        roi_step_windows(job, st, li, nx, ny, P, L, [&](int gx, int gy, int &x, int &y) { x = st.startX + gx * st.step; y = st.startY + gy * st.step; }, hits, hit_cap, rejbuf ROI_STAMP_PASS);
        return;
    }
    // scale-cascade scan, one ladder step: the image's integral pair, the features scaled by the step's factor, stride max(2, factor), adaptive x step
    const int P = job.w + 1;
    roi_integral([&](int x, int y) { return img[(size_t)y * job.stride + x]; }, job.w, job.h, L.s, L.q, P);
    ROI_STAMP(0);
    roi_step_windows(job, st, li, st.endX - st.startX, st.endY - st.startY, P, L,
                     [&](int gx, int gy, int &x, int &y) { x = roi_cvround((st.startX + gx) * st.ystep); y = roi_cvround((st.startY + gy) * st.ystep); }, hits, hit_cap, rejbuf ROI_STAMP_PASS);
}

void launch_roi(hipStream_t st, const RoiJobDev *jobs, int nsteps, const RoiStep *steps, const unsigned char *tabs, unsigned long long *hits,
                unsigned hit_cap, int plane_words, int lds_bytes, unsigned long long *rej)
{
    NVCA_LAUNCH(k_roi, dim3(nsteps), dim3(kRoiThreads), (size_t)lds_bytes, st, jobs, steps, tabs, hits, hit_cap, plane_words, rej);
}
#ifdef NVCA_STAMPS
void roi_stamps_dump(const char *path)
{
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_roi_phase), sizeof(h)) != hipSuccess) return;
    if (FILE *f = fopen(path, "w")) {
        const char *names[2] = {"scale-cascade steps", "scale-image levels"};
        for (int k = 0; k < 2; k++) {
            const unsigned long long *p = h + 8 * k;
            const double wg = p[4] ? (double)p[4] : 1.;
            fprintf(f, "%s: %llu workgroups; ticks per workgroup: image + integral %.0f, A (normaliser + stage 0) %.0f, B (stages 1-3) %.0f, C (late stages) %.0f; windows into C per workgroup %.2f\n",
                    names[k], p[4], p[0] / wg, p[1] / wg, p[2] / wg, p[3] / wg, p[5] / wg);
        }
        fclose(f);
    }
}
#endif
int roi_grant_lds(int bytes)
{
    static std::mutex mu;
    static int granted[64] = {0};
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0; (void)hipGetDevice(&dev); dev &= 63;
    if (bytes <= granted[dev]) return 0;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_roi), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    granted[dev] = bytes;
    return 0;
}

} // namespace nvca
