// kernels_cascade.hip -- the Haar cascade evaluator (scale-cascade variant,
// flags without CV_HAAR_SCALE_IMAGE): replaces the per-window loop of
// HaarDetectObjects_ScaleCascade_Invoker + cvRunHaarClassifierCascadeSum
// (OpenCV 2.4 haar.cpp) behind cascade->detectMultiScale at
// FACE/kmsfacedetect.cpp:809-811.
//
// One workgroup = one strip (a few scan rows of one scale, <= 2048 windows):
//   A. stage 0 + window variance for EVERY window of the strip (dense, one
//      window per lane); stage-0 rejects are recorded as a bit string.
//   B. OpenCV's adaptive x step (ix += result != 0 ? 1 : 2) is resolved in closed
//      form: window j is visited iff the run of stage-0 rejects immediately before
//      it in its row has even length.  Visited survivors are compacted into LDS.
//   C. later stages run on the compacted queue, re-compacted after every stage, so
//      lanes stay dense while most windows die early.
// Stage/rect tables are wave-uniform -> scalar loads; window sums are gathers
// from the integral planes (L2 / Infinity Cache resident).  No MFMA: integer
// rect sums, f32 products, f64 stage sums, exactly the reference's arithmetic
// (compiled with -ffp-contract=off).
#include "nvca_internal.h"

namespace nvca {

__device__ __forceinline__ int rect_sum(const int *__restrict__ sum, int off, const int *p)
{
    return sum[off + p[0]] - sum[off + p[1]] - sum[off + p[2]] + sum[off + p[3]];
}

// one stage on one window; recs are wave-uniform
template <bool PAIR>
__device__ __forceinline__ bool eval_stage(const int *__restrict__ sum, int off, double vnf,
                                           const StumpRec *__restrict__ recs, int count, float stage_thr)
{
    double stage_sum = 0.0;
    for (int j = 0; j < count; j++) {
        const StumpRec &f = recs[j];
        const int s0 = rect_sum(sum, off, f.p[0]);
        const int s1 = rect_sum(sum, off, f.p[1]);
        const double t = (double)f.thr * vnf;
        double v;
        if (PAIR) {
            const float fs = (float)s0 * f.w[0] + (float)s1 * f.w[1];
            v = (double)fs;
        } else {
            v = (double)((float)s0 * f.w[0]);
            v += (double)((float)s1 * f.w[1]);
            if (f.nrect == 3) {
                const int s2 = rect_sum(sum, off, f.p[2]);
                v += (double)((float)s2 * f.w[2]);
            }
        }
        stage_sum += (double)(v >= t ? f.a1 : f.a0);
    }
    return !(stage_sum < (double)stage_thr);
}

__device__ __forceinline__ bool run_stage(const int *__restrict__ sum, int off, double vnf,
                                          const StumpRec *__restrict__ recs, const StageRec &st, int pair_policy)
{
    if (pair_policy && st.two_rects) return eval_stage<true>(sum, off, vnf, recs + st.first, st.count, st.thr);
    return eval_stage<false>(sum, off, vnf, recs + st.first, st.count, st.thr);
}

__global__ __launch_bounds__(256) void k_cascade_sc(CascadeArgs a)
{
    __shared__ unsigned long long failbits[kStripMaxWin / 64];
    __shared__ double vnf_s[kStripMaxWin];
    __shared__ unsigned short q[2][kStripMaxWin];
    __shared__ int qn[2];
    __shared__ unsigned gbase_s;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, frame-major: frames are worked through in dispatch order (few integral planes live
    // at a time), and within a frame slot%8 (= XCD) selects a contiguous equal-work run of strips
    const int slot = blockIdx.x / a.blocks_per_frame;
    const int sidx = a.order[blockIdx.x - slot * a.blocks_per_frame];
    if (sidx < 0) return;
    const StripRec strip = a.strips[sidx];
    const ScaleRec &sc = a.scales[strip.scale];
    const int endX = sc.endX, nwin = strip.nrows * endX;
    const int *__restrict__ sum = a.sum + (size_t)slot * a.sum_slot;
    const unsigned long long *__restrict__ sq = a.sqsum + (size_t)slot * a.sum_slot;
    const StumpRec *__restrict__ recs = a.stumps + sc.stump_off;
    const int *__restrict__ xpos = a.pos + sc.xpos_off;
    const int *__restrict__ ypos = a.pos + sc.ypos_off + strip.iy0;
    const int e0 = sc.eq[0], e1 = sc.eq[1], e2 = sc.eq[2], e3 = sc.eq[3];
    const double inv_area = sc.inv_area;
    const StageRec st0 = a.stages[0];

    if (tid < 2) qn[tid] = 0;

    // ---- A: variance + stage 0, dense
    for (int base = 0; base < nwin; base += 256) {
        const int w = base + tid;
        const bool active = w < nwin;
        bool pass0 = false;
        if (active) {
            const int r = w / endX, ix = w - r * endX;
            const int off = ypos[r] * a.spitch + xpos[ix];
            const int ws = sum[off + e0] - sum[off + e1] - sum[off + e2] + sum[off + e3];
            const double mean = (double)ws * inv_area;
            double vnf = (double)sq[off + e0] - (double)sq[off + e1] - (double)sq[off + e2] + (double)sq[off + e3];
            vnf = vnf * inv_area - mean * mean;
            vnf = vnf >= 0. ? sqrt(vnf) : 1.;
            vnf_s[w] = vnf;
            pass0 = run_stage(sum, off, vnf, recs, st0, a.pair_policy);
        }
        const unsigned long long fb = __ballot(active && !pass0);
        if (lane == 0) failbits[(base >> 6) + wave] = fb;
    }
    __syncthreads();

    // ---- B: adaptive-step reachability + compaction of visited survivors
    for (int base = 0; base < nwin; base += 256) {
        const int w = base + tid;
        bool keep = false;
        if (w < nwin) {
            const bool fail = (failbits[w >> 6] >> (w & 63)) & 1ull;
            if (!fail) {
                const int r = w / endX, ix = w - r * endX;
                int d = 0, pos = w, remaining = ix;
                while (remaining > 0) {
                    const int p = pos - 1, b = p & 63;
                    const unsigned long long m = failbits[p >> 6] << (63 - b);   // bit p at the MSB
                    int ones = (~m == 0ull) ? 64 : __clzll((long long)~m);
                    int lim = b + 1 < remaining ? b + 1 : remaining;
                    if (ones > lim) ones = lim;
                    d += ones;
                    if (ones < lim) break;
                    pos -= ones; remaining -= ones;
                }
                keep = !(d & 1);
            }
        }
        const unsigned long long km = __ballot(keep);
        if (km) {
            int wbase = 0;
            if (lane == 0) wbase = atomicAdd(&qn[0], __popcll(km));
            wbase = __shfl(wbase, 0);
            if (keep) q[0][wbase + __popcll(km & ((1ull << lane) - 1ull))] = (unsigned short)w;
        }
    }

    // ---- C: remaining stages on the compacted queue
    int cur = 0;
    for (int s = 1; s < a.nstages; s++) {
        __syncthreads();
        const int n = qn[cur];
        if (n == 0) break;
        __syncthreads();
        if (tid == 0) qn[cur ^ 1] = 0;
        __syncthreads();
        const StageRec st = a.stages[s];
        for (int base = 0; base < n; base += 256) {
            const int i = base + tid;
            bool pass = false; int w = 0;
            if (i < n) {
                w = q[cur][i];
                const int r = w / endX, ix = w - r * endX;
                const int off = ypos[r] * a.spitch + xpos[ix];
                pass = run_stage(sum, off, vnf_s[w], recs, st, a.pair_policy);
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
    if (tid == 0) gbase_s = (unsigned)atomicAdd(a.hits, (unsigned long long)nh);
    __syncthreads();
    const unsigned gb = gbase_s;
    for (int i = tid; i < nh; i += 256) {
        const int w = q[cur][i];
        const int r = w / endX, ix = w - r * endX;
        const unsigned key = ((unsigned)strip.scale << 26) | ((unsigned)(strip.iy0 + r) << 13) | (unsigned)ix;
        if (gb + i < a.hit_cap) a.hits[1 + gb + i] = ((unsigned long long)slot << 32) | key;
    }
}

void launch_cascade_sc(hipStream_t st, const CascadeArgs &a, int batch)
{
    if (a.blocks_per_frame <= 0 || batch <= 0) return;
    hipLaunchKernelGGL(k_cascade_sc, dim3((unsigned)a.blocks_per_frame * (unsigned)batch), dim3(256), 0, st, a);
}

} // namespace nvca
