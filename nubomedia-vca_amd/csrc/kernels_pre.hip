// kernels_pre.hip -- gfx950 kernels for the image preparation in front of the
// cascade: cv::resize + cv::cvtColor + cv::equalizeHist + cv::integral as the
// reference calls them (FACE/kmsfacedetect.cpp:805-811; integral inside
// detectMultiScale).  All integer; HBM-bound streaming work.
//
// Data layout (per batch slot): gray u8 [h][gpitch]; sum i32 [(h+1)][spitch];
// sqsum [(h+1)][spitch] as a u32 low-word plane + a u8 high-byte plane (exact integers below 2^40 -> bit-identical to OpenCV's f64);
// band partials u32 [nbands][bpitch] for column sums of pixel and pixel^2.
#include "nvca_internal.h"

namespace nvca {

static constexpr int kGrayRows = 8;        // rows per block in the gray kernels

__device__ __forceinline__ int gray_of(int b, int g, int r)
{   // RGB2Gray<uchar>: B2Y 1868, G2Y 9617, R2Y 4899, shift 14, rounding 1<<13
    return (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14;
}

__device__ __forceinline__ void hist_flush(unsigned (*lh)[256], unsigned *hist, int tid)
{
    __syncthreads();
    unsigned v = lh[0][tid] + lh[1][tid] + lh[2][tid] + lh[3][tid];
    if (v) atomicAdd(&hist[tid], v);
}

// ---- K1 generic: one output pixel per thread, any resize mode, any alignment.
// mode 0 identity, 1 bilinear (fixed point, 11-bit coefficients), 2 area 2x2.
__global__ __launch_bounds__(256) void k_gray_generic(
    const uint8_t *const *__restrict__ srcs, PreGeom g, int mode,
    const int *__restrict__ xofs, const short *__restrict__ ialpha,
    const int *__restrict__ yofs, const short *__restrict__ ibeta, int xmax,
    uint8_t *__restrict__ gray, unsigned *__restrict__ hist)
{
    __shared__ unsigned lh[4][256];
    const int tid = threadIdx.x, wave = tid >> 6, slot = blockIdx.z;
    for (int i = tid; i < 1024; i += 256) (&lh[0][0])[i] = 0;
    __syncthreads();
    const uint8_t *src = srcs[slot];
    const int cn = g.cn;
    const int x = blockIdx.x * 256 + tid;
    uint8_t *grow = gray + (size_t)slot * g.gray_slot;
    for (int ry = 0; ry < kGrayRows; ry++) {
        const int y = blockIdx.y * kGrayRows + ry;
        if (y >= g.h) break;
        if (x < g.w) {
            int B, G, R;
            if (mode == 0) {
                const uint8_t *s = src + (size_t)y * g.sstride + (size_t)x * cn;
                B = s[0]; G = s[1]; R = s[2];
            } else if (mode == 2) {
                const uint8_t *s0 = src + (size_t)(2 * y) * g.sstride + (size_t)(2 * x) * cn;
                const uint8_t *s1 = s0 + g.sstride;
                B = (s0[0] + s0[cn] + s1[0] + s1[cn] + 2) >> 2;
                G = (s0[1] + s0[cn + 1] + s1[1] + s1[cn + 1] + 2) >> 2;
                R = (s0[2] + s0[cn + 2] + s1[2] + s1[cn + 2] + 2) >> 2;
            } else {
                int sy0 = yofs[y], sy1 = sy0 + 1;
                sy0 = sy0 >= 0 ? (sy0 < g.sh ? sy0 : g.sh - 1) : 0;
                sy1 = sy1 >= 0 ? (sy1 < g.sh ? sy1 : g.sh - 1) : 0;
                const int sx = xofs[x] * cn;
                const uint8_t *s0 = src + (size_t)sy0 * g.sstride + sx;
                const uint8_t *s1 = src + (size_t)sy1 * g.sstride + sx;
                const int b0 = ibeta[2 * y], b1 = ibeta[2 * y + 1];
                int c[3];
                if (x < xmax) {
                    const int a0 = ialpha[2 * x], a1 = ialpha[2 * x + 1];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        int h0 = s0[k] * a0 + s0[cn + k] * a1;
                        int h1 = s1[k] * a0 + s1[cn + k] * a1;
                        c[k] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        int h0 = s0[k] * 2048, h1 = s1[k] * 2048;
                        c[k] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                    }
                }
                B = c[0] & 255; G = c[1] & 255; R = c[2] & 255;
            }
            const int v = gray_of(B, G, R);
            grow[(size_t)y * g.gpitch + x] = (uint8_t)v;
            if (hist) atomicAdd(&lh[wave][v], 1u);
        }
    }
    if (hist) hist_flush(lh, hist + slot * 256, tid);
}

// ---- K1 fast path: identity geometry, rows and base 4-byte aligned: 4 pixels per thread.
template <int CN>
__global__ __launch_bounds__(256) void k_gray_fast4(
    const uint8_t *const *__restrict__ srcs, PreGeom g, uint8_t *__restrict__ gray,
    unsigned *__restrict__ hist)
{
    __shared__ unsigned lh[4][256];
    const int tid = threadIdx.x, wave = tid >> 6, slot = blockIdx.z;
    for (int i = tid; i < 1024; i += 256) (&lh[0][0])[i] = 0;
    __syncthreads();
    const uint8_t *src = srcs[slot];
    const int x4 = (blockIdx.x * 256 + tid) * 4;
    uint8_t *grow = gray + (size_t)slot * g.gray_slot;
    if (x4 < g.w) {
        for (int ry = 0; ry < kGrayRows; ry++) {
            const int y = blockIdx.y * kGrayRows + ry;
            if (y >= g.h) break;
            const unsigned *s = (const unsigned *)(src + (size_t)y * g.sstride + (size_t)x4 * CN);
            int v[4];
            if (x4 + 4 <= g.w) {
                if (CN == 3) {
                    const unsigned d0 = s[0], d1 = s[1], d2 = s[2];
                    v[0] = gray_of(d0 & 255, (d0 >> 8) & 255, (d0 >> 16) & 255);
                    v[1] = gray_of(d0 >> 24, d1 & 255, (d1 >> 8) & 255);
                    v[2] = gray_of((d1 >> 16) & 255, d1 >> 24, d2 & 255);
                    v[3] = gray_of((d2 >> 8) & 255, (d2 >> 16) & 255, d2 >> 24);
                } else {
                    const uint4 d = *(const uint4 *)s;
                    v[0] = gray_of(d.x & 255, (d.x >> 8) & 255, (d.x >> 16) & 255);
                    v[1] = gray_of(d.y & 255, (d.y >> 8) & 255, (d.y >> 16) & 255);
                    v[2] = gray_of(d.z & 255, (d.z >> 8) & 255, (d.z >> 16) & 255);
                    v[3] = gray_of(d.w & 255, (d.w >> 8) & 255, (d.w >> 16) & 255);
                }
                *(unsigned *)(grow + (size_t)y * g.gpitch + x4) =
                    (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24);
                if (hist) {
                    atomicAdd(&lh[wave][v[0]], 1u); atomicAdd(&lh[wave][v[1]], 1u);
                    atomicAdd(&lh[wave][v[2]], 1u); atomicAdd(&lh[wave][v[3]], 1u);
                }
            } else {
                const uint8_t *sb = (const uint8_t *)s;
                for (int k = 0; x4 + k < g.w; k++) {
                    const int vv = gray_of(sb[k * CN], sb[k * CN + 1], sb[k * CN + 2]);
                    grow[(size_t)y * g.gpitch + x4 + k] = (uint8_t)vv;
                    if (hist) atomicAdd(&lh[wave][vv], 1u);
                }
            }
        }
    }
    if (hist) hist_flush(lh, hist + slot * 256, tid);
}

void launch_gray(hipStream_t st, const uint8_t *const *d_src, const PreGeom &g, int mode,
                 const int *d_xofs, const short *d_ialpha, const int *d_yofs, const short *d_ibeta, int xmax,
                 uint8_t *gray, unsigned *hist, int batch, bool aligned4)
{
    const int gy = (g.h + kGrayRows - 1) / kGrayRows;
    if (mode == 0 && aligned4 && (g.cn == 3 || g.cn == 4)) {
        dim3 grid((g.w + 1023) / 1024, gy, batch);
        if (g.cn == 3) NVCA_LAUNCH(k_gray_fast4<3>, grid, dim3(256), 0, st, d_src, g, gray, hist);
        else           NVCA_LAUNCH(k_gray_fast4<4>, grid, dim3(256), 0, st, d_src, g, gray, hist);
    } else {
        dim3 grid((g.w + 255) / 256, gy, batch);
        NVCA_LAUNCH(k_gray_generic, grid, dim3(256), 0, st, d_src, g, mode, d_xofs, d_ialpha, d_yofs,
                           d_ibeta, xmax, gray, hist);
    }
}

// ---- 8UC1 resize (gray-then-resize order of the part detectors, pyramid levels)
// one destination sample of cv::resize(INTER_LINEAR) 8UC1 (mode 0: copy, 2: exact 2x area-fast, 1: fixed-point bilinear)
// `px(row, col)`: the source sample (a gray byte, a gray byte through a LUT, or the gray value of a BGR pixel)
template <class Px>
__device__ __forceinline__ int resize1_sample(Px px, int sh, int mode,
                                              const int *__restrict__ xofs, const short *__restrict__ ialpha,
                                              const int *__restrict__ yofs, const short *__restrict__ ibeta, int xmax, int x, int y)
{
    if (mode == 0) return px(y, x);
    if (mode == 2) return (px(2 * y, 2 * x) + px(2 * y, 2 * x + 1) + px(2 * y + 1, 2 * x) + px(2 * y + 1, 2 * x + 1) + 2) >> 2;
    int sy0 = yofs[y], sy1 = sy0 + 1;
    sy0 = sy0 >= 0 ? (sy0 < sh ? sy0 : sh - 1) : 0;
    sy1 = sy1 >= 0 ? (sy1 < sh ? sy1 : sh - 1) : 0;
    const int sx = xofs[x];
    const int b0 = ibeta[2 * y], b1 = ibeta[2 * y + 1];
    int h0, h1;
    if (x < xmax) {
        const int a0 = ialpha[2 * x], a1 = ialpha[2 * x + 1];
        h0 = px(sy0, sx) * a0 + px(sy0, sx + 1) * a1; h1 = px(sy1, sx) * a0 + px(sy1, sx + 1) * a1;
    } else { h0 = px(sy0, sx) * 2048; h1 = px(sy1, sx) * 2048; }
    return ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 255;
}
__device__ __forceinline__ int resize1_value(const uint8_t *__restrict__ src, int sh, int sstride, int mode,
                                             const int *__restrict__ xofs, const short *__restrict__ ialpha,
                                             const int *__restrict__ yofs, const short *__restrict__ ibeta, int xmax, int x, int y)
{
    return resize1_sample([&](int r, int c) { return (int)src[(size_t)r * sstride + c]; }, sh, mode, xofs, ialpha, yofs, ibeta, xmax, x, y);
}

__global__ __launch_bounds__(256) void k_resize1(
    const uint8_t *__restrict__ src, int sw, int sh, int sstride, int mode,
    const int *__restrict__ xofs, const short *__restrict__ ialpha,
    const int *__restrict__ yofs, const short *__restrict__ ibeta, int xmax,
    uint8_t *__restrict__ dst, int dw, int dh, int dstride, unsigned *__restrict__ hist, size_t src_slot, size_t dst_slot)
{
    __shared__ unsigned lh[4][256];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 1024; i += 256) (&lh[0][0])[i] = 0;
    __syncthreads();
    src += (size_t)blockIdx.z * src_slot; dst += (size_t)blockIdx.z * dst_slot;      // several images of one geometry
    if (hist) hist += (size_t)blockIdx.z * 256;
    const int x = blockIdx.x * 256 + tid;
    for (int ry = 0; ry < kGrayRows; ry++) {
        const int y = blockIdx.y * kGrayRows + ry;
        if (y >= dh) break;
        if (x < dw) {
            const int v = resize1_value(src, sh, sstride, mode, xofs, ialpha, yofs, ibeta, xmax, x, y);
            dst[(size_t)y * dstride + x] = (uint8_t)v;
            if (hist) atomicAdd(&lh[wave][v], 1u);
        }
    }
    if (hist) hist_flush(lh, hist, tid);
}

// ---- 8UC3 resize (cv::resize on the BGR frame, FACE/kmsfacedetect.cpp:805, as a stand-alone primitive;
// the face stream fuses it with BGR2GRAY in k_gray_generic)
__global__ __launch_bounds__(256) void k_resize3(
    const uint8_t *__restrict__ src, int sw, int sh, int sstride, int mode,
    const int *__restrict__ xofs, const short *__restrict__ ialpha,
    const int *__restrict__ yofs, const short *__restrict__ ibeta, int xmax,
    uint8_t *__restrict__ dst, int dw, int dh, int dstride)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    for (int ry = 0; ry < kGrayRows; ry++) {
        const int y = blockIdx.y * kGrayRows + ry;
        if (y >= dh || x >= dw) continue;
        uint8_t *d = dst + (size_t)y * dstride + (size_t)x * 3;
        if (mode == 0) {
            const uint8_t *s = src + (size_t)y * sstride + (size_t)x * 3;
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
        } else if (mode == 2) {
            const uint8_t *s0 = src + (size_t)(2 * y) * sstride + (size_t)(2 * x) * 3, *s1 = s0 + sstride;
#pragma unroll
            for (int k = 0; k < 3; k++) d[k] = (uint8_t)((s0[k] + s0[3 + k] + s1[k] + s1[3 + k] + 2) >> 2);
        } else {
            int sy0 = yofs[y], sy1 = sy0 + 1;
            sy0 = sy0 >= 0 ? (sy0 < sh ? sy0 : sh - 1) : 0;
            sy1 = sy1 >= 0 ? (sy1 < sh ? sy1 : sh - 1) : 0;
            const int sx = xofs[x] * 3;
            const uint8_t *s0 = src + (size_t)sy0 * sstride + sx, *s1 = src + (size_t)sy1 * sstride + sx;
            const int b0 = ibeta[2 * y], b1 = ibeta[2 * y + 1];
            const bool inner = x < xmax;
            const int a0 = inner ? ialpha[2 * x] : 2048, a1 = inner ? ialpha[2 * x + 1] : 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int h0 = s0[k] * a0 + (inner ? s0[3 + k] * a1 : 0), h1 = s1[k] * a0 + (inner ? s1[3 + k] * a1 : 0);
                d[k] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
}
void launch_resize3(hipStream_t st, const uint8_t *src, int sw, int sh, int sstride, int mode,
                    const int *d_xofs, const short *d_ialpha, const int *d_yofs, const short *d_ibeta,
                    int xmax, uint8_t *dst, int dw, int dh, int dstride)
{
    dim3 grid((dw + 255) / 256, (dh + kGrayRows - 1) / kGrayRows, 1);
    NVCA_LAUNCH(k_resize3, grid, dim3(256), 0, st, src, sw, sh, sstride, mode, d_xofs, d_ialpha, d_yofs, d_ibeta, xmax,
                       dst, dw, dh, dstride);
}

// ---- image-to-overlay on a device frame: one thread per pixel of the scaled overlay image of one box
__global__ __launch_bounds__(256) void k_overlay(uint8_t *__restrict__ frame, int W, int H, int stride, OverlayPlace p, const uint8_t *__restrict__ img,
                                                 int ih, int istride, int cn, int mode, const int *__restrict__ xofs, const short *__restrict__ ialpha,
                                                 const int *__restrict__ yofs, const short *__restrict__ ibeta, int xmax)
{
    const int w = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
    if (w >= p.w || h >= p.h || w + p.x < 0 || w + p.x >= W || h + p.y < 0 || h + p.y >= H) return;
    int v[4] = {0, 0, 0, 0};
    for (int k = 0; k < cn; k++) v[k] = resize_sample_cn(img, ih, istride, cn, mode, xofs, ialpha, yofs, ibeta, xmax, w, h, k);
    overlay_pixel(frame + (size_t)(h + p.y) * stride + (size_t)(w + p.x) * 3, v, cn);
}
void launch_overlay(hipStream_t st, uint8_t *frame, int W, int H, int stride, const OverlayPlace &p, const uint8_t *img, int ih, int istride, int cn,
                    int mode, const int *xofs, const short *ialpha, const int *yofs, const short *ibeta, int xmax)
{
    NVCA_LAUNCH(k_overlay, dim3((p.w + 255) / 256, p.h), dim3(256), 0, st, frame, W, H, stride, p, img, ih, istride, cn, mode, xofs, ialpha, yofs, ibeta, xmax);
}

void launch_resize1(hipStream_t st, const uint8_t *src, int sw, int sh, int sstride, int mode,
                    const int *d_xofs, const short *d_ialpha, const int *d_yofs, const short *d_ibeta,
                    int xmax, uint8_t *dst, int dw, int dh, int dstride, unsigned *hist, int batch, size_t src_slot, size_t dst_slot)
{
    dim3 grid((dw + 255) / 256, (dh + kGrayRows - 1) / kGrayRows, batch);
    NVCA_LAUNCH(k_resize1, grid, dim3(256), 0, st, src, sw, sh, sstride, mode, d_xofs, d_ialpha, d_yofs,
                       d_ibeta, xmax, dst, dw, dh, dstride, hist, src_slot, dst_slot);
}

// ---- working images of the part detectors, all frames of a batched call in one launch: image z of the launch is
// resize(gray(frame z)) (BGR = true: the gray value of a source pixel is computed where the resize reads it -- cvtColor then
// resize, EYE/kmseyedetect.cpp:948-956, NOSE/kmsnosedetect.cpp:836-841 -- without writing the full-size gray image) or
// resize(lut[gray z]) (the eye detector equalizes the full-size gray image first, EYE :950), plus its histogram.
template <bool BGR>
__global__ __launch_bounds__(256) void k_work_resize(
    const uint8_t *const *__restrict__ srcs, const int *__restrict__ lut_idx, const uint8_t *__restrict__ luts,
    int sh, int sstride, int mode, const int *__restrict__ xofs, const short *__restrict__ ialpha,
    const int *__restrict__ yofs, const short *__restrict__ ibeta, int xmax,
    uint8_t *__restrict__ dst, int dw, int dh, int dstride, size_t dst_slot, unsigned *__restrict__ hist)
{
    __shared__ unsigned lh[4][256];
    __shared__ uint8_t sl[256];
    const int tid = threadIdx.x, wave = tid >> 6, z = blockIdx.z;
    for (int i = tid; i < 1024; i += 256) (&lh[0][0])[i] = 0;
    const uint8_t *__restrict__ src = srcs[z];
    const bool use_lut = !BGR && lut_idx != nullptr;
    if (use_lut) sl[tid] = luts[(size_t)lut_idx[z] * 256 + tid];
    __syncthreads();
    dst += (size_t)z * dst_slot;
    const int x = blockIdx.x * 256 + tid;
    for (int ry = 0; ry < kGrayRows; ry++) {
        const int y = blockIdx.y * kGrayRows + ry;
        if (y >= dh) break;
        if (x < dw) {
            int v;
            if (BGR) v = resize1_sample([&](int r, int c) { const uint8_t *p = src + (size_t)r * sstride + (size_t)c * 3; return gray_of(p[0], p[1], p[2]); },
                                        sh, mode, xofs, ialpha, yofs, ibeta, xmax, x, y);
            else if (use_lut) v = resize1_sample([&](int r, int c) { return (int)sl[src[(size_t)r * sstride + c]]; }, sh, mode, xofs, ialpha, yofs, ibeta, xmax, x, y);
            else v = resize1_value(src, sh, sstride, mode, xofs, ialpha, yofs, ibeta, xmax, x, y);
            dst[(size_t)y * dstride + x] = (uint8_t)v;
            if (hist) atomicAdd(&lh[wave][v], 1u);
        }
    }
    if (hist) hist_flush(lh, hist + (size_t)z * 256, tid);
}
void launch_work_resize(hipStream_t st, bool bgr, const uint8_t *const *d_srcs, const int *d_lut_idx, const uint8_t *d_luts, int sh, int sstride,
                        int mode, const int *d_xofs, const short *d_ialpha, const int *d_yofs, const short *d_ibeta, int xmax,
                        uint8_t *dst, int dw, int dh, int dstride, size_t dst_slot, unsigned *hist, int batch)
{
    dim3 grid((dw + 255) / 256, (dh + kGrayRows - 1) / kGrayRows, batch);
    if (bgr) NVCA_LAUNCH(k_work_resize<true>, grid, dim3(256), 0, st, d_srcs, d_lut_idx, d_luts, sh, sstride, mode, d_xofs, d_ialpha, d_yofs, d_ibeta, xmax,
                         dst, dw, dh, dstride, dst_slot, hist);
    else NVCA_LAUNCH(k_work_resize<false>, grid, dim3(256), 0, st, d_srcs, d_lut_idx, d_luts, sh, sstride, mode, d_xofs, d_ialpha, d_yofs, d_ibeta, xmax,
                     dst, dw, dh, dstride, dst_slot, hist);
}

// ---- K2: equalizeHist LUT from the histogram (one block per slot)
// `rezero`: the histogram is cleared again once read (the next frame's gray kernel accumulates into it) and the two
// list counters of the cascade that follows are reset -- three fill launches less per batch.
__global__ __launch_bounds__(256) void k_lut(unsigned *__restrict__ hist, int total, uint8_t *__restrict__ lut, int rezero,
                                             unsigned long long *__restrict__ zero_a, unsigned long long *__restrict__ zero_b)
{
    __shared__ unsigned wsum[4];
    __shared__ unsigned long long wmask[4];
    __shared__ unsigned hs[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, slot = blockIdx.x;
    const unsigned h = hist[slot * 256 + tid];
    hs[tid] = h;
    if (rezero) hist[slot * 256 + tid] = 0;
    if (slot == 0 && tid == 0) { if (zero_a) *zero_a = 0; if (zero_b) *zero_b = 0; }
    unsigned incl = h;
    for (int d = 1; d < 64; d <<= 1) { unsigned t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    const unsigned long long m = __ballot(h != 0);
    if (lane == 63) wsum[wave] = incl;
    if (lane == 0) wmask[wave] = m;
    __syncthreads();
    for (int j = 0; j < wave; j++) incl += wsum[j];
    int first = 256;
    for (int j = 3; j >= 0; j--) if (wmask[j]) first = j * 64 + __ffsll((long long)wmask[j]) - 1;
    uint8_t out;
    if (first == 256) out = 0;
    else {
        const unsigned hf = hs[first];
        if (hf == (unsigned)total) out = (uint8_t)first;         // dst.setTo(i)
        else if (tid <= first) out = 0;
        else {
            const float scale = 255.f / (float)(int)(total - (int)hf);   // (hist_sz-1.f)/(total-hist[i])
            const float v = (float)(int)(incl - hf) * scale;             // sum*scale, int -> float
            int iv = (int)rintf(v);                                      // saturate_cast<uchar>: cvRound + clamp
            out = (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
        }
    }
    lut[slot * 256 + tid] = out;
}

void launch_lut(hipStream_t st, unsigned *hist, int total, uint8_t *lut, int batch, int rezero, unsigned long long *zero_a,
                unsigned long long *zero_b)
{
    NVCA_LAUNCH(k_lut, dim3(batch), dim3(256), 0, st, hist, total, lut, rezero, zero_a, zero_b);
}

__global__ __launch_bounds__(256) void k_hist(const uint8_t *__restrict__ gray, int w, int h, int pitch,
                                              unsigned *__restrict__ hist)
{
    __shared__ unsigned lh[4][256];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 1024; i += 256) (&lh[0][0])[i] = 0;
    __syncthreads();
    const int x = blockIdx.x * 256 + tid;
    for (int ry = 0; ry < kGrayRows; ry++) {
        const int y = blockIdx.y * kGrayRows + ry;
        if (y < h && x < w) atomicAdd(&lh[wave][gray[(size_t)y * pitch + x]], 1u);
    }
    hist_flush(lh, hist, tid);
}
void launch_hist(hipStream_t st, const uint8_t *gray, int w, int h, int pitch, unsigned *hist)
{
    dim3 grid((w + 255) / 256, (h + kGrayRows - 1) / kGrayRows, 1);
    NVCA_LAUNCH(k_hist, grid, dim3(256), 0, st, gray, w, h, pitch, hist);
}

__global__ __launch_bounds__(256) void k_apply_lut(const uint8_t *__restrict__ src, int w, int h, int spitch,
                                                   const uint8_t *__restrict__ lut, uint8_t *__restrict__ dst, int dpitch,
                                                   size_t src_slot, size_t dst_slot)
{
    __shared__ uint8_t sl[256];
    sl[threadIdx.x] = lut[(size_t)blockIdx.z * 256 + threadIdx.x];          // image z of the launch: its own LUT and slot
    src += (size_t)blockIdx.z * src_slot; dst += (size_t)blockIdx.z * dst_slot;
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x;
    for (int ry = 0; ry < kGrayRows; ry++) {
        const int y = blockIdx.y * kGrayRows + ry;
        if (y < h && x < w) dst[(size_t)y * dpitch + x] = sl[src[(size_t)y * spitch + x]];
    }
}
void launch_apply_lut(hipStream_t st, const uint8_t *src, int w, int h, int spitch, const uint8_t *lut,
                      uint8_t *dst, int dpitch, int batch, size_t src_slot, size_t dst_slot)
{
    dim3 grid((w + 255) / 256, (h + kGrayRows - 1) / kGrayRows, batch);
    NVCA_LAUNCH(k_apply_lut, grid, dim3(256), 0, st, src, w, h, spitch, lut, dst, dpitch, src_slot, dst_slot);
}

// ---- cv::flip(src, dst, 1) (EAR/kmseardetect.cpp:800)
__global__ __launch_bounds__(256) void k_flip_h(const uint8_t *__restrict__ src, int w, int h, int spitch,
                                                uint8_t *__restrict__ dst, int dpitch, size_t src_slot, size_t dst_slot)
{
    src += (size_t)blockIdx.z * src_slot; dst += (size_t)blockIdx.z * dst_slot;
    const int x = blockIdx.x * 256 + threadIdx.x;
    for (int ry = 0; ry < kGrayRows; ry++) {
        const int y = blockIdx.y * kGrayRows + ry;
        if (y < h && x < w) dst[(size_t)y * dpitch + x] = src[(size_t)y * spitch + (w - 1 - x)];
    }
}
void launch_flip_h(hipStream_t st, const uint8_t *src, int w, int h, int spitch, uint8_t *dst, int dpitch, int batch, size_t src_slot,
                   size_t dst_slot)
{
    dim3 grid((w + 255) / 256, (h + kGrayRows - 1) / kGrayRows, batch);
    NVCA_LAUNCH(k_flip_h, grid, dim3(256), 0, st, src, w, h, spitch, dst, dpitch, src_slot, dst_slot);
}

// ---- K3a: per-band column sums of lut[gray] and its square
__global__ __launch_bounds__(256) void k_colsum(const uint8_t *__restrict__ gray, const uint8_t *__restrict__ lut,
                                                int lut_stride, PreGeom g, unsigned *__restrict__ bandsum,
                                                unsigned *__restrict__ bandsq)
{
    __shared__ uint8_t sl[256];
    const int tid = threadIdx.x, band = blockIdx.y, slot = blockIdx.z;
    sl[tid] = lut ? lut[(size_t)slot * lut_stride + tid] : (uint8_t)tid;
    __syncthreads();
    const int x4 = (blockIdx.x * 256 + tid) * 4;
    const int bpitch = (int)(g.band_slot / g.nbands);
    if (x4 >= bpitch) return;
    const int y0 = band * kIntegralBand, y1 = min(g.h, y0 + kIntegralBand);
    unsigned s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (x4 < g.w) {
        const uint8_t *base = gray + (size_t)slot * g.gray_slot + x4;
        for (int y = y0; y < y1; y++) {
            const unsigned px = *(const unsigned *)(base + (size_t)y * g.gpitch);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned v = (x4 + k < g.w) ? sl[(px >> (8 * k)) & 255] : 0u;
                s[k] += v; q[k] += v * v;
            }
        }
    }
    const size_t o = (size_t)slot * g.band_slot + (size_t)band * bpitch + x4;
    *(uint4 *)(bandsum + o) = make_uint4(s[0], s[1], s[2], s[3]);
    *(uint4 *)(bandsq + o) = make_uint4(q[0], q[1], q[2], q[3]);
}
void launch_colsum(hipStream_t st, const uint8_t *gray, const uint8_t *lut, int lut_stride, const PreGeom &g,
                   unsigned *bandsum, unsigned *bandsq, int batch)
{
    const int bpitch = (int)(g.band_slot / g.nbands);
    dim3 grid((bpitch / 4 + 255) / 256, g.nbands, batch);
    NVCA_LAUNCH(k_colsum, grid, dim3(256), 0, st, gray, lut, lut_stride, g, bandsum, bandsq);
}

// ---- K3b: exclusive scan over bands, per column (in place)
__global__ __launch_bounds__(256) void k_bandscan(PreGeom g, unsigned *__restrict__ bandsum, unsigned *__restrict__ bandsq)
{
    const int bpitch = (int)(g.band_slot / g.nbands);
    const int x = blockIdx.x * 256 + threadIdx.x, slot = blockIdx.y;
    if (x >= bpitch) return;
    unsigned rs = 0, rq = 0;
    size_t o = (size_t)slot * g.band_slot + x;
    // 16 bands at a time: the loads of a chunk are all in flight before the first store (a load-add-store loop
    // serialises on the round trip: the stores may alias the next loads as far as the compiler knows)
    for (int b0 = 0; b0 < g.nbands; b0 += 16) {
        unsigned ts[16], tq[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const bool in = b0 + k < g.nbands;
            ts[k] = in ? bandsum[o + (size_t)k * bpitch] : 0u; tq[k] = in ? bandsq[o + (size_t)k * bpitch] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (b0 + k < g.nbands) {
                bandsum[o + (size_t)k * bpitch] = rs; bandsq[o + (size_t)k * bpitch] = rq;
                rs += ts[k]; rq += tq[k];
            }
        o += (size_t)16 * bpitch;
    }
}
void launch_bandscan(hipStream_t st, const PreGeom &g, unsigned *bandsum, unsigned *bandsq, int batch)
{
    const int bpitch = (int)(g.band_slot / g.nbands);
    NVCA_LAUNCH(k_bandscan, dim3((bpitch + 255) / 256, batch), dim3(256), 0, st, g, bandsum, bandsq);
}

// wave64 inclusive add-scan on the VALU (DPP row shifts inside each row of 16 lanes, then the three row totals
// through readlane): no LDS traffic, unlike ds_bpermute-based __shfl_up
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v, int lane)
{
    unsigned x = v;
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true);   // row_shr:3
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xe, true);   // row_shr:4, lanes 4..15 of a row
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xc, true);   // row_shr:8, lanes 8..15
    const unsigned t0 = (unsigned)__builtin_amdgcn_readlane((int)x, 15), t1 = (unsigned)__builtin_amdgcn_readlane((int)x, 31),
                   t2 = (unsigned)__builtin_amdgcn_readlane((int)x, 47);
    const int row = lane >> 4;
    return x + (row >= 1 ? t0 : 0u) + (row >= 2 ? t1 : 0u) + (row >= 3 ? t2 : 0u);
}

// ---- K3c: integral + squared integral.  One block (512 threads) per band of rows; a block scans whole
// rows (4 integral columns per thread, 2048 per pass), keeps the vertical running sums in registers and
// writes one fully contiguous 16-byte vector per thread and plane: sum (i32) and the squared integral as
// two u32 planes (low / high word) -- 64-bit values written 8 B per column would leave every store
// instruction touching a quarter of each 64-byte sector (measured: 2.7 TB/s vs 3.7 TB/s for the sum plane).
// Thread t owns integral columns X in [4t, 4t+4): value(X) = prefix up to pixel X-1
// = exclusive base of the thread (X = 4t) or base + local inclusive (X > 4t).
static constexpr int kIntThreads = 512;
static constexpr int kIntWaves = kIntThreads / 64;

__global__ __launch_bounds__(kIntThreads) void k_integral(const uint8_t *__restrict__ gray, const uint8_t *__restrict__ lut,
                                                          int lut_stride, PreGeom g, const unsigned *__restrict__ bandsum,
                                                          const unsigned *__restrict__ bandsq, int *__restrict__ sum,
                                                          unsigned *__restrict__ sq32)
{
    __shared__ uint8_t sl[256];
    __shared__ unsigned wrow_s[kIntegralBand][kIntWaves], wrow_q[kIntegralBand][kIntWaves];
    __shared__ unsigned wb_s[kIntWaves];
    __shared__ unsigned long long wb_q[kIntWaves];
    __shared__ unsigned carry_s[kIntegralBand], carry_q[kIntegralBand];
    __shared__ unsigned cbase_s;
    __shared__ unsigned long long cbase_q;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int band = blockIdx.x, slot = blockIdx.y;
    if (tid < 256) sl[tid] = lut ? lut[(size_t)slot * lut_stride + tid] : (uint8_t)tid;
    if (tid < kIntegralBand) { carry_s[tid] = 0; carry_q[tid] = 0; }
    if (tid == 0) { cbase_s = 0; cbase_q = 0; }
    __syncthreads();
    const int y0 = band * kIntegralBand, y1 = min(g.h, y0 + kIntegralBand);
    const int bpitch = (int)(g.band_slot / g.nbands);
    const uint8_t *gbase = gray + (size_t)slot * g.gray_slot;
    int *sbase = sum + (size_t)slot * g.sum_slot;
    unsigned *lbase = sq32 + (size_t)slot * 2 * g.sum_slot;
    uint8_t *hbase = (uint8_t *)(lbase + g.sum_slot);        // high bytes (bits 32..39), one per element
    const unsigned *bs = bandsum + (size_t)slot * g.band_slot + (size_t)band * bpitch;
    const unsigned *bq = bandsq + (size_t)slot * g.band_slot + (size_t)band * bpitch;
    const int nchunks = (g.w + 1 + 2047) / 2048;

    for (int c = 0; c < nchunks; c++) {
        const int X0 = c * 2048 + tid * 4;
        const bool in_pitch = X0 < g.spitch;
        // ---- base row: prefix over x of the column sums above this band
        unsigned ps[4]; unsigned long long pq[4];
        {
            uint4 a = make_uint4(0, 0, 0, 0), e = make_uint4(0, 0, 0, 0);
            if (X0 + 4 <= bpitch) { a = *(const uint4 *)(bs + X0); e = *(const uint4 *)(bq + X0); }
            const unsigned vs[4] = {a.x, a.y, a.z, a.w}, vq[4] = {e.x, e.y, e.z, e.w};
            unsigned rs = 0; unsigned long long rq = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool ok = X0 + k < g.w;
                rs += ok ? vs[k] : 0u; rq += ok ? vq[k] : 0u;
                ps[k] = rs; pq[k] = rq;
            }
        }
        const unsigned ts = ps[3]; const unsigned long long tq = pq[3];
        unsigned is = wave_incl_scan_u32(ts + (tid == 0 ? cbase_s : 0u), lane);
        unsigned long long iq = tq + (tid == 0 ? cbase_q : 0ull);
        for (int d = 1; d < 64; d <<= 1) { const unsigned long long b = __shfl_up(iq, d); if (lane >= d) iq += b; }
        if (lane == 63) { wb_s[wave] = is; wb_q[wave] = iq; }
        __syncthreads();
        for (int j = 0; j < wave; j++) { is += wb_s[j]; iq += wb_q[j]; }
        unsigned acc_s[4]; unsigned long long acc_q[4];
        {
            const unsigned es = is - ts; const unsigned long long eq = iq - tq;
            acc_s[0] = es; acc_q[0] = eq;
#pragma unroll
            for (int k = 1; k < 4; k++) { acc_s[k] = es + ps[k - 1]; acc_q[k] = eq + pq[k - 1]; }
        }
        __syncthreads();                                   // wb_* consumed
        if (tid == kIntThreads - 1) { cbase_s = is; cbase_q = iq; }
        if (band == 0 && in_pitch) {                        // integral row 0 (all zero)
            *(int4 *)(sbase + X0) = make_int4(0, 0, 0, 0);
            *(uint4 *)(lbase + X0) = make_uint4(0, 0, 0, 0);
            *(unsigned *)(hbase + X0) = 0u;
        }
        // ---- rows of the band.  All rows' pixels are requested at once; every row's prefix over x is scanned inside the waves
        // (DPP), the wave totals of all rows are exchanged through LDS behind ONE barrier (it was one per row), and the
        // vertical running sums are then carried and stored row by row without further synchronisation.
        const int nr = y1 - y0;
        unsigned px[kIntegralBand];
#pragma unroll
        for (int r = 0; r < kIntegralBand; r++)
            px[r] = (X0 < g.w && r < nr) ? *(const unsigned *)(gbase + (size_t)(y0 + r) * g.gpitch + X0) : 0u;
        unsigned inc_s[kIntegralBand], inc_q[kIntegralBand];          // inclusive prefix (inside the wave) of the thread totals, per row
#pragma unroll
        for (int r = 0; r < kIntegralBand; r++) {
            unsigned rs = 0, rq = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned v = (X0 + k < g.w) ? (unsigned)sl[(px[r] >> (8 * k)) & 255] : 0u;
                rs += v; rq += v * v;
            }
            inc_s[r] = wave_incl_scan_u32(rs, lane); inc_q[r] = wave_incl_scan_u32(rq, lane);
            if (lane == 63) { wrow_s[r][wave] = inc_s[r]; wrow_q[r][wave] = inc_q[r]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kIntegralBand; r++) {
            if (r < nr) {
            // exclusive prefix of this thread in row r: the chunks to the left (carry), the waves to the left, the lanes to the left
            unsigned e_s = carry_s[r], e_q = carry_q[r];
            for (int j = 0; j < wave; j++) { e_s += wrow_s[r][j]; e_q += wrow_q[r][j]; }
            unsigned ls[4], lq[4];
            unsigned rs = 0, rq = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned v = (X0 + k < g.w) ? (unsigned)sl[(px[r] >> (8 * k)) & 255] : 0u;
                rs += v; rq += v * v;
                ls[k] = rs; lq[k] = rq;
            }
            e_s += inc_s[r] - rs; e_q += inc_q[r] - rq;
            acc_s[0] += e_s; acc_q[0] += e_q;
#pragma unroll
            for (int k = 1; k < 4; k++) { acc_s[k] += e_s + ls[k - 1]; acc_q[k] += e_q + lq[k - 1]; }
            if (in_pitch) {
                const size_t o = (size_t)(y0 + r + 1) * g.spitch + X0;
                *(int4 *)(sbase + o) = make_int4((int)acc_s[0], (int)acc_s[1], (int)acc_s[2], (int)acc_s[3]);
                *(uint4 *)(lbase + o) = make_uint4((unsigned)acc_q[0], (unsigned)acc_q[1], (unsigned)acc_q[2], (unsigned)acc_q[3]);
                *(unsigned *)(hbase + o) = (unsigned)(acc_q[0] >> 32) | ((unsigned)(acc_q[1] >> 32) << 8) | ((unsigned)(acc_q[2] >> 32) << 16) |
                                           ((unsigned)(acc_q[3] >> 32) << 24);
            }
            }
        }
        __syncthreads();                                   // wrow_* and carry_* consumed
        if (tid < kIntegralBand) {                          // this chunk's row totals join the carries (images wider than one chunk)
            unsigned ts_ = 0, tq_ = 0;
            for (int j = 0; j < kIntWaves; j++) { ts_ += wrow_s[tid][j]; tq_ += wrow_q[tid][j]; }
            carry_s[tid] += ts_; carry_q[tid] += tq_;
        }
        __syncthreads();
    }
}
void launch_integral(hipStream_t st, const uint8_t *gray, const uint8_t *lut, int lut_stride, const PreGeom &g,
                     const unsigned *bandsum, const unsigned *bandsq, int *sum, unsigned long long *sqsum,
                     int batch)
{
    NVCA_LAUNCH(k_integral, dim3(g.nbands, batch), dim3(kIntThreads), 0, st, gray, lut, lut_stride, g, bandsum,
                       bandsq, sum, (unsigned *)sqsum);
}

// ---- CV_HAAR_SCALE_IMAGE pyramids: every level of every image in one launch per step -----------------------------
// The levels of a pyramid are small (the part detectors work at 320 pixels width) and there are ~15 of them: resizing
// and integrating them one level at a time is a chain of ~60 tiny dependent launches.  k_pyr_resize writes all levels
// (grid.z = level x image); k_pyr_integral gives each (level, image) one workgroup that walks the rows with one column
// per thread: running column sums in registers, one workgroup-wide scan per row (sum i32, squared sum u64 -> two u32
// planes, same layout as k_integral).  Levels wider than 1024 pixels take the general three-kernel path.
__global__ __launch_bounds__(256) void k_pyr_resize(const uint8_t *__restrict__ src, int sw, int sh, int sstride, size_t src_slot,
                                                    const PyrLevelDev *__restrict__ levels, int nimg,
                                                    uint8_t *__restrict__ aux, size_t aux_slot)
{
    const int lev = blockIdx.z / nimg, img = blockIdx.z - lev * nimg;
    const PyrLevelDev L = levels[lev];
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= L.szw || blockIdx.y * kGrayRows >= L.szh) return;
    const uint8_t *s = src + (size_t)img * src_slot;
    uint8_t *d = aux + (size_t)img * aux_slot + L.gray_off;
    for (int ry = 0; ry < kGrayRows; ry++) {
        const int y = blockIdx.y * kGrayRows + ry;
        if (y >= L.szh) break;
        if (x < L.szw) d[(size_t)y * L.gpitch + x] = (uint8_t)resize1_value(s, sh, sstride, L.mode, L.xofs, L.ialpha, L.yofs, L.ibeta, L.xmax, x, y);
    }
}

// Integral pair of a SMALL image (rows x (cols | 1) <= kSmallIntWords words of LDS) by one workgroup, without a barrier per
// row: (1) every wave scans whole rows -- 64 pixels per step, DPP wave scan, carry from chunk to chunk -- and leaves the row
// prefix sums in LDS; (2) after one barrier every thread owns a column and adds the rows up out of LDS, writing the
// integral rows to global memory fully coalesced.  Once for the pixel sums, once for their squares (row prefixes of squares
// stay below 2^32; the column sums are 64-bit and leave as the u32 low-word plane + u8 high-byte plane of k_integral).
// 160 x 90 (the part detectors' face-pass image): ~3 us instead of ~36 us for the row-by-row walk below.
static constexpr int kSmallIntWords = 16 * 1024 - 256;     // dynamic LDS: with the static scan words still inside the 64 KiB a kernel gets without asking
__device__ __forceinline__ void small_integral(const uint8_t *__restrict__ g, int gpitch, const uint8_t *__restrict__ lut, int w, int h,
                                               int *__restrict__ s, unsigned *__restrict__ lo, uint8_t *__restrict__ hi, int P, unsigned *sm)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int pitch = w | 1;
    for (int X = tid; X <= w; X += nthreads) { s[X] = 0; lo[X] = 0; hi[X] = 0; }            // integral row 0
    for (int pass = 0; pass < 2; pass++) {
        for (int y = wave; y < h; y += nwaves) {
            const uint8_t *row = g + (size_t)y * gpitch;
            unsigned carry = 0;
            for (int x0 = 0; x0 < w; x0 += 64) {
                const int x = x0 + lane;
                unsigned v = 0;
                if (x < w) { v = row[x]; if (lut) v = lut[v]; if (pass) v *= v; }
                const unsigned inc = wave_incl_scan_u32(v, lane) + carry;
                if (x < w) sm[y * pitch + x] = inc;
                carry = (unsigned)__builtin_amdgcn_readlane((int)inc, 63);
            }
        }
        __syncthreads();
        for (int x = tid; x < w; x += nthreads) {
            if (pass == 0) {
                unsigned acc = 0;
                for (int y = 0; y < h; y++) { acc += sm[y * pitch + x]; s[(size_t)(y + 1) * P + x + 1] = (int)acc; }
            } else {
                unsigned long long acc = 0;
                for (int y = 0; y < h; y++) {
                    acc += sm[y * pitch + x];
                    const size_t o = (size_t)(y + 1) * P + x + 1;
                    lo[o] = (unsigned)acc; hi[o] = (uint8_t)(acc >> 32);
                }
            }
        }
        if (pass == 0) for (int y = tid; y < h; y += nthreads) { const size_t o = (size_t)(y + 1) * P; s[o] = 0; lo[o] = 0; hi[o] = 0; }   // column 0
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_pyr_integral(const uint8_t *__restrict__ aux, size_t aux_slot,
                                                       const PyrLevelDev *__restrict__ levels, int nimg,
                                                       int *__restrict__ sum, unsigned *__restrict__ sq32, size_t sum_slot, int P)
{
    extern __shared__ unsigned pyr_sm[];
    __shared__ unsigned wt_s[2][16], wt_l[2][16], wt_h[2][16];
    const int lev = blockIdx.x / nimg, img = blockIdx.x - lev * nimg;
    const PyrLevelDev L = levels[lev];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = L.szw, h = L.szh;
    const uint8_t *g = aux + (size_t)img * aux_slot + L.gray_off;
    int *s = sum + (size_t)img * sum_slot + L.plane_off;
    unsigned *lo = sq32 + (size_t)img * 2 * sum_slot + L.plane_off;
    uint8_t *hi = (uint8_t *)(sq32 + (size_t)img * 2 * sum_slot + sum_slot) + L.plane_off;   // high-byte plane
    if (h * (w | 1) <= kSmallIntWords) { small_integral(g, L.gpitch, nullptr, w, h, s, lo, hi, P, pyr_sm); return; }
    if (tid <= w) { s[tid] = 0; lo[tid] = 0; hi[tid] = 0; }            // integral row 0
    // running sums of this thread's column; the squared one stays below 2^32 (rows x 255^2), so its row prefix can be
    // scanned as two 32-bit halves (low 16 bits / the rest) on the VALU and recombined in 64 bits
    unsigned cs = 0, cq = 0;
    unsigned pix = (tid < w && h > 0) ? g[tid] : 0;
    int par = 0;
    for (int y = 0; y < h; y++) {
        const unsigned cur = pix;
        if (tid < w && y + 1 < h) pix = g[(size_t)(y + 1) * L.gpitch + tid];      // next row in flight during the scan
        cs += cur; cq += cur * cur;
        unsigned is = wave_incl_scan_u32(cs, lane);
        unsigned il = wave_incl_scan_u32(cq & 0xffffu, lane), ih = wave_incl_scan_u32(cq >> 16, lane);
        if (lane == 63) { wt_s[par][wave] = is; wt_l[par][wave] = il; wt_h[par][wave] = ih; }
        __syncthreads();
        for (int j = 0; j < wave; j++) { is += wt_s[par][j]; il += wt_l[par][j]; ih += wt_h[par][j]; }
        par ^= 1;
        const unsigned long long iq = ((unsigned long long)ih << 16) + il;
        const size_t row = (size_t)(y + 1) * P;
        if (tid < w) { s[row + tid + 1] = (int)is; lo[row + tid + 1] = (unsigned)iq; hi[row + tid + 1] = (uint8_t)(iq >> 32); }
        if (tid == 0) { s[row] = 0; lo[row] = 0; hi[row] = 0; }
    }
}

// one small image per workgroup (batch slots), the planes laid out like k_integral's: the ROI-sized images of the part
// detectors' FIND_BIGGEST searches take one launch instead of column sums + band scan + row pass
__global__ __launch_bounds__(1024) void k_small_integral(const uint8_t *__restrict__ gray, const uint8_t *__restrict__ lut, int lut_stride, PreGeom g,
                                                         int *__restrict__ sum, unsigned *__restrict__ sq32)
{
    extern __shared__ unsigned pyr_sm[];
    const int slot = blockIdx.x;
    unsigned *lo = sq32 + (size_t)slot * 2 * g.sum_slot;
    small_integral(gray + (size_t)slot * g.gray_slot, g.gpitch, lut ? lut + (size_t)slot * lut_stride : nullptr, g.w, g.h,
                   sum + (size_t)slot * g.sum_slot, lo, (uint8_t *)(lo + g.sum_slot), g.spitch, pyr_sm);
}
bool small_integral_fits(const PreGeom &g) { return g.h * (g.w | 1) <= kSmallIntWords; }
void launch_small_integral(hipStream_t st, const uint8_t *gray, const uint8_t *lut, int lut_stride, const PreGeom &g, int *sum,
                           unsigned long long *sqsum, int batch)
{
    NVCA_LAUNCH(k_small_integral, dim3(batch), dim3(1024), (size_t)g.h * (g.w | 1) * sizeof(unsigned), st, gray, lut, lut_stride, g, sum, (unsigned *)sqsum);
}

// ---- tilted integral: cv::integral's third plane, read by tilted Haar features ------------------------------------------
// tilted(X,Y) = sum of image(x,y) over y < Y, abs(x - X + 1) <= Y - y - 1  (int32, (h+1) x (w+1), row 0 zero).
// Row Y of the triangle under (X,Y) differs from row Y-1's by the apex pixel (X-1,Y-1) and by its two end pixels per
// earlier row, which lie on the diagonals through (X-2,Y-2) and (X,Y-2): dl[c - y] / dr[c + y] are running sums along the two
// diagonal families.  One workgroup per image walks the rows; thread X reads its two diagonal sums (as of row Y-2), emits
// tilted(X,Y), then adds pixel (X-1,Y-1) to exactly those two sums -- the element a thread reads in a row is the one it
// updates, so one barrier per row orders everything.  Serial in the rows (cheap: the path is only taken for cascades that
// hold tilted features, on the small working images of the part detectors); exact 32-bit integer arithmetic.
static constexpr int kTiltedCols = 8;                 // columns per thread: w + 1 <= 8 * 1024
__device__ __forceinline__ void tilted_image(const uint8_t *__restrict__ g, int gpitch, const uint8_t *__restrict__ lut, int w, int h,
                                             int *__restrict__ out, int opitch, int *dl, int *dr)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < w + h + 2; i += 1024) { dl[i] = 0; dr[i] = 0; }
    for (int X = tid; X <= w; X += 1024) out[X] = 0;
    int prev[kTiltedCols];
#pragma unroll
    for (int k = 0; k < kTiltedCols; k++) prev[k] = 0;
    __syncthreads();
    for (int Y = 1; Y <= h; Y++) {
        const uint8_t *row = g + (size_t)(Y - 1) * gpitch;
#pragma unroll
        for (int k = 0; k < kTiltedCols; k++) {
            const int X = tid + 1024 * k;
            if (X > w) break;
            int p = 0;
            if (X >= 1) { p = row[X - 1]; if (lut) p = lut[p]; }
            int v = prev[k] + p;
            const int il = X - Y + h, ir = X + Y - 2;         // dl index of diagonal c - y = X - Y (shifted by h), dr index of c + y
            if (Y >= 2) {
                if (X >= 2) v += dl[il];
                if (X < w) v += dr[ir];
            }
            if (X >= 1) { dl[il] += p; dr[ir] += p; }
            out[(size_t)Y * opitch + X] = v;
            prev[k] = v;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_tilted(const uint8_t *__restrict__ gray, const uint8_t *__restrict__ lut, int lut_stride, PreGeom g,
                                                 int *__restrict__ tilted)
{
    extern __shared__ int tl_lds[];
    const int slot = blockIdx.x;
    tilted_image(gray + (size_t)slot * g.gray_slot, g.gpitch, lut ? lut + (size_t)slot * lut_stride : nullptr, g.w, g.h,
                 tilted + (size_t)slot * g.sum_slot, g.spitch, tl_lds, tl_lds + g.w + g.h + 2);
}

__global__ __launch_bounds__(1024) void k_pyr_tilted(const uint8_t *__restrict__ aux, size_t aux_slot, const PyrLevelDev *__restrict__ levels,
                                                     int nimg, int *__restrict__ tilted, size_t sum_slot, int P, int lds_half)
{
    extern __shared__ int tl_lds[];
    const int lev = blockIdx.x / nimg, img = blockIdx.x - lev * nimg;
    const PyrLevelDev L = levels[lev];
    tilted_image(aux + (size_t)img * aux_slot + L.gray_off, L.gpitch, nullptr, L.szw, L.szh,
                 tilted + (size_t)img * sum_slot + L.plane_off, P, tl_lds, tl_lds + lds_half);
}

void launch_tilted(hipStream_t st, const uint8_t *gray, const uint8_t *lut, int lut_stride, const PreGeom &g, int *tilted, int batch)
{
    NVCA_LAUNCH(k_tilted, dim3(batch), dim3(1024), (size_t)2 * (g.w + g.h + 2) * sizeof(int), st, gray, lut, lut_stride, g, tilted);
}
void launch_pyr_tilted(hipStream_t st, const uint8_t *aux, size_t aux_slot, const PyrLevelDev *levels, int nlev, int nimg,
                       int *tilted, size_t sum_slot, int P, int maxw, int maxh)
{
    const int half = maxw + maxh + 2;
    NVCA_LAUNCH(k_pyr_tilted, dim3(nlev * nimg), dim3(1024), (size_t)2 * half * sizeof(int), st, aux, aux_slot, levels, nimg, tilted, sum_slot, P, half);
}

void launch_pyr_resize(hipStream_t st, const uint8_t *src, int sw, int sh, int sstride, size_t src_slot, const PyrLevelDev *levels,
                       int nlev, int nimg, int maxw, int maxh, uint8_t *aux, size_t aux_slot)
{
    dim3 grid((maxw + 255) / 256, (maxh + kGrayRows - 1) / kGrayRows, nlev * nimg);
    NVCA_LAUNCH(k_pyr_resize, grid, dim3(256), 0, st, src, sw, sh, sstride, src_slot, levels, nimg, aux, aux_slot);
}
void launch_pyr_integral(hipStream_t st, const uint8_t *aux, size_t aux_slot, const PyrLevelDev *levels, int nlev, int nimg,
                         int *sum, unsigned *sq32, size_t sum_slot, int P)
{
    // dynamic LDS: the row prefix sums of the largest level that takes the LDS-resident path (64 KiB at most)
    NVCA_LAUNCH(k_pyr_integral, dim3(nlev * nimg), dim3(1024), (size_t)kSmallIntWords * sizeof(unsigned), st, aux, aux_slot, levels, nimg, sum, sq32, sum_slot, P);
}

// ---- view-* outlines on a device frame: a thread per pixel of the shapes' common bounding box; the last shape of the list
// that covers the pixel colours it (= the shapes drawn one after the other)
__global__ __launch_bounds__(256) void k_draw_shapes(uint8_t *__restrict__ data, int w, int h, int stride, int channels,
                                                     const nvca_shape *__restrict__ shapes, int n, int bx0, int by0, int bx1, int by1)
{
    extern __shared__ nvca_shape sh_s[];
    for (int i = threadIdx.x; i < n; i += 256) sh_s[i] = shapes[i];
    __syncthreads();
    const int x = bx0 + blockIdx.x * 64 + (threadIdx.x & 63), y = by0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x > bx1 || y > by1 || x >= w || y >= h) return;
    for (int i = n - 1; i >= 0; i--)
        if (shape_covers(sh_s[i], x, y)) {
            uint8_t *p = data + (size_t)y * stride + (size_t)x * channels;
            for (int k = 0; k < channels; k++) p[k] = sh_s[i].bgra[k];
            return;
        }
}
void launch_draw_shapes(hipStream_t st, uint8_t *data, int w, int h, int stride, int channels, const nvca_shape *d_shapes, int n,
                        int bx0, int by0, int bx1, int by1)
{
    dim3 grid((bx1 - bx0 + 64) / 64, (by1 - by0 + 4) / 4, 1);
    NVCA_LAUNCH(k_draw_shapes, grid, dim3(256), (size_t)n * sizeof(nvca_shape), st, data, w, h, stride, channels, d_shapes, n, bx0, by0, bx1, by1);
}

} // namespace nvca
