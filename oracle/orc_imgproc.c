/*
 * orc_imgproc.c -- oracle (test infrastructure, see nvca_oracle.h): CPU
 * restatement of the OpenCV-2.4 imgproc calls made by the reference at
 *   FACE/kmsfacedetect.cpp:805-807   (resize, cvtColor, equalizeHist)
 *   EYE/kmseyedetect.cpp:949-964, NOSE/kmsnosedetect.cpp:834-851,
 *   MOUTH/kmsmouthdetect.cpp:836-853, EAR/kmseardetect.cpp:786-800,
 *   TRK/gstnubotracker.cpp:356.
 * OpenCV itself is a third-party dependency absent from /root/reference
 * (opencv>=2.0.0, API forces 2.4.x); algorithms restated from OpenCV 2.4.8
 * modules/imgproc/src/{color,imgwarp,histogram,sumpixels}.cpp
 * (SURVEY.md Appendix A.1-A.4).  PARITY UNPINNED (no reference fixtures).
 */
#include "nvca_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_floor(double v) { return (int)floor(v); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* A.1: color.cpp RGB2Gray<uchar>: tab-driven, yuv_shift 14, B2Y 1868, G2Y 9617,
 * R2Y 4899, rounding term 1<<13 folded into the R table. */
void orc_bgr2gray(const uint8_t *src, int w, int h, int sstride, int cn,
                  uint8_t *dst, int dstride)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * sstride;
        uint8_t *d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++, s += cn)
            d[x] = (uint8_t)((s[0] * 1868 + s[1] * 9617 + s[2] * 4899 + 8192) >> 14);
    }
}

/* A.2: imgwarp.cpp cv::resize, INTER_LINEAR, depth 8U (fixed point, 11-bit
 * coefficients), including the "scale exactly 2 -> INTER_AREA fast" switch. */
void orc_resize_linear(const uint8_t *src, int sw, int sh, int sstride, int cn,
                       uint8_t *dst, int dw, int dh, int dstride)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int iscale_x = cv_round(scale_x), iscale_y = cv_round(scale_y); /* saturate_cast<int>(double) */
    int is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON &&
                       fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
        /* ResizeAreaFastVec<uchar>: (a+b+c+d+2)>>2 ; dw == sw/2 exactly here */
        for (int dy = 0; dy < dh; dy++) {
            const uint8_t *S = src + (size_t)(2 * dy) * sstride;
            const uint8_t *N = S + sstride;
            uint8_t *D = dst + (size_t)dy * dstride;
            for (int dx = 0; dx < dw; dx++)
                for (int k = 0; k < cn; k++) {
                    int idx = 2 * dx * cn + k;
                    D[dx * cn + k] = (uint8_t)((S[idx] + S[idx + cn] + N[idx] + N[idx + cn] + 2) >> 2);
                }
        }
        return;
    }

    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
    int *yofs = (int *)malloc(sizeof(int) * dh);
    short *ibeta = (short *)malloc(sizeof(short) * 2 * dh);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        ialpha[2 * dx]     = (short)clampi(cv_round(c0 * 2048), -32768, 32767);
        ialpha[2 * dx + 1] = (short)clampi(cv_round(c1 * 2048), -32768, 32767);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        float c0 = 1.f - fy, c1 = fy;
        ibeta[2 * dy]     = (short)clampi(cv_round(c0 * 2048), -32768, 32767);
        ibeta[2 * dy + 1] = (short)clampi(cv_round(c1 * 2048), -32768, 32767);
    }

    int *row0 = (int *)malloc(sizeof(int) * dw * cn);
    int *row1 = (int *)malloc(sizeof(int) * dw * cn);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy];
        int *rows[2] = { row0, row1 };
        for (int k = 0; k < 2; k++) {
            int sy = sy0 + k; /* clip(sy0 - ksize2 + 1 + k, 0, sh), ksize2 = 1 */
            sy = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0;
            const uint8_t *S = src + (size_t)sy * sstride;
            int *D = rows[k];
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx] * cn;
                if (dx < xmax) {
                    for (int c = 0; c < cn; c++)
                        D[dx * cn + c] = S[sx + c] * ialpha[2 * dx] + S[sx + cn + c] * ialpha[2 * dx + 1];
                } else {
                    for (int c = 0; c < cn; c++)
                        D[dx * cn + c] = S[sx + c] * 2048;
                }
            }
        }
        int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw * cn; x++)
            D[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(yofs); free(ibeta); free(row0); free(row1);
}

/* A.3: histogram.cpp cv::equalizeHist (2.4.3+). */
int orc_equalize_lut(const int hist[256], int total, uint8_t lut[256])
{
    int i = 0;
    while (i < 256 && !hist[i]) ++i;
    if (i == 256) { memset(lut, 0, 256); return 1; }
    if (hist[i] == total) { memset(lut, i, 256); return 1; }
    float scale = (256 - 1.f) / (total - hist[i]);
    int sum = 0;
    memset(lut, 0, 256);
    for (lut[i++] = 0; i < 256; ++i) {
        sum += hist[i];
        float v = sum * scale;               /* int*float -> float */
        lut[i] = (uint8_t)clampi(cv_round(v), 0, 255);
    }
    return 0;
}

void orc_equalize_hist(const uint8_t *src, int w, int h, int sstride,
                       uint8_t *dst, int dstride)
{
    int hist[256] = { 0 };
    uint8_t lut[256];
    if (w <= 0 || h <= 0) return;
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) hist[s[x]]++;
    }
    orc_equalize_lut(hist, w * h, lut);
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * sstride;
        uint8_t *d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) d[x] = lut[s[x]];
    }
}

/* A.4: sumpixels.cpp integral_<uchar,int,double>. */
void orc_integral(const uint8_t *src, int w, int h, int stride,
                  int32_t *sum, double *sqsum)
{
    int W1 = w + 1;
    for (int x = 0; x < W1; x++) { sum[x] = 0; if (sqsum) sqsum[x] = 0; }
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * stride;
        int32_t *srow = sum + (size_t)(y + 1) * W1, *sprev = srow - W1;
        double *qrow = sqsum ? sqsum + (size_t)(y + 1) * W1 : 0;
        double *qprev = qrow ? qrow - W1 : 0;
        int32_t rs = 0; double rq = 0;
        srow[0] = 0; if (qrow) qrow[0] = 0;
        for (int x = 0; x < w; x++) {
            int it = s[x];
            rs += it; rq += (double)it * it;
            srow[x + 1] = sprev[x + 1] + rs;
            if (qrow) qrow[x + 1] = qprev[x + 1] + rq;
        }
    }
}

/* The "tilted" (45 degree rotated) integral cv::integral also produces, as published for OpenCV 2.4
 * (imgproc, "integral"):  tilted(X,Y) = sum over y < Y, abs(x - X + 1) <= Y - y - 1 of image(x,y);
 * int32, (h+1)*(w+1) dense like sum.  cvHaarDetectObjectsForROC asks for it whenever the cascade has a tilted
 * feature (haar.cpp: cvIntegral(img, sum, sqsum, tilted)).  Restated from the definition: a row of the triangle
 * under (X,Y) differs from the row of the triangle under (X,Y-1) by its two end pixels, which lie on the two
 * diagonals through (X-2,Y-2) and (X,Y-2); dl / dr are the running sums along those diagonals. */
void orc_integral_tilted(const uint8_t *src, int w, int h, int stride, int32_t *tilted)
{
    int W1 = w + 1;
    int32_t *dl = (int32_t *)calloc((size_t)w * (h > 0 ? h : 1), sizeof(int32_t));   /* dl[y][c] = sum_k src[y-k][c-k] */
    int32_t *dr = (int32_t *)calloc((size_t)w * (h > 0 ? h : 1), sizeof(int32_t));   /* dr[y][c] = sum_k src[y-k][c+k] */
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * stride;
        for (int c = 0; c < w; c++) {
            dl[(size_t)y * w + c] = s[c] + (y > 0 && c > 0 ? dl[(size_t)(y - 1) * w + c - 1] : 0);
            dr[(size_t)y * w + c] = s[c] + (y > 0 && c + 1 < w ? dr[(size_t)(y - 1) * w + c + 1] : 0);
        }
    }
    for (int X = 0; X < W1; X++) tilted[X] = 0;
    for (int Y = 1; Y <= h; Y++) {
        const uint8_t *s = src + (size_t)(Y - 1) * stride;
        int32_t *trow = tilted + (size_t)Y * W1, *tprev = trow - W1;
        for (int X = 0; X < W1; X++) {
            int32_t v = tprev[X];
            if (X >= 1) v += s[X - 1];                                       /* the apex row: pixel (X-1, Y-1) */
            if (Y >= 2) {
                if (X >= 2) v += dl[(size_t)(Y - 2) * w + X - 2];            /* left ends  (X-1-k, Y-1-k), k >= 1 */
                if (X < w) v += dr[(size_t)(Y - 2) * w + X];                 /* right ends (X-1+k, Y-1-k), k >= 1 */
            }
            trow[X] = v;
        }
    }
    free(dl); free(dr);
}

void orc_flip_h(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * sstride;
        uint8_t *d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) d[x] = s[w - 1 - x];
    }
}
