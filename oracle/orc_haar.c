/*
 * orc_haar.c -- oracle (test infrastructure, see nvca_oracle.h): CPU
 * restatement of cv::CascadeClassifier::detectMultiScale for old-format Haar
 * cascades as called by the reference at
 *   FACE/kmsfacedetect.cpp:809-811  (flags 0)
 *   EYE/kmseyedetect.cpp:958-960,991-993,1003-1005 (flags 0 / SCALE_IMAGE)
 *   NOSE/kmsnosedetect.cpp:843-846,870-873, MOUTH/kmsmouthdetect.cpp:845-848,
 *   870-873, EAR/kmseardetect.cpp:656-659,712-715 (SCALE_IMAGE / FIND_BIGGEST).
 * Restated from OpenCV 2.4.8 modules/objdetect/src/haar.cpp
 * (cvHaarDetectObjectsForROC, cvSetImagesForHaarClassifierCascade,
 * cvRunHaarClassifierCascadeSum) and cascadedetect.cpp (groupRectangles),
 * core operations.hpp (partition) -- SURVEY.md Appendix A.5-A.9.
 * OpenCV is absent from /root/reference: PARITY UNPINNED.
 */
#include "nvca_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

static inline int cv_round(double v) { return (int)lrint(v); }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

void orc_resize_linear(const uint8_t *, int, int, int, int, uint8_t *, int, int, int);

/* ---- hidden cascade (CvHidHaarClassifierCascade) ----------------------- */
typedef struct {
    int p0, p1, p2, p3;   /* offsets into sum (elements), relative to window origin */
    float weight;
    int used;
} hid_rect;
typedef struct {
    hid_rect r[3];
    float threshold;
    int left, right;
    int tilted;
} hid_node;
typedef struct {
    int first_cls, ncls;
    float threshold;      /* stage_threshold - 0.0001f */
    int two_rects;
} hid_stage;
typedef struct {
    const orc_cascade *c;
    int n_nodes;
    hid_node *nodes;
    int *cls_first_node;  /* [n_cls] */
    int *cls_first_alpha; /* [n_cls] */
    hid_stage *stages;
    int is_stump_based;
    /* set by set_images */
    const int32_t *sum; const double *sqsum; int step; /* step in elements */
    const int32_t *tilted;    /* tilted integral (same shape as sum) or NULL: no tilted feature in the cascade */
    int sum_w, sum_h;     /* sum.cols, sum.rows (= img+1) */
    double inv_window_area;
    int ep0, ep1, ep2, ep3;    /* equRect corner offsets */
    int real_w, real_h;
    int policy;
    orc_stats *stats;
} hid_cascade;

static hid_cascade *hid_create(const orc_cascade *c, int policy, orc_stats *stats)
{
    hid_cascade *h = (hid_cascade *)calloc(1, sizeof(*h));
    h->c = c; h->policy = policy; h->stats = stats;
    h->n_nodes = c->n_nodes;
    h->nodes = (hid_node *)calloc(c->n_nodes, sizeof(hid_node));
    h->cls_first_node = (int *)malloc(sizeof(int) * (c->n_cls + 1));
    h->cls_first_alpha = (int *)malloc(sizeof(int) * (c->n_cls + 1));
    h->stages = (hid_stage *)calloc(c->n_stages, sizeof(hid_stage));
    int node = 0, alpha = 0;
    h->is_stump_based = 1;
    for (int i = 0; i < c->n_cls; i++) {
        h->cls_first_node[i] = node; h->cls_first_alpha[i] = alpha;
        if (c->cls_nnodes[i] != 1) h->is_stump_based = 0;
        node += c->cls_nnodes[i]; alpha += c->cls_nnodes[i] + 1;
    }
    int cls = 0;
    for (int s = 0; s < c->n_stages; s++) {
        hid_stage *st = &h->stages[s];
        st->first_cls = cls; st->ncls = c->stage_ncls[s];
        /* icvCreateHidHaarClassifierCascade: threshold - icv_stage_threshold_bias (float) */
        st->threshold = c->stage_thr[s] - 0.0001f;
        st->two_rects = 1;
        for (int j = 0; j < st->ncls; j++, cls++)
            for (int l = 0; l < c->cls_nnodes[cls]; l++) {
                int n = h->cls_first_node[cls] + l;
                hid_node *hn = &h->nodes[n];
                hn->threshold = c->node_thr[n];
                hn->left = c->left[n]; hn->right = c->right[n];
                hn->tilted = c->tilted[n];
                const int *r2 = &c->rects[(n * 3 + 2) * 4];
                hn->r[0].used = hn->r[1].used = 1;
                if (fabs(c->rweights[n * 3 + 2]) < DBL_EPSILON || r2[2] == 0 || r2[3] == 0)
                    hn->r[2].used = 0;
                else { hn->r[2].used = 1; st->two_rects = 0; }
            }
    }
    return h;
}

static void hid_free(hid_cascade *h)
{
    free(h->nodes); free(h->cls_first_node); free(h->cls_first_alpha); free(h->stages); free(h);
}

/* cvSetImagesForHaarClassifierCascade (A.6).  An upright rectangle (x,y,w,h) reads the four corners of sum; a
 * tilted one (the rectangle rotated by 45 degrees about its top corner (x,y), w along the down-right and h along the
 * down-left diagonal) reads the tilted integral at p0 = (y, x), p1 = (y+h, x-h), p2 = (y+w, x+w),
 * p3 = (y+w+h, x+w-h) and its weight is halved (correction_ratio 0.5: a tilted w x h rectangle covers 2wh pixels). */
static void hid_set_images(hid_cascade *h, const int32_t *sum, const double *sqsum, const int32_t *tilted,
                           int sum_w, int sum_h, double scale)
{
    const orc_cascade *c = h->c;
    h->sum = sum; h->sqsum = sqsum; h->tilted = tilted; h->step = sum_w; h->sum_w = sum_w; h->sum_h = sum_h;
    h->real_w = cv_round(c->ow * scale);
    h->real_h = cv_round(c->oh * scale);
    int ex = cv_round(scale), ey = ex;
    int ew = cv_round((c->ow - 2) * scale), eh = cv_round((c->oh - 2) * scale);
    double weight_scale = 1. / (ew * eh);
    h->inv_window_area = weight_scale;
    h->ep0 = ey * sum_w + ex;          h->ep1 = ey * sum_w + ex + ew;
    h->ep2 = (ey + eh) * sum_w + ex;   h->ep3 = (ey + eh) * sum_w + ex + ew;
    for (int n = 0; n < c->n_nodes; n++) {
        hid_node *hn = &h->nodes[n];
        double sum0 = 0, area0 = 0;
        int nr = hn->r[2].used ? 3 : 2;
        for (int k = 0; k < nr; k++) {
            const int *r = &c->rects[(n * 3 + k) * 4];
            int tx = cv_round(r[0] * scale), tw = cv_round(r[2] * scale);
            int ty = cv_round(r[1] * scale), th = cv_round(r[3] * scale);
            double correction_ratio = weight_scale * (!hn->tilted ? 1 : 0.5);
            if (!hn->tilted) {
                hn->r[k].p0 = ty * sum_w + tx;          hn->r[k].p1 = ty * sum_w + tx + tw;
                hn->r[k].p2 = (ty + th) * sum_w + tx;   hn->r[k].p3 = (ty + th) * sum_w + tx + tw;
            } else {
                hn->r[k].p0 = ty * sum_w + tx;                 hn->r[k].p1 = (ty + th) * sum_w + tx - th;
                hn->r[k].p2 = (ty + tw) * sum_w + tx + tw;     hn->r[k].p3 = (ty + tw + th) * sum_w + tx + tw - th;
            }
            hn->r[k].weight = (float)(c->rweights[n * 3 + k] * correction_ratio);
            if (k == 0)
                area0 = tw * th;
            else {
                float t = hn->r[k].weight * tw;   /* float*int -> float, left to right */
                t = t * th;
                sum0 += t;
            }
        }
        hn->r[0].weight = (float)(-sum0 / area0);
    }
}

#define CALC_SUM_P(P, R, off) ((P)[(R).p0 + (off)] - (P)[(R).p1 + (off)] - (P)[(R).p2 + (off)] + (P)[(R).p3 + (off)])
#define CALC_SUM(R, off) CALC_SUM_P(pl, R, off)

/* feature value of one node under the selected accumulation policy.
 * pair_f32: this node sits in a two_rects stage of a stump cascade evaluated
 * by the SSE2 code path. */
static inline double node_sum(const hid_cascade *h, const hid_node *n, int off, int pair_f32)
{
    const int32_t *pl = n->tilted ? h->tilted : h->sum;     /* the plane the feature's corner pointers refer to */
    if (pair_f32) {
        float s = CALC_SUM(n->r[0], off) * n->r[0].weight + CALC_SUM(n->r[1], off) * n->r[1].weight;
        return (double)s;
    }
    double s = CALC_SUM(n->r[0], off) * n->r[0].weight;   /* int*float -> float -> double */
    s += CALC_SUM(n->r[1], off) * n->r[1].weight;
    if (n->r[2].used) s += CALC_SUM(n->r[2], off) * n->r[2].weight;
    return s;
}

/* cvRunHaarClassifierCascadeSum(cascade, pt, stage_sum, start_stage=0) */
static int hid_run(hid_cascade *h, int x, int y)
{
    const orc_cascade *c = h->c;
    if (x < 0 || y < 0 || x + h->real_w >= h->sum_w || y + h->real_h >= h->sum_h)
        return -1;
    int off = y * h->step + x;
    if (h->stats) h->stats->windows++;
    double mean = (h->sum[h->ep0 + off] - h->sum[h->ep1 + off] - h->sum[h->ep2 + off] + h->sum[h->ep3 + off]) * h->inv_window_area;
    double vnf = h->sqsum[h->ep0 + off] - h->sqsum[h->ep1 + off] - h->sqsum[h->ep2 + off] + h->sqsum[h->ep3 + off];
    vnf = vnf * h->inv_window_area - mean * mean;
    if (vnf >= 0.) vnf = sqrt(vnf); else vnf = 1.;

    for (int i = 0; i < c->n_stages; i++) {
        const hid_stage *st = &h->stages[i];
        double stage_sum = 0.0;
        if (h->stats && i < 64) h->stats->stage_enter[i]++;
        if (h->is_stump_based) {
            int pair = (h->policy == ORC_SUM_F32PAIR) && st->two_rects;
            for (int j = 0; j < st->ncls; j++) {
                int cls = st->first_cls + j;
                const hid_node *n = &h->nodes[h->cls_first_node[cls]];
                const float *alpha = &c->alpha[h->cls_first_alpha[cls]];
                double t = n->threshold * vnf;
                double s = node_sum(h, n, off, pair);
                stage_sum += alpha[s >= t];
            }
            if (h->stats) h->stats->stumps += st->ncls;
        } else {
            for (int j = 0; j < st->ncls; j++) {
                int cls = st->first_cls + j;
                const hid_node *base = &h->nodes[h->cls_first_node[cls]];
                const float *alpha = &c->alpha[h->cls_first_alpha[cls]];
                int idx = 0;
                do {
                    const hid_node *n = base + idx;
                    double t = n->threshold * vnf;
                    double s = node_sum(h, n, off, 0);
                    idx = s < t ? n->left : n->right;
                    if (h->stats) h->stats->stumps++;
                } while (idx > 0);
                stage_sum += alpha[-idx];
            }
        }
        if (stage_sum < st->threshold) return -i;
    }
    return 1;
}

/* ---- groupRectangles (A.9) --------------------------------------------- */
static int similar_rects(const orc_rect *a, const orc_rect *b, double eps)
{
    double delta = eps * (imin(a->w, b->w) + imin(a->h, b->h)) * 0.5;
    return abs(a->x - b->x) <= delta && abs(a->y - b->y) <= delta &&
           abs(a->x + a->w - b->x - b->w) <= delta && abs(a->y + a->h - b->y - b->h) <= delta;
}

/* cv::partition with SimilarRects */
static int partition_rects(const orc_rect *v, int N, double eps, int *labels)
{
    int (*nodes)[2] = (int (*)[2])malloc(sizeof(int) * 2 * (N > 0 ? N : 1));
    for (int i = 0; i < N; i++) { nodes[i][0] = -1; nodes[i][1] = 0; }
    for (int i = 0; i < N; i++) {
        int root = i;
        while (nodes[root][0] >= 0) root = nodes[root][0];
        for (int j = 0; j < N; j++) {
            if (i == j || !similar_rects(&v[i], &v[j], eps)) continue;
            int root2 = j;
            while (nodes[root2][0] >= 0) root2 = nodes[root2][0];
            if (root2 != root) {
                int rank = nodes[root][1], rank2 = nodes[root2][1];
                if (rank > rank2) nodes[root2][0] = root;
                else {
                    nodes[root][0] = root2;
                    nodes[root2][1] += rank == rank2;
                    root = root2;
                }
                int k = j, parent;
                while ((parent = nodes[k][0]) >= 0) { nodes[k][0] = root; k = parent; }
                k = i;
                while ((parent = nodes[k][0]) >= 0) { nodes[k][0] = root; k = parent; }
            }
        }
    }
    int nclasses = 0;
    for (int i = 0; i < N; i++) {
        int root = i;
        while (nodes[root][0] >= 0) root = nodes[root][0];
        if (nodes[root][1] >= 0) nodes[root][1] = ~nclasses++;
        labels[i] = ~nodes[root][1];
    }
    free(nodes);
    return nclasses;
}

int orc_group_rectangles(orc_rect *rects, int n, int group_threshold, double eps, int *weights)
{
    if (group_threshold <= 0 || n == 0) {
        if (weights) for (int i = 0; i < n; i++) weights[i] = 1;
        return n;
    }
    int *labels = (int *)malloc(sizeof(int) * n);
    int nclasses = partition_rects(rects, n, eps, labels);
    orc_rect *rr = (orc_rect *)calloc(nclasses, sizeof(orc_rect));
    int *rw = (int *)calloc(nclasses, sizeof(int));
    for (int i = 0; i < n; i++) {
        int cls = labels[i];
        rr[cls].x += rects[i].x; rr[cls].y += rects[i].y;
        rr[cls].w += rects[i].w; rr[cls].h += rects[i].h;
        rw[cls]++;
    }
    for (int i = 0; i < nclasses; i++) {
        float s = 1.f / rw[i];
        orc_rect r = rr[i];
        rr[i].x = cv_round(r.x * s); rr[i].y = cv_round(r.y * s);   /* int*float -> float */
        rr[i].w = cv_round(r.w * s); rr[i].h = cv_round(r.h * s);
    }
    int nout = 0;
    for (int i = 0; i < nclasses; i++) {
        orc_rect r1 = rr[i];
        int n1 = rw[i], j;
        if (n1 <= group_threshold) continue;
        for (j = 0; j < nclasses; j++) {
            int n2 = rw[j];
            if (j == i || n2 <= group_threshold) continue;
            orc_rect r2 = rr[j];
            int dx = cv_round(r2.w * eps), dy = cv_round(r2.h * eps);
            if (r1.x >= r2.x - dx && r1.y >= r2.y - dy &&
                r1.x + r1.w <= r2.x + r2.w + dx && r1.y + r1.h <= r2.y + r2.h + dy &&
                (n2 > imax(3, n1) || n1 < 3))
                break;
        }
        if (j == nclasses) {
            rects[nout] = r1;
            if (weights) weights[nout] = n1;
            nout++;
        }
    }
    free(labels); free(rr); free(rw);
    return nout;
}

/* ---- detectMultiScale (A.5, A.7, A.8) ---------------------------------- */
typedef struct { orc_rect *v; int n, cap; } rvec;
static void rv_push(rvec *r, orc_rect x)
{
    if (r->n == r->cap) { r->cap = r->cap ? r->cap * 2 : 256; r->v = (orc_rect *)realloc(r->v, sizeof(orc_rect) * r->cap); }
    r->v[r->n++] = x;
}

#define GROUP_EPS 0.2

static int detect_impl(const orc_cascade *c, const uint8_t *img, int cols, int rows, int stride,
                       double scaleFactor, int minNeighbors, int flags, int minw, int minh,
                       int maxw, int maxh, int policy, int raw_only, orc_rect *out, int cap,
                       orc_stats *stats, double *grid, int grid_cap, int *grid_n)
{
    rvec all = { 0, 0, 0 };
    double factor;
    int findBiggest = (flags & ORC_HAAR_FIND_BIGGEST_OBJECT) != 0;
    int roughSearch = (flags & ORC_HAAR_DO_ROUGH_SEARCH) != 0;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (maxh == 0 || maxw == 0) { maxh = rows; maxw = cols; }
    if (findBiggest) flags &= ~(ORC_HAAR_SCALE_IMAGE | ORC_HAAR_DO_CANNY_PRUNING);
    /* DO_CANNY_PRUNING is never set by the reference; not restated. */

    hid_cascade *h = (grid ? NULL : hid_create(c, policy, stats));
    int32_t *sum = NULL, *tilted = NULL; double *sqsum = NULL;
    if (!grid) {
        sum = (int32_t *)malloc(sizeof(int32_t) * (size_t)(rows + 1) * (cols + 1));
        sqsum = (double *)malloc(sizeof(double) * (size_t)(rows + 1) * (cols + 1));
        int has_tilted = 0;                  /* hid_cascade->has_tilted_features: the third plane of cvIntegral */
        for (int n = 0; n < c->n_nodes; n++) has_tilted |= c->tilted[n] != 0;
        if (has_tilted) tilted = (int32_t *)malloc(sizeof(int32_t) * (size_t)(rows + 1) * (cols + 1));
    }

    if (flags & ORC_HAAR_SCALE_IMAGE) {
        uint8_t *small = grid ? NULL : (uint8_t *)malloc((size_t)(rows + 1) * (cols + 1));
        for (factor = 1;; factor *= scaleFactor) {
            int winw = cv_round(c->ow * factor), winh = cv_round(c->oh * factor);
            int szw = cv_round(cols / factor), szh = cv_round(rows / factor);
            int sz1w = szw - c->ow + 1, sz1h = szh - c->oh + 1;
            if (sz1w <= 0 || sz1h <= 0) break;
            if (winw > maxw || winh > maxh) break;
            if (winw < minw || winh < minh) continue;
            if (grid) { if (*grid_n < grid_cap) grid[*grid_n] = factor; (*grid_n)++; continue; }
            if (stats) stats->n_scales++;
            orc_resize_linear(img, cols, rows, stride, 1, small, szw, szh, szw);
            orc_integral(small, szw, szh, szw, sum, sqsum);
            if (tilted) orc_integral_tilted(small, szw, szh, szw, tilted);
            hid_set_images(h, sum, sqsum, tilted, szw + 1, szh + 1, 1.);
            int ystep = factor > 2 ? 1 : 2;
            /* strips only partition rows into multiples of ystep: same set of y */
            int y2 = (szh + 1) - 1 - c->oh;
            int ssw = (szw + 1) - 1 - c->ow;
            if ((szw + 1) <= 1 + c->ow) continue;
            for (int y = 0; y < y2; y += ystep)
                for (int x = 0; x < ssw; x += ystep)
                    if (hid_run(h, x, y) > 0) {
                        orc_rect r = { cv_round(x * factor), cv_round(y * factor), winw, winh };
                        rv_push(&all, r);
                    }
        }
        free(small);
    } else {
        int n_factors = 0;
        orc_rect scanROI = { 0, 0, 0, 0 };
        if (!grid) { orc_integral(img, cols, rows, stride, sum, sqsum); if (tilted) orc_integral_tilted(img, cols, rows, stride, tilted); }
        for (n_factors = 0, factor = 1;
             factor * c->ow < cols - 10 && factor * c->oh < rows - 10;
             n_factors++, factor *= scaleFactor)
            ;
        if (findBiggest) { scaleFactor = 1. / scaleFactor; factor *= scaleFactor; }
        else factor = 1;
        for (; n_factors-- > 0; factor *= scaleFactor) {
            const double ystep = factor > 2. ? factor : 2.;
            int winw = cv_round(c->ow * factor), winh = cv_round(c->oh * factor);
            int startX = 0, startY = 0;
            int endX = cv_round((cols - winw) / ystep);
            int endY = cv_round((rows - winh) / ystep);
            if (winw < minw || winh < minh) { if (findBiggest) break; continue; }
            if (winw > maxw || winh > maxh) { if (!findBiggest) break; continue; }
            if (grid) { if (*grid_n < grid_cap) grid[*grid_n] = factor; (*grid_n)++; continue; }
            if (stats) stats->n_scales++;
            hid_set_images(h, sum, sqsum, tilted, cols + 1, rows + 1, factor);
            if (scanROI.w * scanROI.h > 0) {
                startY = cv_round(scanROI.y / ystep);
                endY = cv_round((scanROI.y + scanROI.h - winh) / ystep);
                startX = cv_round(scanROI.x / ystep);
                endX = cv_round((scanROI.x + scanROI.w - winw) / ystep);
            }
            for (int iy = startY; iy < endY; iy++) {
                int y = cv_round(iy * ystep), ixstep = 1;
                for (int ix = startX; ix < endX; ix += ixstep) {
                    int x = cv_round(ix * ystep);
                    int result = hid_run(h, x, y);
                    if (result > 0) { orc_rect r = { x, y, winw, winh }; rv_push(&all, r); }
                    ixstep = result != 0 ? 1 : 2;
                }
            }
            if (findBiggest && all.n > 0 && scanROI.w * scanROI.h == 0) {
                orc_rect *tmp = (orc_rect *)malloc(sizeof(orc_rect) * all.n);
                memcpy(tmp, all.v, sizeof(orc_rect) * all.n);
                int nt = orc_group_rectangles(tmp, all.n, imax(minNeighbors, 1), GROUP_EPS, NULL);
                if (nt > 0) {
                    orc_rect maxRect = { 0, 0, 0, 0 };
                    for (int i = 0; i < nt; i++)
                        if (tmp[i].w * tmp[i].h > maxRect.w * maxRect.h) maxRect = tmp[i];
                    rv_push(&all, maxRect);
                    scanROI = maxRect;
                    int dx = cv_round(maxRect.w * GROUP_EPS), dy = cv_round(maxRect.h * GROUP_EPS);
                    scanROI.x = imax(scanROI.x - dx, 0);
                    scanROI.y = imax(scanROI.y - dy, 0);
                    scanROI.w = imin(scanROI.w + dx * 2, cols - 1 - scanROI.x);
                    scanROI.h = imin(scanROI.h + dy * 2, rows - 1 - scanROI.y);
                    double minScale = roughSearch ? 0.6 : 0.4;
                    minw = cv_round(maxRect.w * minScale);
                    minh = cv_round(maxRect.h * minScale);
                }
                free(tmp);
            }
        }
    }
    if (grid) { free(all.v); return 0; }
    if (stats) stats->raw_hits = all.n;

    int n = all.n;
    if (!raw_only) {
        int *weights = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
        if (minNeighbors != 0 || findBiggest)
            n = orc_group_rectangles(all.v, n, imax(minNeighbors, 1), GROUP_EPS, weights);
        if (findBiggest && n > 0) {
            orc_rect best = { 0, 0, 0, 0 };
            for (int i = 0; i < n; i++)
                if (all.v[i].w * all.v[i].h > best.w * best.h) best = all.v[i];
            all.v[0] = best; n = 1;
        }
        free(weights);
    }
    int nout = n < cap ? n : cap;
    if (nout > 0) memcpy(out, all.v, sizeof(orc_rect) * nout);
    free(all.v); free(sum); free(sqsum); free(tilted); hid_free(h);
    return nout;
}

int orc_detect_multiscale(const orc_cascade *c, const uint8_t *gray, int w, int h, int stride,
                          double scale_factor, int min_neighbors, int flags, int min_w, int min_h,
                          int max_w, int max_h, int policy, orc_rect *out, int cap, orc_stats *stats)
{
    return detect_impl(c, gray, w, h, stride, scale_factor, min_neighbors, flags, min_w, min_h,
                       max_w, max_h, policy, 0, out, cap, stats, NULL, 0, NULL);
}

int orc_detect_raw(const orc_cascade *c, const uint8_t *gray, int w, int h, int stride,
                   double scale_factor, int flags, int min_w, int min_h, int max_w, int max_h,
                   int policy, orc_rect *out, int cap, orc_stats *stats)
{
    if (flags & ORC_HAAR_FIND_BIGGEST_OBJECT) return -1;
    return detect_impl(c, gray, w, h, stride, scale_factor, 0, flags, min_w, min_h, max_w, max_h,
                       policy, 1, out, cap, stats, NULL, 0, NULL);
}

int orc_scale_grid(int ow, int oh, int w, int h, double scale_factor, int min_w, int min_h,
                   int max_w, int max_h, double *factors, int cap)
{
    orc_cascade c; memset(&c, 0, sizeof(c)); c.ow = ow; c.oh = oh;
    int n = 0;
    detect_impl(&c, NULL, w, h, w, scale_factor, 0, 0, min_w, min_h, max_w, max_h, 0, 1,
                NULL, 0, NULL, factors, cap, &n);
    return n;
}
