/*
 * orc_pipe.c -- oracle (test infrastructure, see nvca_oracle.h): CPU
 * restatement of the reference's own per-frame glue
 *   kms_face_detect_conf_images / _process_frame / kms_face_send_event
 *       FACE/kmsfacedetect.cpp:282-306, 757-853, 179-249
 *   Faces::track_faces and helpers   FACE/Faces.cpp:78-188,
 *   BaseFace::calc_center            FACE/BaseFace.cpp:97-101
 *   gst_nubo_tracker_process, __join_objects, __merge, calc_dist
 *       TRK/gstnubotracker.cpp:119-200, 339-421
 * and of the OpenCV 2.4.8 video/motempl.cpp calls behind the tracker
 * (updateMotionHistory, segmentMotion -> cvFloodFill floating range,
 * SURVEY.md A.10-A.11).  PARITY UNPINNED (no reference fixtures).
 */
#include "nvca_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <limits.h>

static inline int cv_round(double v)
{
    /* _mm_cvtsd_si32: out-of-range / inf -> INT_MIN */
    if (!(v > -2147483648.5 && v < 2147483647.5)) return INT_MIN;
    return (int)lrint(v);
}

/* ======================= Faces (FACE/Faces.cpp) ========================== */
static int face_area(const orc_rect *r) { return r->w * r->h; }
static void face_center(const orc_rect *r, int *cx, int *cy)
{   /* BaseFace::calc_center FACE/BaseFace.cpp:97-101 */
    *cx = r->x + r->w / 2; *cy = r->y + r->h / 2;
}
static int calc_distance(const orc_rect *a, const orc_rect *b)
{   /* Faces::calc_distance FACE/Faces.cpp:183-188 */
    int ax, ay, bx, by;
    face_center(a, &ax, &ay); face_center(b, &bx, &by);
    double h2 = sqrt(pow((double)(bx - ax), 2) + pow((double)(by - ay), 2));
    return (int)h2;
}
static int get_distance_limit(int size1, int size2)
{   /* FACE/Faces.cpp:166-181 */
    int big = size1 > size2 ? size1 : size2;
    return big > 5000 ? 8 : (big > 2500 ? 5 : 3);
}
static int calc_diff_area_percentage(int size1, int size2)
{   /* FACE/Faces.cpp:160-164 */
    int diff = abs(size1 - size2);
    return (diff * 100) / size2;
}

#define AREA_PERCENTAGE 15

int orc_track_faces(orc_rect *faces, int *ids, int n_faces, int *next_id,
                    const orc_rect *cur, int n_cur, int track_threshold, int cap)
{   /* Faces::track_faces FACE/Faces.cpp:78-153.  cf = Faces(vector<Rect>) */
    orc_rect *cf = (orc_rect *)malloc(sizeof(orc_rect) * (n_cur > 0 ? n_cur : 1));
    orc_rect *nv = (orc_rect *)malloc(sizeof(orc_rect) * (n_faces + n_cur + 1));
    int *nid = (int *)malloc(sizeof(int) * (n_faces + n_cur + 1));
    int ncf = n_cur, nn = 0;
    memcpy(cf, cur, sizeof(orc_rect) * n_cur);
    for (int f = 0; f < n_faces; f++) {
        int t_distance = track_threshold, pos = -1;
        for (int i = 0; i < ncf; i++) {
            int d = calc_distance(&cf[i], &faces[f]);
            if (t_distance > d) { pos = i; t_distance = d; }
        }
        if (pos >= 0) {
            int d = calc_distance(&faces[f], &cf[pos]);
            if (get_distance_limit(face_area(&faces[f]), face_area(&cf[pos])) < d) {
                nv[nn] = cf[pos]; nid[nn++] = ids[f];
            } else if (AREA_PERCENTAGE < calc_diff_area_percentage(face_area(&faces[f]), face_area(&cf[pos]))) {
                orc_rect r = { faces[f].x, faces[f].y, cf[pos].w, cf[pos].h };
                nv[nn] = r; nid[nn++] = ids[f];
            } else {
                nv[nn] = faces[f]; nid[nn++] = ids[f];
            }
            memmove(&cf[pos], &cf[pos + 1], sizeof(orc_rect) * (ncf - pos - 1));
            ncf--;
        }
    }
    for (int i = 0; i < ncf; i++) { nv[nn] = cf[i]; nid[nn++] = (*next_id)++; }
    if (nn > cap) nn = cap;
    memcpy(faces, nv, sizeof(orc_rect) * nn);
    memcpy(ids, nid, sizeof(int) * nn);
    free(cf); free(nv); free(nid);
    return nn;
}

/* ================= NuboFaceDetector per-frame state machine ============== */
#define ORC_MAX_FACES 256
struct orc_face_stream {
    const orc_cascade *c;
    orc_face_params p;
    orc_rect faces[ORC_MAX_FACES]; int ids[ORC_MAX_FACES]; int n_faces; int next_id;
    int num_frame, num_iter, frames_with_no_detection, num_frames_to_process;
};

void orc_face_params_default(orc_face_params *p)
{
    p->width_to_process = 160; p->process_x_every_4 = 4; p->scale_factor_pct = 25;
    p->track_threshold = 40; p->euclidean_threshold = 8; p->area_threshold = 500;
    p->full_res = 0; p->min_neighbors = 3; p->policy = ORC_SUM_F32PAIR;
}

orc_face_stream *orc_face_stream_create(const orc_cascade *c, const orc_face_params *p)
{
    orc_face_stream *s = (orc_face_stream *)calloc(1, sizeof(*s));
    s->c = c; s->p = *p;
    return s;
}
void orc_face_stream_destroy(orc_face_stream *s) { free(s); }

#define GOP 4
#define MAX_NUM_FPS_WITH_NO_DETECTION 1

/* working-image geometry of one frame: conf_images FACE/kmsfacedetect.cpp:304 (INTEGER division, kept in a float) and
 * process_frame :770-783 */
static void face_geometry(const orc_face_params *p, int W, int H, int *cols, int *rows, int *norm_scale)
{
    float fscale = p->full_res ? 1.f : (float)(p->width_to_process ? W / p->width_to_process : 0);
    double scale = fscale;
    *norm_scale = p->full_res ? 1 : (p->width_to_process ? W / p->width_to_process : 0);
    *rows = H; *cols = W;
    if (cv_round(H / scale) > 0) *rows = cv_round(H / scale); else scale = 1;
    if (cv_round(W / scale) > 0) *cols = cv_round(W / scale); else scale = 1;
}

/* frame gating of kms_face_detect_process_frame (:794-803) with detect_event == 0 (__receive_event returns true,
 * :722-726): 1 if this frame is analysed.  Advances the frame counters. */
int orc_face_stream_gate(orc_face_stream *s)
{
    s->num_frame++; s->num_iter++;
    int px = s->p.process_x_every_4;
    if ((2 == px && (1 == s->num_frame % 2)) || ((2 != px) && (s->num_frame <= px))) {
        s->num_frames_to_process--;
        return 1;
    }
    return 0;
}

/* the stateless part of an analysed frame: resize -> gray -> equalizeHist -> detectMultiScale (:805-811) */
int orc_face_frame_detect(const orc_cascade *c, const orc_face_params *p, const uint8_t *bgr, int W, int H, int stride,
                          orc_rect *cur, int cap)
{
    int rows, cols, norm_scale;
    face_geometry(p, W, H, &cols, &rows, &norm_scale);
    uint8_t *aux = (uint8_t *)malloc((size_t)rows * cols * 3);
    uint8_t *gray = (uint8_t *)malloc((size_t)rows * cols);
    orc_resize_linear(bgr, W, H, stride, 3, aux, cols, rows, cols * 3);   /* :805 */
    orc_bgr2gray(aux, cols, rows, cols * 3, 3, gray, cols);                /* :806 */
    orc_equalize_hist(gray, cols, rows, cols, gray, cols);                 /* :807 */
    int n = orc_detect_multiscale(c, gray, cols, rows, cols, 1 + p->scale_factor_pct * 1.0 / 100, p->min_neighbors, 0,
                                  cols / 20, rows / 20, 0, 0, p->policy, cur, cap, NULL);
    free(aux); free(gray);
    return n;
}

/* the temporal part (:813-830) and the emission (kms_face_send_event :190,208-211); cur is read only if analysed */
int orc_face_stream_finish(orc_face_stream *s, int analysed, const orc_rect *cur, int n_cur, int W, int H,
                           orc_rect *out, int *ids, int cap)
{
    int rows, cols, norm_scale;
    face_geometry(&s->p, W, H, &cols, &rows, &norm_scale);
    if (analysed) {
        if (n_cur > 0) {
            s->n_faces = orc_track_faces(s->faces, s->ids, s->n_faces, &s->next_id, cur, n_cur,
                                         s->p.track_threshold, ORC_MAX_FACES);
        } else {
            if (s->frames_with_no_detection < MAX_NUM_FPS_WITH_NO_DETECTION)
                s->frames_with_no_detection += 1;
            else { s->frames_with_no_detection = 0; s->n_faces = 0; }
        }
    }
    if (GOP == s->num_frame) s->num_frame = 0;

    /* (guint) r->x * norm_scale */
    int n = s->n_faces < cap ? s->n_faces : cap;
    for (int i = 0; i < n; i++) {
        out[i].x = (int)((unsigned)s->faces[i].x * (unsigned)norm_scale);
        out[i].y = (int)((unsigned)s->faces[i].y * (unsigned)norm_scale);
        out[i].w = (int)((unsigned)s->faces[i].w * (unsigned)norm_scale);
        out[i].h = (int)((unsigned)s->faces[i].h * (unsigned)norm_scale);
        if (ids) ids[i] = s->ids[i];
    }
    return n;
}

int orc_face_stream_process(orc_face_stream *s, const uint8_t *bgr, int W, int H,
                            int stride, orc_rect *out, int *ids, int cap)
{
    orc_rect cur[ORC_MAX_FACES];
    int n = 0;
    const int analysed = orc_face_stream_gate(s);
    if (analysed) n = orc_face_frame_detect(s->c, &s->p, bgr, W, H, stride, cur, ORC_MAX_FACES);
    return orc_face_stream_finish(s, analysed, cur, n, W, H, out, ids, cap);
}

/* ============================ NuboTracker =============================== */
void orc_tracker_params_default(orc_tracker_params *p)
{   /* TRK/gstnubotracker.cpp:23-31 */
    p->threshold = 20; p->min_area = 50; p->max_area = 30000; p->distance = 35;
    p->mhi_duration = 0.2; p->seg_thresh = 32;
}

struct orc_tracker {
    orc_tracker_params p;
    int w, h, num_frames;
    uint8_t *prev; float *mhi;
};

orc_tracker *orc_tracker_create(const orc_tracker_params *p)
{
    orc_tracker *t = (orc_tracker *)calloc(1, sizeof(*t));
    t->p = *p;
    return t;
}
void orc_tracker_destroy(orc_tracker *t) { if (t) { free(t->prev); free(t->mhi); free(t); } }

/* cvUpdateMotionHistory (A.10) */
void orc_update_mhi(const uint8_t *silh, int w, int h, float *mhi, double timestamp, double dur)
{
    float ts = (float)timestamp;
    float delbound = (float)(timestamp - dur);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        float val = mhi[i];
        val = silh[i] ? ts : (val < delbound ? 0 : val);
        mhi[i] = val;
    }
}

/* cvSegmentMotion (A.11): floating-range 4-connected flood fill from every
 * unlabelled pixel equal to (float)timestamp, in raster order. */
int orc_segment_motion(float *mhi, int w, int h, double timestamp, double seg_thresh,
                       orc_rect *out, int cap)
{
    union { float f; int32_t i; } v;
    int32_t ts, stub;
    v.f = (float)timestamp; ts = v.i;
    v.f = FLT_MAX * 0.1f; stub = v.i;
    int32_t *mi = (int32_t *)mhi;
    size_t N = (size_t)w * h;
    for (size_t i = 0; i < N; i++) if (mi[i] == 0) mi[i] = stub;
    uint8_t *mask = (uint8_t *)calloc(N, 1);
    int *stack = (int *)malloc(sizeof(int) * (N > 0 ? N : 1));
    float lo = -(float)seg_thresh, up = (float)seg_thresh;
    int ncomp = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t idx = (size_t)y * w + x;
            if (mi[idx] != ts || mask[idx]) continue;
            int sp = 0, xmin = x, xmax = x, ymin = y, ymax = y;
            stack[sp++] = (int)idx; mask[idx] = 1;
            while (sp > 0) {
                int p = stack[--sp];
                int px = p % w, py = p / w;
                if (px < xmin) xmin = px;
                if (px > xmax) xmax = px;
                if (py < ymin) ymin = py;
                if (py > ymax) ymax = py;
                const int nx[4] = { px - 1, px + 1, px, px }, ny[4] = { py, py, py - 1, py + 1 };
                for (int k = 0; k < 4; k++) {
                    if (nx[k] < 0 || nx[k] >= w || ny[k] < 0 || ny[k] >= h) continue;
                    int q = ny[k] * w + nx[k];
                    if (mask[q]) continue;
                    float d = mhi[q] - mhi[p];          /* Diff32fC1: lo <= a-b <= up */
                    if (lo <= d && d <= up) { mask[q] = 1; stack[sp++] = q; }
                }
            }
            if (ncomp < cap) { orc_rect r = { xmin, ymin, xmax - xmin + 1, ymax - ymin + 1 }; out[ncomp] = r; }
            ncomp++;
        }
    for (size_t i = 0; i < N; i++) if (mi[i] == stub) mi[i] = 0;
    free(mask); free(stack);
    return ncomp < cap ? ncomp : cap;
}

static float trk_calc_dist(const orc_rect *a, const orc_rect *b)
{   /* TRK/gstnubotracker.cpp:119-129 */
    int c1x = a->x + a->w / 2, c1y = a->y + a->h / 2;
    int c2x = b->x + b->w / 2, c2y = b->y + b->h / 2;
    return (float)sqrt((double)((c1x - c2x) * (c1x - c2x) + (c1y - c2y) * (c1y - c2y)));
}
static int pt_inside(int px, int py, const orc_rect *r)
{   /* cv::Point::inside == Rect::contains */
    return r->x <= px && px < r->x + r->w && r->y <= py && py < r->y + r->h;
}
static orc_rect trk_merge(orc_rect r1, orc_rect r2)
{   /* TRK/gstnubotracker.cpp:131-169 */
    int tl1x = r1.x, tl1y = r1.y, br1x = r1.x + r1.w, br1y = r1.y + r1.h;
    int tl2x = r2.x, tl2y = r2.y, br2x = r2.x + r2.w, br2y = r2.y + r2.h;
    if (pt_inside(tl2x, tl2y, &r1) && pt_inside(br2x, br2y, &r1)) return r1;
    if (pt_inside(tl1x, tl1y, &r2) && pt_inside(br1x, br1y, &r2)) return r2;
    int tx = tl1x < tl2x ? tl1x : tl2x, ty = tl1y < tl2y ? tl1y : tl2y;
    int bx = br1x > br2x ? br1x : br2x, by = br1y > br2y ? br1y : br2y;
    /* cv::Rect(pt1, pt2) */
    orc_rect r;
    r.x = tx < bx ? tx : bx; r.y = ty < by ? ty : by;
    r.w = (tx > bx ? tx : bx) - r.x; r.h = (ty > by ? ty : by) - r.y;
    return r;
}

int orc_join_objects(orc_rect *sb, int n, int min_area, long max_area, int distance)
{   /* __join_objects TRK/gstnubotracker.cpp:171-200 */
    for (int a = n - 1; a >= 0; a--) {
        if (sb[a].w * sb[a].h > min_area && sb[a].w * sb[a].h < max_area) {
            for (int b = a - 1; b >= 0; b--) {
                if (sb[b].w * sb[b].h > min_area && sb[b].w * sb[b].h < max_area)
                    if ((float)distance > trk_calc_dist(&sb[a], &sb[b])) {
                        sb[b] = trk_merge(sb[a], sb[b]);
                        memmove(&sb[a], &sb[a + 1], sizeof(orc_rect) * (n - a - 1)); n--;
                        break;
                    }
            }
        } else {
            memmove(&sb[a], &sb[a + 1], sizeof(orc_rect) * (n - a - 1)); n--;
        }
    }
    return n;
}

int orc_tracker_process(orc_tracker *t, const uint8_t *bgra, int w, int h, int stride,
                        double timestamp_ms, orc_rect *out, int cap)
{   /* gst_nubo_tracker_img_conf :202-237 + gst_nubo_tracker_process :339-421 */
    size_t N = (size_t)w * h;
    if (t->w != w || t->h != h) {
        free(t->mhi); t->mhi = (float *)calloc(N, sizeof(float));
        /* deviation: img_prev is per stream (reference keeps one process-global
         * Mat, TRK/gstnubotracker.cpp:108) */
        if (!t->prev || t->w * t->h != w * h) { free(t->prev); t->prev = (uint8_t *)calloc(N, 1); }
        t->w = w; t->h = h;
    }
    uint8_t *gray = (uint8_t *)malloc(N);
    orc_bgr2gray(bgra, w, h, stride, 4, gray, w);
    int n = 0;
    if (t->num_frames > 0) {
        uint8_t *mask = (uint8_t *)malloc(N);
        for (size_t i = 0; i < N; i++) {
            int d = abs((int)gray[i] - (int)t->prev[i]);         /* absdiff  :361 */
            mask[i] = d > t->p.threshold ? 255 : 0;              /* threshold BINARY :364 */
        }
        orc_update_mhi(mask, w, h, t->mhi, timestamp_ms, t->p.mhi_duration);     /* :368 */
        /* calcMotionGradient :372 -- outputs never read; no observable effect */
        int cap2 = (int)(N < 1000000 ? N : 1000000);
        orc_rect *sb = (orc_rect *)malloc(sizeof(orc_rect) * (cap2 > 0 ? cap2 : 1));
        int ns = orc_segment_motion(t->mhi, w, h, timestamp_ms, t->p.seg_thresh, sb, cap2); /* :376 */
        ns = orc_join_objects(sb, ns, t->p.min_area, t->p.max_area, t->p.distance);         /* :380 */
        n = ns < cap ? ns : cap;
        memcpy(out, sb, sizeof(orc_rect) * n);
        free(sb); free(mask);
    }
    memcpy(t->prev, gray, N);                                     /* :415-419 */
    free(gray);
    t->num_frames++;
    return n;
}
