/*
 * orc_parts.c -- oracle (test infrastructure, see nvca_oracle.h): CPU restatement of the
 * part detectors' per-frame glue
 *   kms_eye_detect_conf_images / _process_frame, __merge_eyes_current_frame,
 *   __merge_eyes_consecutives_frames, transform_2_global_coordinates
 *       EYE/kmseyedetect.cpp:310-341, 766-913, 915-1064
 *   kms_nose_detect_*   NOSE/kmsnosedetect.cpp:275-308, 700-743, 745-868
 *   kms_mouth_detect_*  MOUTH/kmsmouthdetect.cpp:285-315, 750-873
 *   kms_ear_detect_*    EAR/kmseardetect.cpp:292-319, 644-729, 767-812
 * on top of the OpenCV restatements in orc_imgproc.c / orc_haar.c.  PARITY UNPINNED.
 * std::vector idioms of the reference (erase through reverse iterators, erase(end()-i)) are
 * restated with the behaviour libstdc++ gives them.
 */
#include "nvca_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

static inline int cv_round(double v)
{
    if (!(v > -2147483648.5 && v < 2147483647.5)) return INT_MIN;
    return (int)lrint(v);
}

#define MAXR 256
typedef struct { orc_rect v[MAXR]; int n; } rlist;
static void rl_push(rlist *l, orc_rect r) { if (l->n < MAXR) l->v[l->n++] = r; }
static void rl_erase(rlist *l, int i) { memmove(&l->v[i], &l->v[i + 1], sizeof(orc_rect) * (l->n - i - 1)); l->n--; }

struct orc_part_stream {
    orc_part_params p;
    const orc_cascade *face, *a, *b;
    rlist faces, la, lb;                 /* faces; eyes_r/noses/mouths/lear ; eyes_l/rear */
    int num_frame, num_frames_to_process;
    int no_det_a, no_det_b;              /* frames_with_no_detection_er / _el ; ear: shared counter in no_det_a */
    rlist queue[16]; int qn;             /* pending upstream face messages */
};

void orc_part_params_default(orc_part_params *p, int kind)
{
    p->kind = kind; p->width_to_process = 320; p->process_x_every_4 = 4; p->scale_factor_pct = 25;
    p->detect_event = 0; p->policy = ORC_SUM_F32PAIR;
}
orc_part_stream *orc_part_stream_create(const orc_part_params *p, const orc_cascade *face, const orc_cascade *a,
                                        const orc_cascade *b)
{
    orc_part_stream *s = (orc_part_stream *)calloc(1, sizeof(*s));
    s->p = *p; s->face = face; s->a = a; s->b = b;
    return s;
}
void orc_part_stream_destroy(orc_part_stream *s) { free(s); }
void orc_part_stream_push_faces(orc_part_stream *s, const orc_rect *faces, int n)
{
    if (s->qn >= 16) return;
    rlist *q = &s->queue[s->qn++];
    q->n = 0;
    for (int i = 0; i < n; i++) rl_push(q, faces[i]);
}

/* detectMultiScale on a sub-rectangle of an image (cv::Mat ROI) */
static int detect_roi(const orc_cascade *c, const uint8_t *img, int cols, int rows, orc_rect roi, double sf, int mn,
                      int flags, int minw, int minh, int policy, orc_rect *out, int cap)
{
    if (roi.x < 0 || roi.y < 0 || roi.w <= 0 || roi.h <= 0 || roi.x + roi.w > cols || roi.y + roi.h > rows) return -1;
    return orc_detect_multiscale(c, img + (size_t)roi.y * cols + roi.x, roi.w, roi.h, cols, sf, mn, flags, minw, minh, 0, 0,
                                 policy, out, cap, NULL);
}

/* __merge_{noses,mouths}_consecutives_frames */
static void merge_consecutive_nm(rlist *cn, const rlist *old, orc_rect face, int scale, int dis, rlist *res)
{
    res->n = 0;
    for (int i = 0; i < old->n; i++) {
        const int ocx = old->v[i].x + old->v[i].w / 2, ocy = old->v[i].y + old->v[i].h / 2;
        for (int j = 0; j < cn->n; j++) {
            const int ncx = (cn->v[j].x + face.x) * scale + ((cn->v[j].w * scale) / 2);
            const int ncy = (cn->v[j].y + face.y) * scale + ((cn->v[j].h * scale) / 2);
            const double h2 = sqrt(pow((double)(ncx - ocx), 2) + pow((double)(ncy - ocy), 2));
            if (h2 < dis) { rl_push(res, old->v[i]); rl_erase(cn, j); break; }
        }
    }
    for (int j = 0; j < cn->n; j++) {
        orc_rect r = cn->v[j];
        r.x = cv_round((face.x + r.x) * scale); r.y = cv_round((face.y + r.y) * scale);
        r.w = (r.w - 1) * scale; r.h = (r.h - 1) * scale;
        rl_push(res, r);
    }
}

/* ---- eye helpers ------------------------------------------------------- */
static int contain_bb(int px, int py, orc_rect r)
{
    return (py >= r.y && py <= r.y + r.h) && (px >= r.x && px <= r.x + r.w);
}
static void merge_eyes_current(orc_rect face_bb, const rlist *eye_r, rlist *eyes, int scale, int eye_left)
{
    for (int i = eyes->n - 1; i > 0; i--) {
        int cx = eyes->v[i].x + eyes->v[i].w / 2, cy = eyes->v[i].y + eyes->v[i].h / 2;
        if (contain_bb(cx, cy, eyes->v[i - 1]) && eyes->v[i].w * eyes->v[i].h < eyes->v[i - 1].w * eyes->v[i - 1].h)
            rl_erase(eyes, eyes->n - i - 1);                       /* eyes.erase(eyes.end()-i-1) */
        else {
            cx = eyes->v[i - 1].x + eyes->v[i - 1].w / 2; cy = eyes->v[i - 1].y + eyes->v[i - 1].h / 2;
            if (contain_bb(cx, cy, eyes->v[i]) && eyes->v[i - 1].w * eyes->v[i - 1].h < eyes->v[i].w * eyes->v[i].h)
                rl_erase(eyes, eyes->n - i);                       /* eyes.erase(eyes.end()-i) */
        }
    }
    for (int i = eyes->n - 1; i >= 0; i--) {                       /* reverse-iterator loop with erase(--r.base()) */
        const int y_aux = face_bb.y * scale + face_bb.h * scale * 60 / 100;
        if (face_bb.y * scale + eyes->v[i].y < y_aux) {
            if (i == 0 && eyes->n == 1) {
                if (eye_r->n > 0 && eye_left) eyes->v[i].y = eye_r->v[0].y;
            } else
                rl_erase(eyes, i);
        }
    }
    if (eyes->n > 1) {
        const int middle_y = face_bb.x * scale + face_bb.h * scale / 2;      /* sic */
        const int middle_x = face_bb.y * scale + face_bb.w * scale / 2;
        for (int i = eyes->n - 1; i > 0; i--) {
            const int cy = eyes->v[i].y + eyes->v[i].h / 2, cx = eyes->v[i].x + eyes->v[i].w / 2;
            const int cy2 = eyes->v[i - 1].y + eyes->v[i - 1].h / 2, cx2 = eyes->v[i - 1].x + eyes->v[i - 1].w / 2;
            const float s1 = (float)sqrt(pow((double)(middle_x - cx), 2) + pow((double)(middle_y - cy), 2));
            const float s2 = (float)sqrt(pow((double)(middle_x - cx2), 2) + pow((double)(middle_y - cy2), 2));
            if (s1 < s2) rl_erase(eyes, eyes->n - i - 1); else rl_erase(eyes, eyes->n - i);
        }
    }
    if (eye_left && eye_r->n > 0 && eyes->n > 0) eyes->v[0].y = eye_r->v[0].y;
}
static void merge_eyes_consecutive(rlist *ce, const rlist *old, rlist *res)
{
    res->n = 0;
    for (int i = 0; i < old->n; i++) {
        const int ocx = old->v[i].x + old->v[i].w / 2, ocy = old->v[i].y + old->v[i].h / 2;
        for (int j = 0; j < ce->n; j++) {
            const int ncx = ce->v[j].x + ce->v[j].w / 2, ncy = ce->v[j].y + ce->v[j].h / 2;
            const double h2 = sqrt(pow((double)(ncx - ocx), 2) + pow((double)(ncy - ocy), 2));
            if (h2 < 7) { rl_push(res, old->v[i]); rl_erase(ce, j); break; }
        }
    }
    for (int j = 0; j < ce->n; j++) rl_push(res, ce->v[j]);
}
static void to_global(rlist *v, orc_rect face, int scale)
{
    for (int i = 0; i < v->n; i++) {
        v->v[i].x = (face.x + v->v[i].x) * scale; v->v[i].y = (face.y + v->v[i].y) * scale;
        v->v[i].w = (v->v[i].w - 1) * scale; v->v[i].h = (v->v[i].h - 1) * scale;
    }
}

/* ear: kms_ear_detect_find_ears */
static void find_ears(orc_part_stream *s, const uint8_t *face_img, int fcols, int frows, const uint8_t *ear_img, int ecols,
                      int erows, const orc_cascade *ear_cascade, double scale_f2e, double scale_e2o, int side)
{
    orc_rect tmp[MAXR];
    int nf = orc_detect_multiscale(s->face, face_img, fcols, frows, fcols, 1 + s->p.scale_factor_pct * 1.0 / 100, 2,
                                   ORC_HAAR_SCALE_IMAGE, 3, 3, 0, 0, s->p.policy, tmp, MAXR, NULL);
    s->faces.n = 0;
    for (int i = 0; i < nf; i++) rl_push(&s->faces, tmp[i]);
    if (nf == 0) return;
    rlist *ears = side == 0 ? &s->la : &s->lb;
    if (ears->n > 0) ears->n = 0;
    else if (s->no_det_a < 4) s->no_det_a += 1;
    else { s->no_det_a = 0; ears->n = 0; }
    for (int i = 0; i < s->faces.n; i++) {
        orc_rect *r = &s->faces.v[i];
        const int top_height = cv_round((float)r->h * 20 / 100), down_height = cv_round((float)r->h * 20 / 100);
        if (side == 0) {
            r->y = (int)((r->y + top_height) * scale_f2e);
            r->x = (int)((r->x + (r->w / 2)) * scale_f2e);
            r->h = (int)((r->h - down_height) * scale_f2e);
            r->w = (int)((r->w / 2) * scale_f2e + 50);
            if (r->x + r->w > ecols) r->w = ecols - r->x - 1;
        } else {
            r->y = (int)((r->y + top_height) * scale_f2e);
            r->x = (int)((fcols - r->x - r->w) * scale_f2e - 50);
            r->h = (int)((r->h - down_height) * scale_f2e);
            r->w = (int)((r->w / 2) * scale_f2e);
            if (r->x < 0) r->x = 0;
        }
        orc_rect ear[MAXR];
        int ne = detect_roi(ear_cascade, ear_img, ecols, erows, *r, 1.1, 3, ORC_HAAR_FIND_BIGGEST_OBJECT, 1, 1, s->p.policy, ear, MAXR);
        for (int e = 0; e < ne; e++) {
            orc_rect a;
            a.x = cv_round((r->x + ear[e].x) * scale_e2o); a.y = cv_round((r->y + ear[e].y) * scale_e2o);
            a.w = (int)((ear[e].w - 1) * scale_e2o); a.h = (int)((ear[e].h - 1) * scale_e2o);
            rl_push(ears, a);
        }
    }
}

int orc_part_stream_process(orc_part_stream *s, const uint8_t *bgr, int W, int H, int stride,
                            orc_rect *out_a, int cap_a, int *n_a, orc_rect *out_b, int cap_b, int *n_b)
{
    const int kind = s->p.kind;
    /* conf_images: all float arithmetic */
    const float o2f = (kind != ORC_PART_EAR && s->p.detect_event) ? ((float)W) / ((float)W) : ((float)W) / ((float)160);
    const float x2o = ((float)W) / ((float)s->p.width_to_process);          /* scale_o2e / n2o / m2o / e2o */
    const float f2x = ((float)o2f) / ((float)x2o);
    const double scale_o2f = o2f, scale_x2o = x2o, scale_f2x = f2x;
    int received = 1, early_return = 0;
    if (kind != ORC_PART_EAR) {                                            /* __receive_event */
        if (s->p.detect_event) {
            received = 0;
            if (s->qn > 0) {
                s->faces = s->queue[0];
                memmove(&s->queue[0], &s->queue[1], sizeof(rlist) * (s->qn - 1)); s->qn--;
                received = 1;
                s->num_frames_to_process = 10 / (5 - s->p.process_x_every_4);
            }
        }
        if (!received && s->num_frames_to_process <= 0) early_return = 1;
    }
    if (!early_return) {
        s->num_frame++;
        const int px = s->p.process_x_every_4;
        const int run = (2 == px && (1 == s->num_frame % 2)) || ((2 != px) && (s->num_frame <= px));
        rlist res_a, res_b; res_a.n = res_b.n = 0;
        if (run) {
            s->num_frames_to_process--;
            uint8_t *gray = (uint8_t *)malloc((size_t)W * H);
            orc_bgr2gray(bgr, W, H, stride, 3, gray, W);
            if (kind == ORC_PART_EYE) orc_equalize_hist(gray, W, H, W, gray, W);      /* EYE/kmseyedetect.cpp:950 */
            const int fw = cv_round(W / scale_o2f), fh = cv_round(H / scale_o2f);
            const int pw = cv_round(W / scale_x2o), ph = cv_round(H / scale_x2o);
            uint8_t *small = (uint8_t *)malloc((size_t)(fw > 0 ? fw : 1) * (fh > 0 ? fh : 1));
            uint8_t *part = (uint8_t *)malloc((size_t)pw * ph);
            orc_rect tmp[MAXR];
            if (kind == ORC_PART_EAR) {
                orc_resize_linear(gray, W, H, W, 1, small, fw, fh, fw);
                orc_equalize_hist(small, fw, fh, fw, small, fw);
                orc_resize_linear(gray, W, H, W, 1, part, pw, ph, pw);
                orc_equalize_hist(part, pw, ph, pw, part, pw);
                find_ears(s, small, fw, fh, part, pw, ph, s->a, scale_f2x, scale_x2o, 0);
                uint8_t *flip = (uint8_t *)malloc((size_t)fw * fh);
                orc_flip_h(small, fw, fh, fw, flip, fw);
                find_ears(s, flip, fw, fh, part, pw, ph, s->b, scale_f2x, scale_x2o, 1);
                free(flip);
            } else {
                if (0 == s->p.detect_event) {
                    orc_resize_linear(gray, W, H, W, 1, small, fw, fh, fw);
                    int nf;
                    if (kind == ORC_PART_EYE)
                        nf = orc_detect_multiscale(s->face, small, fw, fh, fw, 1 + s->p.scale_factor_pct * 1.0 / 100, 3, 0, 30, 30,
                                                   0, 0, s->p.policy, tmp, MAXR, NULL);
                    else {
                        orc_equalize_hist(small, fw, fh, fw, small, fw);
                        nf = orc_detect_multiscale(s->face, small, fw, fh, fw, 1 + s->p.scale_factor_pct * 1.0 / 100, 2,
                                                   ORC_HAAR_SCALE_IMAGE, 3, 3, 0, 0, s->p.policy, tmp, MAXR, NULL);
                    }
                    s->faces.n = 0;
                    for (int i = 0; i < nf; i++) rl_push(&s->faces, tmp[i]);
                }
                orc_resize_linear(gray, W, H, W, 1, part, pw, ph, pw);
                orc_equalize_hist(part, pw, ph, pw, part, pw);
                const int iscale = (int)scale_x2o;                 /* double -> int parameter */
                for (int i = 0; i < s->faces.n; i++) {
                    const orc_rect r = s->faces.v[i];
                    if (kind == ORC_PART_EYE) {
                        orc_rect ra, fr, fl;
                        ra.x = (int)(r.x * scale_f2x); ra.y = (int)(r.y * scale_f2x);
                        ra.w = (int)(r.w * scale_f2x); ra.h = (int)(r.h * scale_f2x);
                        const int down_height = cv_round((float)ra.h * 40 / 100), top_height = cv_round((float)ra.h * 25 / 100);
                        fr.x = ra.x; fr.y = ra.y + top_height; fr.h = ra.h - top_height - down_height; fr.w = ra.w / 2;
                        fl.x = ra.x + ra.w / 2; fl.y = ra.y + top_height; fl.h = ra.h - top_height - down_height; fl.w = ra.w / 2;
                        rlist eye_r, eye_l, aux;
                        int n = detect_roi(s->a, part, pw, ph, fr, 1.1, 2, ORC_HAAR_SCALE_IMAGE, 20, 20, s->p.policy, tmp, MAXR);
                        eye_r.n = 0; for (int k = 0; k < n; k++) rl_push(&eye_r, tmp[k]);
                        n = detect_roi(s->b, part, pw, ph, fl, 1.1, 2, ORC_HAAR_SCALE_IMAGE, 20, 20, s->p.policy, tmp, MAXR);
                        eye_l.n = 0; for (int k = 0; k < n; k++) rl_push(&eye_l, tmp[k]);
                        to_global(&eye_r, fr, iscale); to_global(&eye_l, fl, iscale);
                        if (eye_r.n > 0) {
                            merge_eyes_current(fr, &eye_r, &eye_r, iscale, 0);
                            merge_eyes_consecutive(&eye_r, &s->la, &aux);
                            for (int k = 0; k < aux.n; k++) rl_push(&res_a, aux.v[k]);
                        }
                        if (eye_l.n > 0) {
                            merge_eyes_current(fl, &res_a, &eye_l, iscale, 1);
                            merge_eyes_consecutive(&eye_l, &s->lb, &aux);
                            for (int k = 0; k < aux.n; k++) rl_push(&res_b, aux.v[k]);
                        }
                    } else {
                        orc_rect ra;
                        int dis;
                        if (kind == ORC_PART_NOSE) {
                            const int top = cv_round((float)r.h * 25 / 100), down = cv_round((float)r.h * 10 / 100);
                            const int side = cv_round((float)r.w * 25 / 100);
                            ra.y = (int)((r.y + top) * scale_f2x); ra.x = (int)((r.x + side) * scale_f2x);
                            ra.h = (int)((r.h - down - top) * scale_f2x); ra.w = (int)((r.w - side) * scale_f2x);
                            dis = 6;
                        } else {
                            const int half = cv_round((float)r.h / 1.8);
                            ra.y = (int)((r.y + half) * scale_f2x); ra.x = (int)(r.x * scale_f2x);
                            ra.h = (int)(half * scale_f2x); ra.w = (int)(r.w * scale_f2x);
                            dis = 4;
                        }
                        int n = detect_roi(s->a, part, pw, ph, ra, 1.1, 3, ORC_HAAR_FIND_BIGGEST_OBJECT, 1, 1, s->p.policy, tmp, MAXR);
                        if (n > 0) {
                            rlist cn, aux; cn.n = 0;
                            for (int k = 0; k < n; k++) rl_push(&cn, tmp[k]);
                            merge_consecutive_nm(&cn, &s->la, ra, iscale, dis, &aux);
                            for (int k = 0; k < aux.n; k++) rl_push(&res_a, aux.v[k]);
                        }
                    }
                }
            }
            free(gray); free(small); free(part);
            if (kind == ORC_PART_EYE) {                             /* per-side hysteresis :1034-1064 */
                if (res_a.n < 1) { if (s->no_det_a < 1) s->no_det_a += 1; else { s->no_det_a = 0; s->la.n = 0; } }
                else { s->no_det_a = 0; s->la = res_a; }
                if (res_b.n < 1) { if (s->no_det_b < 1) s->no_det_b += 1; else { s->no_det_b = 0; s->lb.n = 0; } }
                else { s->no_det_b = 0; s->lb = res_b; }
            }
        }
        if (kind == ORC_PART_NOSE || kind == ORC_PART_MOUTH) s->la = res_a;     /* cleared on every call that gets here */
        if (4 == s->num_frame) s->num_frame = 0;
    }
    int na = s->la.n < cap_a ? s->la.n : cap_a, nb = s->lb.n < cap_b ? s->lb.n : cap_b;
    for (int i = 0; i < na; i++) out_a[i] = s->la.v[i];
    for (int i = 0; i < nb; i++) out_b[i] = s->lb.v[i];
    *n_a = na; *n_b = nb;
    return 0;
}
