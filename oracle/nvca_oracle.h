/*
 * nvca_oracle.h -- CPU restatement of the OpenCV-2.4 arithmetic behind
 * NUBOMEDIA-VCA's per-frame Haar detection path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.
 *
 * PARITY UNPINNED: the reference (/root/reference) holds no tests, fixtures or
 * golden vectors, and the library that performs its arithmetic (OpenCV 2.4.x,
 * pinned only as `opencv>=2.0.0` in
 * modules/nubo_face/nubo-face-detector/CMakeLists.txt:35,46) is absent from
 * the build container together with its haarcascade XML files.  This file
 * restates OpenCV 2.4.8's published algorithms and the
 * reference's own call sites; it is pinned only by closed-form known answers
 * (tests/test_oracle_*.py).
 *
 * All functions are plain C, single-threaded unless stated, deterministic.
 * Compile with -ffp-contract=off (see oracle/Makefile).
 */
#ifndef NVCA_ORACLE_H
#define NVCA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int x, y, w, h; } orc_rect;

/* detectMultiScale flags (OpenCV objdetect: CV_HAAR_*) */
#define ORC_HAAR_DO_CANNY_PRUNING    1
#define ORC_HAAR_SCALE_IMAGE         2
#define ORC_HAAR_FIND_BIGGEST_OBJECT 4
#define ORC_HAAR_DO_ROUGH_SEARCH     8

/* feature-sum accumulation policy */
#define ORC_SUM_F32PAIR 0 /* SSE2 path: 2-rect stages add the two products in f32 */
#define ORC_SUM_F64     1 /* plain C path: accumulate in double                   */

/* ---- imgproc ---------------------------------------------------------- */
/* cv::cvtColor(BGR2GRAY / BGRA2GRAY), 8U.  cn = 3 or 4. */
void orc_bgr2gray(const uint8_t *src, int w, int h, int sstride, int cn,
                  uint8_t *dst, int dstride);
/* cv::resize(INTER_LINEAR), 8U, cn = 1 or 3 (or 4). */
void orc_resize_linear(const uint8_t *src, int sw, int sh, int sstride, int cn,
                       uint8_t *dst, int dw, int dh, int dstride);
/* cv::equalizeHist (>= 2.4.3).  In place allowed. */
void orc_equalize_hist(const uint8_t *src, int w, int h, int sstride,
                       uint8_t *dst, int dstride);
/* the 256-entry LUT equalizeHist applies; returns 1 if the image is constant
 * (then lut[] is filled with that constant) */
int orc_equalize_lut(const int hist[256], int total, uint8_t lut[256]);
/* cv::integral: sum int32 and sqsum double, both (h+1)*(w+1) dense. */
void orc_integral(const uint8_t *src, int w, int h, int stride,
                  int32_t *sum, double *sqsum);
/* the tilted integral of cv::integral (int32, (h+1)*(w+1) dense): tilted(X,Y) = sum of image(x,y) over
 * y < Y, abs(x - X + 1) <= Y - y - 1.  Read by tilted Haar features (cvSetImagesForHaarClassifierCascade). */
void orc_integral_tilted(const uint8_t *src, int w, int h, int stride, int32_t *tilted);
/* cv::flip(src, dst, 1) */
void orc_flip_h(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);

/* ---- cascade ---------------------------------------------------------- */
/* Flat description of an old-format ("opencv-haar-classifier") cascade,
 * produced by the python XML reader (tests) -- independent of the product's
 * own C++ loader.  All floating values are already cast to float exactly as
 * OpenCV's loader does ((float)fn->data.f). */
typedef struct {
    int ow, oh;             /* <size> */
    int n_stages;
    const int   *stage_ncls;   /* [n_stages] classifiers (trees) per stage   */
    const float *stage_thr;    /* [n_stages] <stage_threshold>               */
    int n_cls;
    const int   *cls_nnodes;   /* [n_cls] nodes per classifier (1 = stump)   */
    int n_nodes;
    const int   *rects;        /* [n_nodes*3*4] x,y,w,h ; all-zero if absent */
    const float *rweights;     /* [n_nodes*3]                                */
    const int   *tilted;       /* [n_nodes]                                  */
    const float *node_thr;     /* [n_nodes]                                  */
    const int   *left;         /* [n_nodes]  >0: child node, <=0: -alpha idx */
    const int   *right;        /* [n_nodes]                                  */
    const float *alpha;        /* per classifier nnodes+1 values, concatenated */
} orc_cascade;

typedef struct {
    int64_t windows;        /* windows handed to runCascade           */
    int64_t stumps;         /* weak classifiers evaluated             */
    int64_t raw_hits;       /* candidates before grouping             */
    int     n_scales;       /* scales actually evaluated              */
    int     pad_;
    int64_t stage_enter[64]; /* windows that entered stage i (i < 64): the per-stage selectivity of a cascade on a frame */
} orc_stats;

/* cv::CascadeClassifier::detectMultiScale for an old-format cascade
 * (= cvHaarDetectObjectsForROC).  Returns number of rects written (<= cap).
 * max_w/max_h == 0 means image size.  Raw candidates are produced in the
 * canonical serial order (scale, y, x). */
int orc_detect_multiscale(const orc_cascade *c, const uint8_t *gray, int w, int h,
                          int stride, double scale_factor, int min_neighbors,
                          int flags, int min_w, int min_h, int max_w, int max_h,
                          int policy, orc_rect *out, int cap, orc_stats *stats);
/* Same, but stops before grouping: the raw candidate list (not valid with
 * FIND_BIGGEST_OBJECT, whose grouping is interleaved with the scan). */
int orc_detect_raw(const orc_cascade *c, const uint8_t *gray, int w, int h,
                   int stride, double scale_factor, int flags, int min_w,
                   int min_h, int max_w, int max_h, int policy, orc_rect *out,
                   int cap, orc_stats *stats);
/* cv::groupRectangles(rects, groupThreshold, eps); in place, returns new n.
 * weights (may be NULL) receives the neighbour count of each kept rect. */
int orc_group_rectangles(orc_rect *rects, int n, int group_threshold, double eps,
                         int *weights);

/* Scale grid of the scale-cascade branch, for host-side table building tests:
 * writes factors of the scales that are actually evaluated. */
int orc_scale_grid(int ow, int oh, int w, int h, double scale_factor, int min_w,
                   int min_h, int max_w, int max_h, double *factors, int cap);

/* ---- NuboFaceDetector glue (FACE/kmsfacedetect.cpp, Faces.cpp) --------- */
typedef struct orc_face_stream orc_face_stream;
typedef struct {
    int width_to_process;        /* FACE/kmsfacedetect.cpp:26  (160)  */
    int process_x_every_4;       /* :24  (4)                          */
    int scale_factor_pct;        /* :25  (25 -> 1.25)                 */
    int track_threshold;         /* :33  (40)                         */
    int euclidean_threshold;     /* :32  (8)                          */
    int area_threshold;          /* :34  (500)                        */
    int full_res;                /* benchmark mode: scale = 1 */
    int min_neighbors;           /* 3 (:810)                          */
    int policy;
} orc_face_params;
void orc_face_params_default(orc_face_params *p);
orc_face_stream *orc_face_stream_create(const orc_cascade *c, const orc_face_params *p);
void orc_face_stream_destroy(orc_face_stream *s);
/* One kms_face_detect_transform_frame_ip: returns number of boxes as emitted
 * by kms_face_send_event (original-frame pixels, i.e. * norm_scale). ids
 * (may be NULL) receives the Faces ids. */
int orc_face_stream_process(orc_face_stream *s, const uint8_t *bgr, int w, int h,
                            int stride, orc_rect *out, int *ids, int cap);
/* The same in three pieces, so that a harness which feeds the same frames again and again (bench.py) can compute each
 * distinct frame's detections once and replay the temporal logic: process == gate -> (analysed ? frame_detect) -> finish. */
int orc_face_stream_gate(orc_face_stream *s);
int orc_face_frame_detect(const orc_cascade *c, const orc_face_params *p, const uint8_t *bgr, int w, int h, int stride,
                          orc_rect *cur, int cap);
int orc_face_stream_finish(orc_face_stream *s, int analysed, const orc_rect *cur, int n_cur, int w, int h,
                           orc_rect *out, int *ids, int cap);
/* Faces::track_faces on explicit lists (for unit tests).  State in/out:
 * faces[n_faces] with ids; cur[n_cur] current detections. Returns new count. */
int orc_track_faces(orc_rect *faces, int *ids, int n_faces, int *next_id,
                    const orc_rect *cur, int n_cur, int track_threshold, int cap);

/* ---- NuboTracker (TRK/gstnubotracker.cpp) ------------------------------ */
typedef struct orc_tracker orc_tracker;
typedef struct {
    int  threshold;  /* 20    */
    int  min_area;   /* 50    */
    long max_area;   /* 30000 */
    int  distance;   /* 35    */
    double mhi_duration; /* 0.2 */
    double seg_thresh;   /* 32  */
} orc_tracker_params;
void orc_tracker_params_default(orc_tracker_params *p);
orc_tracker *orc_tracker_create(const orc_tracker_params *p);
void orc_tracker_destroy(orc_tracker *t);
/* One gst_nubo_tracker_process on a BGRA frame with explicit timestamp (ms). */
int orc_tracker_process(orc_tracker *t, const uint8_t *bgra, int w, int h, int stride,
                        double timestamp_ms, orc_rect *out, int cap);
/* pieces, for unit tests */
void orc_update_mhi(const uint8_t *silh, int w, int h, float *mhi, double ts, double dur);
int  orc_segment_motion(float *mhi, int w, int h, double ts, double seg_thresh,
                        orc_rect *out, int cap);
int  orc_join_objects(orc_rect *r, int n, int min_area, long max_area, int distance);

/* ---- part detectors (EYE/ NOSE/ MOUTH/ EAR/ kms*detect.cpp process_frame) -- */
#define ORC_PART_EYE   0
#define ORC_PART_NOSE  1
#define ORC_PART_MOUTH 2
#define ORC_PART_EAR   3
typedef struct orc_part_stream orc_part_stream;
typedef struct {
    int kind;
    int width_to_process;    /* 320 (EYE_WIDTH etc.)          */
    int process_x_every_4;   /* 4                             */
    int scale_factor_pct;    /* 25                            */
    int detect_event;        /* 0                             */
    int policy;
} orc_part_params;
void orc_part_params_default(orc_part_params *p, int kind);
/* cascades: face (frontalface_alt; profileface for EAR), a, b:
 *   EYE: a = righteye, b = lefteye;  EAR: a = LEFT_SIDE cascade, b = RIGHT_SIDE cascade;  NOSE/MOUTH: a only */
orc_part_stream *orc_part_stream_create(const orc_part_params *p, const orc_cascade *face, const orc_cascade *a,
                                        const orc_cascade *b);
void orc_part_stream_destroy(orc_part_stream *s);
/* detect-event mode: one upstream "message" worth of faces (original-frame pixels) */
void orc_part_stream_push_faces(orc_part_stream *s, const orc_rect *faces, int n);
/* one transform_frame_ip; list A: eyes_r / noses / mouths / lear, list B: eyes_l / rear */
int orc_part_stream_process(orc_part_stream *s, const uint8_t *bgr, int w, int h, int stride,
                            orc_rect *out_a, int cap_a, int *n_a, orc_rect *out_b, int cap_b, int *n_b);

#ifdef __cplusplus
}
#endif
#endif
