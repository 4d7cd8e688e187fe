/*
 * nubovca.h -- C ABI of libnubovca_hip: the MI355X (gfx950) implementation of
 * NUBOMEDIA-VCA's per-frame Haar detection hot path.
 *
 * Every entry point replaces a call the reference makes into OpenCV 2.4 (or a
 * static function of the reference's GStreamer elements); the reference site
 * each one replaces is cited as file:line relative to the reference tree, with
 *   FACE/ = modules/nubo_face/nubo-face-detector/src/gst-plugins/
 *   TRK/  = modules/nubo_tracker/nubo-tracker/src/gst-plugins/
 *   EYE/ NOSE/ MOUTH/ EAR/ analogous.
 *
 * Conventions: plain C types only; opaque handles; the caller owns all frame
 * memory; the library owns device state.  Every function returns an int status
 * (NVCA_OK == 0, < 0 error) and never throws: every entry point is a function-try-block
 * (csrc/nvca_internal.h, NVCA_API_CATCH) that turns std::bad_alloc / std::length_error into NVCA_ERR_NOMEM and anything
 * else into NVCA_ERR_INTERNAL -- the reference never lets a frame error out of the element either
 * (FACE/kmsfacedetect.cpp:897 always returns GST_FLOW_OK).  A handle may be used by one
 * thread at a time; distinct contexts may be used concurrently.  There is no
 * CPU fallback: without a HIP device nvca_ctx_create fails.
 */
#ifndef NUBOVCA_H
#define NUBOVCA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVCA_OK              0
#define NVCA_ERR_ARG        -1   /* bad argument                                   */
#define NVCA_ERR_NO_DEVICE  -2   /* no HIP device / device id out of range         */
#define NVCA_ERR_HIP        -3   /* a HIP runtime call failed (see nvca_last_error)*/
#define NVCA_ERR_IO         -4   /* cascade file unreadable                        */
#define NVCA_ERR_PARSE      -5   /* cascade XML malformed                          */
#define NVCA_ERR_UNSUPPORTED -6  /* tree-structured STAGE graph (parent / next), or a
                                    feature that leaves the image at some scale    */
#define NVCA_ERR_OVERFLOW   -7   /* more raw candidates than the context's cap     */
#define NVCA_ERR_NOMEM      -8
#define NVCA_ERR_INTERNAL   -9   /* an internal check failed (a device result out of range, an exception caught at the
                                    ABI boundary): the call's results are void, the context stays usable          */

/* where a buffer handed to the library lives */
#define NVCA_MEM_HOST   0
#define NVCA_MEM_DEVICE 1

/* detectMultiScale flags -- values of OpenCV's CV_HAAR_* */
#define NVCA_HAAR_DO_CANNY_PRUNING    1   /* accepted, ignored (never set by the reference) */
#define NVCA_HAAR_SCALE_IMAGE         2
#define NVCA_HAAR_FIND_BIGGEST_OBJECT 4
#define NVCA_HAAR_DO_ROUGH_SEARCH     8

/* feature-sum accumulation policy (OpenCV build variant) */
#define NVCA_SUM_F32PAIR 0   /* SSE2 build (distro default): 2-rect stages add in f32 */
#define NVCA_SUM_F64     1   /* plain C build                                          */

typedef struct nvca_ctx nvca_ctx;
typedef struct nvca_cascade nvca_cascade;
typedef struct nvca_face_stream nvca_face_stream;
typedef struct nvca_tracker nvca_tracker;

typedef struct nvca_rect { int x, y, w, h; } nvca_rect;

typedef struct nvca_frame {
    const void *data;     /* packed rows, BGR (3 B/px) or BGRA (4 B/px)              */
    int width, height;
    int stride;           /* bytes per row; the reference assumes align4(width*bpp),
                             FACE/kmsfacedetect.cpp:300-305                          */
    int mem;              /* NVCA_MEM_HOST or NVCA_MEM_DEVICE                        */
    uint64_t pts;
} nvca_frame;

/* ---- context ----------------------------------------------------------- */
/* HIP devices visible to the process (the GStreamer shim spreads its elements over them, one context per GPU) */
int  nvca_device_count(int *n);
int  nvca_ctx_create(int device_id, nvca_ctx **out);
void nvca_ctx_destroy(nvca_ctx *ctx);
/* text of the last error on this context (never NULL) */
const char *nvca_last_error(const nvca_ctx *ctx);
const char *nvca_version(void);
/* Raw-candidate capacity per frame the lists start with (default 16384, at most 2^22).  A detectMultiScale call
 * (nvca_detect_multiscale / _raw, the part detectors' searches) that produces more re-runs its launch set once with lists of
 * the exact size and answers as the reference does -- a CV_HAAR_FIND_BIGGEST_OBJECT search with a lenient cascade included
 * (NOSE/kmsnosedetect.cpp:870-873).  The batched face path reports NVCA_ERR_OVERFLOW for the batch that overflowed (never a
 * truncated list) and sizes the lists of the following batches for what that batch produced. */
int  nvca_ctx_set_hit_capacity(nvca_ctx *ctx, int cap);
int  nvca_ctx_set_sum_policy(nvca_ctx *ctx, int policy);
/* Measurement / bisecting switches, per context.  The process-wide defaults come from the environment
 * (NVCA_BAND, NVCA_TILES, ... read once, when the first context is created); this sets one for this context.  Names: "band"
 * (-1 / 0 / 1), "band_map", "tiles", "deep_stage", "deep_lds", "pyr_off", "host_group", "group_zerocopy", "sparse_ingest",
 * "ingest_chunk", "skip_cascade", "host_profile", "part_stats", "trk_order", "plan_debug", "quiet", "roi", "host_threads", "two_lanes", "pre_cus".  None of them changes a
 * result.  Switches that shape plans drop the context's cached plans (not while a submitted batch is in flight).
 * nvca_ctx_get_option reads the value a switch holds for this context (so that a caller can put it back). */
int  nvca_ctx_set_option(nvca_ctx *ctx, const char *name, int value);
int  nvca_ctx_get_option(nvca_ctx *ctx, const char *name, int *value);
/* block until everything queued on the context's HIP stream has finished */
int  nvca_ctx_synchronize(nvca_ctx *ctx);
/* the hipStream_t the context launches on (for interop / profiling) */
void *nvca_ctx_stream(nvca_ctx *ctx);
/* Zero-copy ingest: page-lock caller-owned frame memory (e.g. the buffers of a GstBufferPool, mapped by
 * kms_face_detect_conf_images, FACE/kmsfacedetect.cpp:282-306) so that the H2D copies of NVCA_MEM_HOST frames are true
 * asynchronous DMA.  Optional: pageable frames work, they are just copied through the driver's staging buffers.
 * Unregister before the memory is freed. */
int  nvca_host_register(nvca_ctx *ctx, void *ptr, size_t bytes);
int  nvca_host_unregister(nvca_ctx *ctx, void *ptr);

/* Per-kernel timing with HIP events on the context's stream.  While enabled,
 * kernel launches are bracketed by events; nvca_ctx_kernel_timing drains them.
 * on == 0: off; on == 1: every launch; on == N > 1: the launches of every N-th
 * batch of the face-detector / tracker entry points only (the first one included) --
 * event-carrying launches do not overlap their neighbours, which costs a few
 * microseconds per launch. */
#define NVCA_K_GRAY      0  /* resize + BGR2GRAY + histogram            */
#define NVCA_K_LUT       1  /* equalizeHist LUT                         */
#define NVCA_K_COLSUM    2  /* integral: band column sums               */
#define NVCA_K_BANDSCAN  3  /* integral: scan over bands                */
#define NVCA_K_INTEGRAL  4  /* integral: row scan + write               */
#define NVCA_K_STAGE0    5  /* cascade: variance + stage 0, all windows */
#define NVCA_K_STRIP     6  /* cascade: early stages, window per lane   */
#define NVCA_K_DEEP      7  /* cascade: late stages, stump per lane     */
#define NVCA_K_GROUP     8  /* candidate sort + groupRectangles         */
#define NVCA_K_TRACKER   9  /* tracker pixel pass + labelling           */
#define NVCA_K_RESIZE1  10  /* 8UC1 resize (pyramid / parts)            */
#define NVCA_K_TILE     11  /* cascade: early stages from LDS-staged tiles */
#define NVCA_K_BAND     12  /* cascade: variance + stage 0 + early stages, one row of tiles per workgroup */
#define NVCA_K_ROI     13  /* cascade: a whole detectMultiScale on a small image (a face region) per workgroup */
#define NVCA_K_COUNT    14
int  nvca_ctx_enable_kernel_timing(nvca_ctx *ctx, int on);
/* total_ms[NVCA_K_COUNT], launches[NVCA_K_COUNT]; resets the accumulators */
int  nvca_ctx_kernel_timing(nvca_ctx *ctx, double *total_ms, int64_t *launches);
const char *nvca_kernel_name(int k);

/* ---- cascade: replaces cv::CascadeClassifier::load ---------------------
 * FACE/kmsfacedetect.cpp:162-177 (HAAR_CONF_FILE :40), EYE/kmseyedetect.cpp:27-29,
 * NOSE/kmsnosedetect.cpp:31-32, MOUTH/kmsmouthdetect.cpp:37-38, EAR/kmseardetect.cpp:29-31.
 * Old-format ("opencv-haar-classifier") XML only: stump or tree-structured weak classifiers, upright or tilted
 * features (cascades with trees / tilted features run through the general evaluator). */
int  nvca_cascade_load_xml(nvca_ctx *ctx, const char *path, nvca_cascade **out);
int  nvca_cascade_load_mem(nvca_ctx *ctx, const char *xml, int64_t len, nvca_cascade **out);
void nvca_cascade_free(nvca_cascade *c);
/* The loader alone: parses an old-format cascade on the host and reports its shape (any out pointer may be NULL) or the
 * loader's error text -- no device and no context needed, so cascade files can be checked on a box without a GPU.
 * Same status codes as nvca_cascade_load_mem. */
int  nvca_cascade_validate_mem(const char *xml, int64_t len, int *win_w, int *win_h, int *n_stages, int *n_weak,
                               char *err, int err_cap);
/* Exercises the exception barrier of the ABI (see "never throws" above): throws the exception `kind` names from inside an
 * entry point and returns what the barrier made of it -- 0: std::bad_alloc -> NVCA_ERR_NOMEM, 1: std::length_error ->
 * NVCA_ERR_NOMEM, 2: std::runtime_error, 3: a non-standard exception, 4: std::out_of_range -> NVCA_ERR_INTERNAL. */
int  nvca_abi_selftest(int kind);
/* window size, stage count, weak-classifier count (any pointer may be NULL) */
int  nvca_cascade_info(const nvca_cascade *c, int *win_w, int *win_h, int *n_stages, int *n_weak);
/* has_tilted / has_trees (either pointer may be NULL): what kind of cascade the loader found */
int  nvca_cascade_kind(const nvca_cascade *c, int *has_tilted, int *has_trees);
/* flat dump for cross-checking the loader: arrays sized by nvca_cascade_info;
 * rects[n_weak*12] (3 rects x,y,w,h), weights[n_weak*3], thr/left_val/right_val[n_weak],
 * stage_sizes/stage_thr[n_stages] (stump cascades only) */
int  nvca_cascade_dump(const nvca_cascade *c, int *rects, float *weights, float *thr,
                       float *left_val, float *right_val, int *stage_sizes, float *stage_thr);

/* ---- imgproc primitives (each = one cv:: call of the reference) -------- */
/* cv::cvtColor(CV_BGR2GRAY) FACE/kmsfacedetect.cpp:806, TRK/gstnubotracker.cpp:356 (channels 4) */
int nvca_bgr2gray(nvca_ctx *ctx, const void *src, int w, int h, int stride, int channels,
                  int mem, void *dst_gray, int dst_stride);
/* cv::resize(INTER_LINEAR) FACE/kmsfacedetect.cpp:805 (3 ch), EYE/kmseyedetect.cpp:956,963 (1 ch) */
int nvca_resize_linear(nvca_ctx *ctx, const void *src, int sw, int sh, int sstride, int channels,
                       int mem, void *dst, int dw, int dh, int dstride);
/* cv::equalizeHist FACE/kmsfacedetect.cpp:807 */
int nvca_equalize_hist(nvca_ctx *ctx, const void *src_gray, int w, int h, int stride, int mem,
                       void *dst_gray, int dst_stride);
/* cv::flip(src, dst, 1) EAR/kmseardetect.cpp:800 */
int nvca_flip_horizontal(nvca_ctx *ctx, const void *src_gray, int w, int h, int stride, int mem,
                         void *dst_gray, int dst_stride);
/* view-faces / view-eyes / ...: the outlines the elements draw on a viewed frame, on the frame where it is --
 * host memory (the GStreamer shim's mapped buffer) or device memory (a viewed stream that stays in HBM makes no round trip).
 * NVCA_SHAPE_RECT3: cvRectangle(img, (x, y), (x + w, y + h), colour, 3, 8, 0) as FACE/kmsfacedetect.cpp:836-845,
 * TRK/gstnubotracker.cpp:388-395 and the nose / mouth / ear elements call it: the 3-pixel band around the outline, the four
 * outermost corner pixels left out (round joins of radius 1).  NVCA_SHAPE_RING4: circle(img, (x, y), w, colour, 4, 8, 0),
 * EYE/kmseyedetect.cpp:1075-1092: the pixels at distance [w - 2, w + 2] from the centre.  Shapes are drawn in order.
 * The rasterisation rules are this library's statement of OpenCV's thick-line code, not pixel-verified against it.
 * Host frames need no device: ctx may be NULL for them. */
#define NVCA_SHAPE_RECT3 0
#define NVCA_SHAPE_RING4 1
typedef struct { int kind; int x, y, w, h; uint8_t bgra[4]; } nvca_shape;
int nvca_draw_shapes(nvca_ctx *ctx, const nvca_frame *frame, int channels, const nvca_shape *shapes, int n);
/* image-to-overlay: kms_face_detect_display_detections_overlay_img, FACE/kmsfacedetect.cpp:427-502 -- for every
 * box, in order, the overlay image is scaled (cvResize, CV_INTER_LINEAR) to (box.w * width_percent) x (box.h * height_percent),
 * placed at box.x + box.w * offset_x_percent, box.y + box.h * offset_y_percent (truncated as the reference's int arithmetic does)
 * and written onto the BGR frame where it lies inside it: 1 channel -> copied to B, G and R; 3 channels -> copied; 4 channels ->
 * blended per pixel with weight alpha / 255 in double arithmetic, truncated to 8 bits.  Boxes are frame pixels (the reference
 * passes box * scale, :840-844).  The image (what cvLoadImage(..., CV_LOAD_IMAGE_UNCHANGED) returned: fetching and decoding it
 * stay with the element) is host memory; the frame may be host memory (plain loops, ctx may be NULL) or device memory (one kernel
 * per box; a viewed stream that stays in HBM makes no round trip).  A box whose scaled size is not positive is skipped (the
 * reference's cvCreateImage would throw there). */
typedef struct nvca_overlay {
    const void *data; int width, height, stride, channels;       /* 8-bit, 1 / 3 / 4 interleaved channels, host memory */
    double offset_x_percent, offset_y_percent, width_percent, height_percent;   /* the image-to-overlay structure's fields, :351-367 */
} nvca_overlay;
int nvca_overlay_blend(nvca_ctx *ctx, const nvca_frame *frame_bgr, const nvca_rect *boxes, int n, const nvca_overlay *overlay);
/* cv::integral as used inside detectMultiScale: sum int32 and sqsum float64,
 * both dense (h+1)*(w+1) */
int nvca_integral(nvca_ctx *ctx, const void *src_gray, int w, int h, int stride, int mem,
                  int32_t *sum, double *sqsum);

/* the third plane of cv::integral, which cvHaarDetectObjectsForROC asks for when the cascade holds tilted features
 * (haarcascade_profileface.xml, EAR/kmseardetect.cpp:29): tilted(X,Y) = sum of image(x,y) over y < Y,
 * abs(x - X + 1) <= Y - y - 1; int32, dense (h+1)*(w+1) */
int nvca_integral_tilted(nvca_ctx *ctx, const void *src_gray, int w, int h, int stride, int mem, int32_t *tilted);

/* ---- detectMultiScale ---------------------------------------------------
 * cv::CascadeClassifier::detectMultiScale(gray, objects, scaleFactor, minNeighbors,
 * flags, minSize, maxSize) -- FACE/kmsfacedetect.cpp:809-811 and the sibling call
 * sites listed in SURVEY.md Appendix B.  max_w/max_h == 0 -> image size.
 * Objects are written in OpenCV's serial order; *n_out may exceed cap (then only
 * cap are written). */
int nvca_detect_multiscale(nvca_ctx *ctx, const nvca_cascade *cascade, const void *gray,
                           int w, int h, int stride, int mem, double scale_factor,
                           int min_neighbors, int flags, int min_w, int min_h, int max_w,
                           int max_h, nvca_rect *out, int cap, int *n_out);
/* raw candidates before groupRectangles, canonical (scale, y, x) order; not
 * defined for FIND_BIGGEST_OBJECT */
int nvca_detect_raw(nvca_ctx *ctx, const nvca_cascade *cascade, const void *gray, int w, int h,
                    int stride, int mem, double scale_factor, int flags, int min_w, int min_h,
                    int max_w, int max_h, nvca_rect *out, int cap, int *n_out);
/* cv::groupRectangles(rects, groupThreshold, eps) on the device; in place */
int nvca_group_rectangles(nvca_ctx *ctx, nvca_rect *rects, int n, int group_threshold, double eps,
                          int *n_out);

/* ---- NuboFaceDetector stream -------------------------------------------
 * One handle == one element instance == one media stream.  Replaces
 * kms_face_detect_conf_images + kms_face_detect_process_frame + the box scaling
 * of kms_face_send_event (FACE/kmsfacedetect.cpp:282-306, 757-853, 190-211),
 * with Faces::track_faces (FACE/Faces.cpp:78-153) kept on the host. */
typedef struct nvca_face_params {
    int width_to_process;      /* "width-to-process", default 160  (:26)            */
    int process_x_every_4;     /* "process-x-every-4-frames", 4    (:24)            */
    int scale_factor_pct;      /* "multi-scale-factor", 25 -> 1.25 (:25,142)        */
    int track_threshold;       /* 40 (:33)                                          */
    int euclidean_threshold;   /* "euclidean-distance" 8 (:32) (unused by track_faces)*/
    int area_threshold;        /* 500 (:34) (unused by track_faces)                 */
    int min_neighbors;         /* 3 (:810)                                          */
    int detect_event;          /* "detect-event" 0 (:722): 1 = analyse only after
                                  nvca_face_stream_motion_event()                   */
} nvca_face_params;
void nvca_face_params_default(nvca_face_params *p);
int  nvca_face_stream_create(nvca_ctx *ctx, const nvca_cascade *cascade,
                             const nvca_face_params *params, nvca_face_stream **out);
void nvca_face_stream_destroy(nvca_face_stream *s);
int  nvca_face_stream_set_params(nvca_face_stream *s, const nvca_face_params *params);
/* a "motion" custom event arrived (FACE/kmsfacedetect.cpp:680-755): analyse the
 * next NUM_FRAMES_TO_PROCESS (10) frames */
int  nvca_face_stream_motion_event(nvca_face_stream *s);
/* One kms_face_detect_transform_frame_ip (FACE/kmsfacedetect.cpp:857-898): boxes
 * in original-frame pixels exactly as kms_face_send_event emits them; ids (may
 * be NULL) are the Faces ids. */
int  nvca_face_stream_process(nvca_face_stream *s, const nvca_frame *frame, nvca_rect *out,
                              int *ids, int cap, int *n_out);
/* Batched frontend: frame i belongs to streams[i]; a stream may appear several
 * times (consecutive frames, in order).  All analysed frames go through one
 * launch set per distinct geometry.  out is [n][cap], ids (may be NULL) likewise. */
int  nvca_face_batch_process(nvca_ctx *ctx, int n, nvca_face_stream *const *streams,
                             const nvca_frame *frames, nvca_rect *out, int *ids, int cap,
                             int *n_out);
/* The same in two halves, for a serving loop that keeps the GPU busy across batches: submit() gates the frames and queues
 * every launch, collect() waits for that batch and runs the temporal logic (Faces::track_faces, FACE/Faces.cpp:78-153).
 * Up to two batches may be in flight; they are collected in submission order.  Frame memory and the streams of a batch
 * must stay valid until the batch has been collected. */
int  nvca_face_batch_submit(nvca_ctx *ctx, int n, nvca_face_stream *const *streams, const nvca_frame *frames, int *ticket);
int  nvca_face_batch_collect(nvca_ctx *ctx, int ticket, nvca_rect *out, int *ids, int cap, int *n_out);

/* ---- part detectors: NuboEyeDetector / NuboNoseDetector / NuboMouthDetector / NuboEarDetector ----
 * One handle == one element instance.  Replaces kms_{eye,nose,mouth,ear}_detect_conf_images +
 * _process_frame (EYE/kmseyedetect.cpp:310-341,915-1064; NOSE/kmsnosedetect.cpp:275-308,792-868;
 * MOUTH/kmsmouthdetect.cpp:285-315,798-873; EAR/kmseardetect.cpp:292-319,644-729,767-812): gray (+equalize) at
 * full resolution, a face pass on the 160-px image (or faces handed over by an upstream element), then per face
 * the reference's ROI geometry and part cascades; the per-frame merging heuristics stay on the host. */
#define NVCA_PART_EYE   0
#define NVCA_PART_NOSE  1
#define NVCA_PART_MOUTH 2
#define NVCA_PART_EAR   3
typedef struct nvca_part_stream nvca_part_stream;
typedef struct nvca_part_params {
    int kind;                /* NVCA_PART_*                                                  */
    int width_to_process;    /* "width-to-process", 320 (EYE/kmseyedetect.cpp:25)            */
    int process_x_every_4;   /* "process-x-every-4-frames", 4                                */
    int scale_factor_pct;    /* "multi-scale-factor", 25 (face pass)                         */
    int detect_event;        /* "detect-event": 1 = faces come from nvca_part_stream_push_faces */
} nvca_part_params;
void nvca_part_params_default(nvca_part_params *p, int kind);
/* cascades: face (frontalface_alt; profileface for EAR); a and b:
 *   EYE  a = mcs_righteye, b = mcs_lefteye (EYE/kmseyedetect.cpp:28-29)
 *   EAR  a = LEFT_SIDE cascade (mcs_rightear, sic), b = RIGHT_SIDE cascade (EAR/kmseardetect.cpp:29-31,796,801)
 *   NOSE / MOUTH  a only (b may be NULL) */
int  nvca_part_stream_create(nvca_ctx *ctx, const nvca_part_params *params, const nvca_cascade *face,
                             const nvca_cascade *a, const nvca_cascade *b, nvca_part_stream **out);
void nvca_part_stream_destroy(nvca_part_stream *s);
int  nvca_part_stream_set_params(nvca_part_stream *s, const nvca_part_params *params);
/* one upstream "message" worth of faces, original-frame pixels (EYE/kmseyedetect.cpp:680-764) */
int  nvca_part_stream_push_faces(nvca_part_stream *s, const nvca_rect *faces, int n);
/* the face list the last processed frame worked with (working-image pixels of the face pass, or the pushed
 * faces): what kms_mouth_send_event / kms_ear_send_event put into their events */
int  nvca_part_stream_faces(const nvca_part_stream *s, nvca_rect *out, int cap, int *n_out);
/* One transform_frame_ip.  List A: eyes_r / noses / mouths / left ears; list B: eyes_l / right ears. */
int  nvca_part_stream_process(nvca_part_stream *s, const nvca_frame *frame_bgr, nvca_rect *out_a, int cap_a,
                              int *n_a, nvca_rect *out_b, int cap_b, int *n_b);

/* Batched frontend for the part detectors (BASELINE config 3, the ROI chain): frame i belongs to streams[i] (a stream at most
 * once per call; the streams may be of different kinds).  The device work of all streams is queued together: one wait for
 * every face pass, one for every part search, instead of several per stream.  Streams handed the same frame (same data
 * pointer and geometry: the detectors of one video stream) share its upload and whatever they compute identically from it;
 * working images of all frames come out of one launch set per size, face passes run as N-image jobs.  Every frame is
 * validated before any stream's frame gate advances, and a call that fails later (a plan that cannot be built, an allocation,
 * a refused launch) restores every stream's gates, queued face events and lists before it returns: ANY error leaves all
 * streams as they were, and nothing of the call stays in flight.  Results are those of
 * nvca_part_stream_process per stream.  out_a is [n][cap_a], out_b [n][cap_b]. */
int  nvca_part_batch_process(nvca_ctx *ctx, int n, nvca_part_stream *const *streams, const nvca_frame *frames,
                             nvca_rect *out_a, int cap_a, int *n_a, nvca_rect *out_b, int cap_b, int *n_b);

/* The same call in two halves, for a serving loop that keeps the GPU busy between the waits of a call (the counterpart of
 * nvca_face_batch_submit / _collect; the reference has no such boundary: its elements run one frame at a time on their streaming
 * threads, kms*detect.cpp transform_frame_ip).  submit: the streams' frame gates, the working images and the face passes are
 * queued -- nothing is waited for -- and *ticket names the call; the frames must stay valid until the ticket is collected.
 * collect: the face passes' results, the part searches, the merging heuristics; outputs as nvca_part_batch_process.  Tickets are
 * collected in submit order (the streams' state machines advance in that order); at most two are outstanding, so the usual loop
 * is submit(k + 1), collect(k) on ONE serving thread (a synchronous nvca_part_batch_process takes one of the two slots while it runs: it is
 * refused while two tickets are outstanding).  A stream may appear in both outstanding calls (not in a synchronous nvca_part_batch_process /
 * nvca_part_stream_process while a ticket of its is outstanding: refused).  A collect that fails rolls its streams back as a
 * failed nvca_part_batch_process does and abandons a newer outstanding ticket with it (rolled back first); a ticket that is never
 * collected is rolled back when the context is destroyed -- or when one of its streams is (nvca_part_stream_destroy gives up every
 * outstanding call of the context first: their tickets become unknown). */
int  nvca_part_batch_submit(nvca_ctx *ctx, int n, nvca_part_stream *const *streams, const nvca_frame *frames, int *ticket);
int  nvca_part_batch_collect(nvca_ctx *ctx, int ticket, nvca_rect *out_a, int cap_a, int *n_a, nvca_rect *out_b, int cap_b, int *n_b);

/* ---- NuboTracker stream -------------------------------------------------
 * Replaces gst_nubo_tracker_img_conf + gst_nubo_tracker_process
 * (TRK/gstnubotracker.cpp:202-237, 339-421): BGRA -> gray, absdiff, threshold,
 * updateMotionHistory, segmentMotion, __join_objects.  img_prev is per stream
 * (the reference's process-global Mat, :108, is a bug). */
typedef struct nvca_tracker_params {
    int    threshold;     /* "set_threshold" 20  (:23) */
    int    min_area;      /* "set_min_area"  50  (:24) */
    long   max_area;      /* "set_max_area"  30000 (:25) */
    int    distance;      /* "set_distance"  35  (:26) */
    double mhi_duration;  /* MHI_DURATION 0.2 (:28) */
    double seg_thresh;    /* SEGMENTATION 32  (:31) */
} nvca_tracker_params;
void nvca_tracker_params_default(nvca_tracker_params *p);
int  nvca_tracker_create(nvca_ctx *ctx, const nvca_tracker_params *params, nvca_tracker **out);
void nvca_tracker_destroy(nvca_tracker *t);
int  nvca_tracker_set_params(nvca_tracker *t, const nvca_tracker_params *params);
/* timestamp_ms: the reference passes 1000*clock()/CLOCKS_PER_SEC (:349) */
int  nvca_tracker_process(nvca_tracker *t, const nvca_frame *frame_bgra, double timestamp_ms,
                          nvca_rect *out, int cap, int *n_out);
int  nvca_tracker_batch_process(nvca_ctx *ctx, int n, nvca_tracker *const *trackers,
                                const nvca_frame *frames, const double *timestamps_ms,
                                nvca_rect *out, int cap, int *n_out);

#ifdef __cplusplus
}
#endif
#endif /* NUBOVCA_H */
