// gstnubovca.cpp -- GStreamer shim: the reference's element surface
// on top of libnubovca_hip's C ABI.  Written from scratch; it keeps, per element,
//   factory name, GstVideoFilter base, in-place transform, caps, property names /
//   ranges / initial values, the string signal and the downstream custom event
// of the reference so that an unmodified pipeline (or Kurento's generic filter with
// "filter-factory" = element name) keeps working:
//   nubofacedetector  FACE/kmsfacedetect.cpp   (caps BGR  :129-133, props :1043-1102, signal "face-event" :1108-1113,
//                                               event "message" :196-226, motion events :680-755)
//   nubotracker       TRK/gstnubotracker.cpp   (caps BGRA :57-61,  props :504-542,  signal "tracker-event" :547-552)
//   nuboeyedetector / nubonosedetector / nubomouthdetector / nuboeardetector
//                     EYE/kmseyedetect.cpp:1274-1320 (props), :220-308 (event: eye_left*, eye_right*), :192-218,680-764 (faces in)
//                     NOSE/kmsnosedetect.cpp:1089-1135, :212-273 (event "noses")   MOUTH/kmsmouthdetect.cpp:205-282 (faces + mouths)
//                     EAR/kmseardetect.cpp:994-1038 ("meta-data"), :196-290 (signal only: the event is built but never pushed)
// All pixel work happens in the library (HIP); this file is glue only.  One plugin
// ("nubovca") registers every factory; the reference ships one plugin per element.
#include <gst/gst.h>
#include <gst/video/video.h>
#include <gst/video/gstvideofilter.h>
#include <sys/time.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <mutex>
#include <atomic>
#include <condition_variable>
#include <map>
#include <vector>
#include <algorithm>
#include <math.h>
#include "nubovca.h"

GST_DEBUG_CATEGORY_STATIC(nubovca_debug);
// No C++ exception crosses into GLib / GStreamer's C frames: every GObject / GstBaseTransform callback of the elements is a
// function-try-block ending in one of these.  A frame whose processing failed is passed on untouched with GST_FLOW_OK -- the
// reference never lets an error out of the element either (FACE/kmsfacedetect.cpp:897).
static void nvca_gst_caught(const char *where) noexcept
{
    try { throw; }
    catch (const std::exception &e) { g_warning("nubovca: %s: %s (frame / request dropped)", where, e.what()); }
    catch (...) { g_warning("nubovca: %s: unknown exception (frame / request dropped)", where); }
}
#define NVCA_GST_CATCH(where, ret) catch (...) { nvca_gst_caught(where); return ret; }
#define NVCA_GST_CATCH_VOID(where) catch (...) { nvca_gst_caught(where); }
// The element mutex inside a guarded callback is held through this: when an exception unwinds the try block the mutex is released
// BEFORE the handler runs -- a streaming thread that caught bad_alloc must not keep the element's GRecMutex locked for good (a
// property set from the application thread, or finalize, would block on it forever).
struct NvcaRecLock {
    GRecMutex *m;
    explicit NvcaRecLock(GRecMutex *mu) : m(mu) { g_rec_mutex_lock(m); }
    ~NvcaRecLock() { g_rec_mutex_unlock(m); }
    NvcaRecLock(const NvcaRecLock &) = delete;
    NvcaRecLock &operator=(const NvcaRecLock &) = delete;
};
#define GST_CAT_DEFAULT nubovca_debug

#define OPENCV_CASCADE_DIR "/usr/share/opencv/haarcascades"     /* FACE/kmsfacedetect.cpp:40 */

// ---------------------------------------------------------------- per-GPU frontend
// A media stream (= one element instance, one streaming thread: FACE/kmsfacedetect.cpp:857-898) is the independent unit: it
// lives on ONE GPU for its whole life, because its temporal state (Faces, the tracker's MHI and previous frame) is device
// resident.  The process holds one slot per visible GPU -- context, cascade handles, frame combiners -- created on first
// use; an element takes slot (instance counter mod slots) at its first frame and never migrates.  No data crosses GPUs.
//   NVCA_DEVICE=d        pin the whole process to GPU d (one slot)
//   NVCA_VIRTUAL_GPUS=n  n slots on one device (NVCA_DEVICE or 0): the multi-GPU code path on a one-GPU box, and a way to
//                        let several launch-bound pipelines queue their work in parallel
struct SharedCascade { nvca_cascade *h; int refs; };
struct GpuSlot;
static std::mutex g_ctx_mutex;
static std::vector<GpuSlot *> g_slots;          // filled once, entries live for the process
static bool g_slots_ready = false;
static int g_instances = 0;                     // elements that have taken a slot so far
// Frames of different elements of one kind that arrive while the GPU is busy are combined into one batched call.  The
// first streaming thread to arrive leads: it takes whatever has queued up (its own frame included), runs the batched
// entry point on the lot and wakes the owners; threads that arrive meanwhile queue up for the next round, so the batch
// grows with the load and an idle pipeline still sees single-frame latency.  Results are those of per-frame calls:
// streams are independent and a batch keeps the order of each stream's frames.
template <class Req> struct Combiner {
    std::mutex m;
    std::condition_variable cv;
    std::vector<Req *> q;
    bool leader = false;
    int max_batch = 0;                                      /* largest batch seen (NVCA_GST_STATS) */
    template <class Run> int process(Req *req, Run run)
    {
        static const bool off = getenv("NVCA_GST_NO_COMBINE") != NULL;     /* A/B switch: every frame on its own */
        if (off) { std::vector<Req *> one(1, req); run(one); return req->rc; }
        std::unique_lock<std::mutex> lk(m);
        req->done = false;
        q.push_back(req);
        while (!req->done) {
            if (leader) { cv.wait(lk); continue; }
            leader = true;
            std::vector<Req *> round;
            round.swap(q);
            lk.unlock();
            run(round);
            lk.lock();
            if ((int)round.size() > max_batch) max_batch = (int)round.size();
            for (Req *r : round) r->done = true;
            leader = false;
            cv.notify_all();
        }
        return req->rc;
    }
};

static const int kFaceCap = 256, kTrkCap = 4096;
struct FaceReq { nvca_face_stream *stream; nvca_frame frame; nvca_rect *out; int n, rc; bool done; };
struct TrkReq { nvca_tracker *trk; nvca_frame frame; double ts; nvca_rect *out; int n, rc; bool done; };
static const int kPartCap = 64;
struct PartReq { nvca_part_stream *stream; nvca_frame frame; nvca_rect *a, *b; int na, nb, rc; bool done; };
struct GpuSlot {
    int index = 0, device = 0;
    nvca_ctx *ctx = nullptr;
    // Elements that name the same cascade file share one handle per GPU: their plans and scale tables are then shared in the
    // context, and their frames can ride in one batch (streams batch only when cascade, geometry and parameters agree).
    std::map<std::string, SharedCascade> cascades;
    Combiner<FaceReq> face_q;
    Combiner<TrkReq> trk_q;
    Combiner<PartReq> part_q;                   // eye / nose / mouth / ear elements of all streams of the slot
    std::atomic<int> registered{0};             // pool memories page-locked so far (NVCA_GST_STATS)
};

static void make_slots_locked()
{
    if (g_slots_ready) return;
    g_slots_ready = true;
    const char *dev = getenv("NVCA_DEVICE"), *virt = getenv("NVCA_VIRTUAL_GPUS");
    std::vector<int> devices;
    if (virt && atoi(virt) > 0) devices.assign((size_t)std::min(atoi(virt), 64), dev ? atoi(dev) : 0);
    else if (dev) devices.push_back(atoi(dev));
    else {
        int n = 0;
        if (nvca_device_count(&n) != NVCA_OK || n <= 0) n = 1;      // ctx creation reports the missing device
        for (int d = 0; d < n; d++) devices.push_back(d);
    }
    for (size_t i = 0; i < devices.size(); i++) { GpuSlot *g = new GpuSlot(); g->index = (int)i; g->device = devices[i]; g_slots.push_back(g); }
}
// NVCA_GST_REGISTER=1: pool memory that keeps coming back (decoders and converters recycle their GstBufferPool memories) is
// page-locked once with nvca_host_register, so its H2D copies are DMA from the buffer instead of going through the runtime's
// staging copy; the registration ends with the GstMemory (weak reference).  One-shot memories are left alone.  Off by
// default: measured on 16 x 1080p branches it gains 3 % when whole frames are copied and LOSES 23 % in the elements' default
// shrink-first mode, where only the rows the resize reads cross PCIe -- a strided DMA out of page-locked memory is slower
// than the runtime's packing of those rows from pageable memory.
struct RegNote { nvca_ctx *ctx; void *ptr; };
static void on_memory_gone(gpointer data, GstMiniObject *)
{
    RegNote *r = (RegNote *)data;
    nvca_host_unregister(r->ctx, r->ptr);
    delete r;
}
static void note_frame_memory(GpuSlot *g, GstVideoFrame *frame)
{
    static const bool off = !(getenv("NVCA_GST_REGISTER") && atoi(getenv("NVCA_GST_REGISTER")) != 0);
    static const GQuark seen_q = g_quark_from_static_string("nubovca-seen");
    GstMemory *mem = frame->map[0].memory;
    if (off || !g || !g->ctx || !mem || !frame->map[0].data || !frame->map[0].size) return;
    GstMiniObject *mo = GST_MINI_OBJECT_CAST(mem);
    const int seen = GPOINTER_TO_INT(gst_mini_object_get_qdata(mo, seen_q));
    if (seen < 0) return;                                   // registered, or registration refused once
    if (seen + 1 < 3) { gst_mini_object_set_qdata(mo, seen_q, GINT_TO_POINTER(seen + 1), NULL); return; }
    gst_mini_object_set_qdata(mo, seen_q, GINT_TO_POINTER(-1), NULL);
    if (nvca_host_register(g->ctx, frame->map[0].data, frame->map[0].size) != NVCA_OK) return;
    gst_mini_object_weak_ref(mo, on_memory_gone, new RegNote{g->ctx, frame->map[0].data});
    g->registered++;
}
// the slot an element lives on: assigned round-robin at its first frame, then fixed
static GpuSlot *take_slot()
{
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    make_slots_locked();
    GpuSlot *g = g_slots[(size_t)(g_instances++) % g_slots.size()];
    if (!g->ctx) {
        const int rc = nvca_ctx_create(g->device, &g->ctx);
        if (rc != NVCA_OK) { GST_ERROR("nvca_ctx_create(%d) failed (%d): no HIP device; frames pass through untouched", g->device, rc); g->ctx = nullptr; }
    }
    if (getenv("NVCA_GST_STATS")) fprintf(stderr, "nubovca: element %d -> slot %d (device %d)\n", g_instances - 1, g->index, g->device);
    return g;
}
static nvca_cascade *acquire_cascade(GpuSlot *g, const std::string &path)
{
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    auto it = g->cascades.find(path);
    if (it != g->cascades.end()) { it->second.refs++; return it->second.h; }
    nvca_cascade *h = nullptr;
    if (nvca_cascade_load_xml(g->ctx, path.c_str(), &h) != NVCA_OK) return nullptr;
    g->cascades[path] = SharedCascade{h, 1};
    return h;
}
static void release_cascade(GpuSlot *g, nvca_cascade *h)
{
    if (!g) return;
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    for (auto it = g->cascades.begin(); it != g->cascades.end(); ++it)
        if (it->second.h == h) {
            if (--it->second.refs == 0) { nvca_cascade_free(h); g->cascades.erase(it); }
            return;
        }
}

static void face_run_round(nvca_ctx *ctx, std::vector<FaceReq *> &round)
{
    const int n = (int)round.size();
    bool batched = false;
    if (n > 1) {
        std::vector<nvca_face_stream *> streams(n);
        std::vector<nvca_frame> frames(n);
        std::vector<nvca_rect> out((size_t)n * kFaceCap);
        std::vector<int> n_out(n, 0);
        for (int i = 0; i < n; i++) { streams[i] = round[i]->stream; frames[i] = round[i]->frame; }
        const int rc = nvca_face_batch_process(ctx, n, streams.data(), frames.data(), out.data(), NULL, kFaceCap, n_out.data());
        batched = rc != NVCA_ERR_ARG;
        for (int i = 0; batched && i < n; i++) {
            FaceReq *r = round[i];
            r->rc = rc; r->n = rc == NVCA_OK ? n_out[i] : 0;
            memcpy(r->out, out.data() + (size_t)i * kFaceCap, sizeof(nvca_rect) * (size_t)std::min(n_out[i], kFaceCap));
        }
    }
    // a single frame, or a batch refused with NVCA_ERR_ARG (every frame is validated before any stream is touched): one
    // bad frame must not fail its neighbours, so each gets its own call and its own status.  Any other failure came
    // after the streams' frame gates advanced and is reported to every owner as it is.
    if (!batched)
        for (FaceReq *r : round) r->rc = nvca_face_stream_process(r->stream, &r->frame, r->out, NULL, kFaceCap, &r->n);
}
static void part_run_round(nvca_ctx *ctx, std::vector<PartReq *> &round)
{
    const int n = (int)round.size();
    bool batched = false;
    if (n > 1) {
        std::vector<nvca_part_stream *> streams(n);
        std::vector<nvca_frame> frames(n);
        std::vector<nvca_rect> a((size_t)n * kPartCap), b((size_t)n * kPartCap);
        std::vector<int> na(n, 0), nb(n, 0);
        for (int i = 0; i < n; i++) { streams[i] = round[i]->stream; frames[i] = round[i]->frame; }
        const int rc = nvca_part_batch_process(ctx, n, streams.data(), frames.data(), a.data(), kPartCap, na.data(), b.data(), kPartCap, nb.data());
        batched = rc == NVCA_OK;                            // any failure leaves every stream of the call as it was (parts.cpp, Rollback):
                                                            // each frame then runs on its own, so one bad frame does not fail its neighbours
        for (int i = 0; batched && i < n; i++) {
            PartReq *r = round[i];
            r->rc = rc; r->na = rc == NVCA_OK ? na[i] : 0; r->nb = rc == NVCA_OK ? nb[i] : 0;
            memcpy(r->a, a.data() + (size_t)i * kPartCap, sizeof(nvca_rect) * (size_t)std::min(r->na, kPartCap));
            memcpy(r->b, b.data() + (size_t)i * kPartCap, sizeof(nvca_rect) * (size_t)std::min(r->nb, kPartCap));
        }
    }
    if (!batched)
        for (PartReq *r : round) r->rc = nvca_part_stream_process(r->stream, &r->frame, r->a, kPartCap, &r->na, r->b, kPartCap, &r->nb);
}

static void trk_run_round(nvca_ctx *ctx, std::vector<TrkReq *> &round)
{
    const int n = (int)round.size();
    bool batched = false;
    if (n > 1) {
        std::vector<nvca_tracker *> trks(n);
        std::vector<nvca_frame> frames(n);
        std::vector<double> ts(n);
        std::vector<nvca_rect> out((size_t)n * kTrkCap);
        std::vector<int> n_out(n, 0);
        for (int i = 0; i < n; i++) { trks[i] = round[i]->trk; frames[i] = round[i]->frame; ts[i] = round[i]->ts; }
        const int rc = nvca_tracker_batch_process(ctx, n, trks.data(), frames.data(), ts.data(), out.data(), kTrkCap, n_out.data());
        batched = rc != NVCA_ERR_ARG;
        for (int i = 0; batched && i < n; i++) {
            TrkReq *r = round[i];
            r->rc = rc; r->n = rc == NVCA_OK ? n_out[i] : 0;
            memcpy(r->out, out.data() + (size_t)i * kTrkCap, sizeof(nvca_rect) * (size_t)std::min(n_out[i], kTrkCap));
        }
    }
    if (!batched)
        for (TrkReq *r : round) r->rc = nvca_tracker_process(r->trk, &r->frame, r->ts, r->out, kTrkCap, &r->n);
}

static std::string cascade_path(const char *file)
{
    const char *dir = getenv("NVCA_CASCADE_DIR");
    return std::string(dir ? dir : OPENCV_CASCADE_DIR) + "/" + file;
}
static double now_ms()
{
    struct timeval t; gettimeofday(&t, NULL);
    return t.tv_sec * 1000.0 + t.tv_usec / 1000.0;
}

// ---------------------------------------------------------------- view-* drawing (host side, on the mapped frame)
// The reference draws on the frame in place when a view property is on: cvRectangle(..., thickness 3, line type 8)
// around every box and cv::circle(..., thickness 4) around the eyes (FACE/BaseFace.cpp:70-82, NOSE :898-905,
// MOUTH :896-903, EAR :750-758, EYE :1069-1099, TRK :389).  Frames are host memory here, so this stays on the host
// ("stays on host"); it is written from the documented geometry of OpenCV's thick primitives -- a 3-pixel
// band centred on each edge with radius-1 round joins, a 4-pixel ring -- and is NOT pixel-verified against OpenCV,
// which is not available offline.  Boxes, events and signals do not depend on it.
// view-* outlines: the library draws them (nvca_draw_shapes, host frames here), so a pipeline that keeps its frames in HBM
// gets the same pixels from the device kernel
struct Canvas { nvca_ctx *ctx; nvca_frame f; int bpp; };
static void draw_shape(const Canvas &c, int kind, int x, int y, int w, int h, const guint8 *col)
{
    // host frames need no context (nvca_draw_shapes): outlines are drawn even where no GPU slot came up
    nvca_shape sh; sh.kind = kind; sh.x = x; sh.y = y; sh.w = w; sh.h = h;
    for (int k = 0; k < 4; k++) sh.bgra[k] = col[k];
    nvca_draw_shapes(c.f.mem == NVCA_MEM_HOST ? NULL : c.ctx, &c.f, c.bpp, &sh, 1);
}
// corners (x0, y0) and (x1, y1) inclusive, as cvRectangle takes them
static void draw_rect3(const Canvas &c, int x0, int y0, int x1, int y1, const guint8 *col) { draw_shape(c, NVCA_SHAPE_RECT3, x0, y0, x1 - x0, y1 - y0, col); }
static void draw_ring4(const Canvas &c, int cx, int cy, int radius, const guint8 *col) { draw_shape(c, NVCA_SHAPE_RING4, cx, cy, radius, 0, col); }
#define BGR_OF_RGB(r, g, b) {b, g, r, 0}
static Canvas canvas_of(nvca_ctx *ctx, GstVideoFrame *frame)
{
    Canvas c;
    c.ctx = ctx;
    c.f.data = GST_VIDEO_FRAME_PLANE_DATA(frame, 0);
    c.f.width = GST_VIDEO_FRAME_WIDTH(frame); c.f.height = GST_VIDEO_FRAME_HEIGHT(frame);
    c.f.stride = GST_VIDEO_FRAME_PLANE_STRIDE(frame, 0); c.f.mem = NVCA_MEM_HOST; c.f.pts = 0;
    c.bpp = GST_VIDEO_FRAME_COMP_PSTRIDE(frame, 0);
    return c;
}

// =====================================================================================
// nubofacedetector
// =====================================================================================
struct NvcaFace {
    GstVideoFilter base;
    GRecMutex mutex;
    GpuSlot *slot;                  /* the GPU this stream lives on (taken at the first frame) */
    nvca_cascade *cascade; nvca_face_stream *stream;
    nvca_face_params p;
    int view_faces, send_meta_data, server_events, events_ms;
    double time_events_ms;
    GQueue *events_queue;
    GstStructure *image_to_overlay;
};
struct NvcaFaceClass { GstVideoFilterClass parent; };
G_DEFINE_TYPE(NvcaFace, nvca_face, GST_TYPE_VIDEO_FILTER)

enum { FP_0, FP_VIEW, FP_DETECT_EVENT, FP_META, FP_WIDTH, FP_X_EVERY_4, FP_EUCLID, FP_TRACK, FP_AREA, FP_SCALE, FP_EVENTS,
       FP_EVENTS_MS, FP_OVERLAY };
static guint face_signal = 0;

static void face_sync_params(NvcaFace *f) { if (f->stream) nvca_face_stream_set_params(f->stream, &f->p); }

static void nvca_face_set_property(GObject *o, guint id, const GValue *v, GParamSpec *ps)
try {
    NvcaFace *f = (NvcaFace *)o;
    NvcaRecLock nvca_lock_(&f->mutex);
    switch (id) {
    case FP_VIEW: f->view_faces = g_value_get_int(v); break;
    case FP_DETECT_EVENT: f->p.detect_event = g_value_get_int(v); break;
    case FP_META: f->send_meta_data = g_value_get_int(v); break;
    case FP_X_EVERY_4: f->p.process_x_every_4 = g_value_get_int(v); break;
    case FP_WIDTH: f->p.width_to_process = g_value_get_int(v); break;
    case FP_SCALE: f->p.scale_factor_pct = g_value_get_int(v); break;
    case FP_EUCLID: f->p.euclidean_threshold = g_value_get_int(v); break;
    case FP_TRACK: f->p.euclidean_threshold = g_value_get_int(v); break;   /* sic: FACE/kmsfacedetect.cpp:548-550 */
    case FP_AREA: f->p.area_threshold = g_value_get_int(v); break;
    case FP_EVENTS: f->server_events = g_value_get_int(v); f->time_events_ms = now_ms(); break;
    case FP_EVENTS_MS: f->events_ms = g_value_get_int(v); break;
    case FP_OVERLAY:
        if (f->image_to_overlay) gst_structure_free(f->image_to_overlay);
        f->image_to_overlay = (GstStructure *)g_value_dup_boxed(v);      /* accepted; overlay drawing is out of scope */
        break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps); break;
    }
    face_sync_params(f);
}
NVCA_GST_CATCH_VOID("nvca_face_set_property")
static void nvca_face_get_property(GObject *o, guint id, GValue *v, GParamSpec *ps)
try {
    NvcaFace *f = (NvcaFace *)o;
    NvcaRecLock nvca_lock_(&f->mutex);
    switch (id) {
    case FP_VIEW: g_value_set_int(v, f->view_faces); break;
    case FP_DETECT_EVENT: g_value_set_int(v, f->p.detect_event); break;
    case FP_META: g_value_set_int(v, f->send_meta_data); break;
    case FP_X_EVERY_4: g_value_set_int(v, f->p.process_x_every_4); break;
    case FP_WIDTH: g_value_set_int(v, f->p.width_to_process); break;
    case FP_SCALE: g_value_set_int(v, f->p.scale_factor_pct); break;
    case FP_EUCLID: g_value_set_int(v, f->p.euclidean_threshold); break;
    case FP_TRACK: g_value_set_int(v, f->p.track_threshold); break;
    case FP_AREA: g_value_set_int(v, f->p.area_threshold); break;
    case FP_EVENTS: g_value_set_int(v, f->server_events); break;
    case FP_EVENTS_MS: g_value_set_int(v, f->events_ms); break;
    case FP_OVERLAY: g_value_set_boxed(v, f->image_to_overlay); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps); break;
    }
}
NVCA_GST_CATCH_VOID("nvca_face_get_property")

// a queued upstream "message": does it carry a "motion" structure?  (FACE/kmsfacedetect.cpp:680-712)
static bool message_has_motion(const GstStructure *m)
{
    const gint len = gst_structure_n_fields(m);
    for (gint i = 0; i < len; i++) {
        const gchar *name = gst_structure_nth_field_name(m, i);
        if (g_strcmp0(name, "motion") == 0 && gst_structure_has_field_typed(m, name, GST_TYPE_STRUCTURE)) return true;
    }
    return false;
}

static gboolean nvca_face_sink_event(GstBaseTransform *trans, GstEvent *event)
try {
    NvcaFace *f = (NvcaFace *)trans;
    if (GST_EVENT_TYPE(event) == GST_EVENT_CUSTOM_DOWNSTREAM) {
        const GstStructure *s = gst_event_get_structure(event);
        if (s) { GST_OBJECT_LOCK(f); g_queue_push_tail(f->events_queue, gst_structure_copy(s)); GST_OBJECT_UNLOCK(f); }
    }
    return GST_BASE_TRANSFORM_CLASS(nvca_face_parent_class)->sink_event(trans, event);
}
NVCA_GST_CATCH("nvca_face_sink_event", FALSE)

static void face_lazy_init(NvcaFace *f)
{
    if (f->stream) return;
    if (!f->slot) f->slot = take_slot();
    nvca_ctx *ctx = f->slot->ctx;
    if (!ctx) return;
    if (!f->cascade) {
        const std::string path = cascade_path("haarcascade_frontalface_alt.xml");
        f->cascade = acquire_cascade(f->slot, path);
        if (!f->cascade) {
            GST_ERROR("Error charging cascade %s: %s", path.c_str(), nvca_last_error(ctx));   /* :167-176: logged, not fatal */
            return;
        }
    }
    if (nvca_face_stream_create(ctx, f->cascade, &f->p, &f->stream) != NVCA_OK) f->stream = nullptr;
}

static GstFlowReturn nvca_face_transform_frame_ip(GstVideoFilter *filter, GstVideoFrame *frame)
try {
    NvcaFace *f = (NvcaFace *)filter;
    NvcaRecLock nvca_lock_(&f->mutex);
    face_lazy_init(f);
    if (f->stream && f->p.width_to_process > 0) {
        // upstream "motion" messages arm the detector when detect-event = 1
        GST_OBJECT_LOCK(f);
        while (GstStructure *m = (GstStructure *)g_queue_pop_head(f->events_queue)) {
            if (f->p.detect_event && message_has_motion(m)) nvca_face_stream_motion_event(f->stream);
            gst_structure_free(m);
        }
        GST_OBJECT_UNLOCK(f);
        note_frame_memory(f->slot, frame);
        nvca_frame nf;
        nf.data = GST_VIDEO_FRAME_PLANE_DATA(frame, 0);
        nf.width = GST_VIDEO_FRAME_WIDTH(frame); nf.height = GST_VIDEO_FRAME_HEIGHT(frame);
        nf.stride = GST_VIDEO_FRAME_PLANE_STRIDE(frame, 0);
        nf.mem = NVCA_MEM_HOST; nf.pts = GST_BUFFER_PTS(frame->buffer);
        nvca_rect boxes[256]; int n = 0;
        FaceReq req{f->stream, nf, boxes, 0, NVCA_OK, false};
        nvca_ctx *ctx = f->slot->ctx;
        const int rc = f->slot->face_q.process(&req, [ctx](std::vector<FaceReq *> &round) { face_run_round(ctx, round); });
        n = req.n;
        if (rc != NVCA_OK) GST_ERROR("nvca_face_stream_process: %d", rc);
        else {
            if (n > 256) n = 256;
            // kms_face_send_event :179-249
            GstStructure *message = gst_structure_new_empty("message");
            GstStructure *ts = gst_structure_new("time", "pts", G_TYPE_UINT64, GST_BUFFER_PTS(frame->buffer), NULL);
            gst_structure_set(message, "timestamp", GST_TYPE_STRUCTURE, ts, NULL);
            gst_structure_free(ts);
            std::string faces_str;
            for (int i = 0; i < n; i++) {
                GstStructure *face = gst_structure_new("face", "type", G_TYPE_STRING, "face", "x", G_TYPE_UINT, (guint)boxes[i].x,
                                                       "y", G_TYPE_UINT, (guint)boxes[i].y, "width", G_TYPE_UINT, (guint)boxes[i].w,
                                                       "height", G_TYPE_UINT, (guint)boxes[i].h, NULL);
                char id[16]; snprintf(id, sizeof(id), "%d", i);
                gst_structure_set(message, id, GST_TYPE_STRUCTURE, face, NULL);
                gst_structure_free(face);
                faces_str += "x:" + std::to_string((guint)boxes[i].x) + ",y:" + std::to_string((guint)boxes[i].y) +
                             ",width:" + std::to_string((guint)boxes[i].w) + ",height:" + std::to_string((guint)boxes[i].h) + ";";
            }
            gst_pad_push_event(GST_BASE_TRANSFORM(f)->srcpad, gst_event_new_custom(GST_EVENT_CUSTOM_DOWNSTREAM, message));
            if (f->view_faces > 0 && !f->image_to_overlay) {
                // Faces::draw: (x, y) .. (x + w - 1, y + h - 1) of the working-image box, times the integer scale, in
                // colors[1] = CV_RGB(0,128,255)  (FACE/BaseFace.cpp:70-82, FACE/kmsfacedetect.cpp:144-151,832-850)
                static const guint8 col[4] = BGR_OF_RGB(0, 128, 255);
                const Canvas cv = canvas_of(f->slot->ctx, frame);
                const int scale = nf.width / f->p.width_to_process;
                for (int i = 0; i < n; i++)
                    draw_rect3(cv, boxes[i].x, boxes[i].y, boxes[i].x + boxes[i].w - scale, boxes[i].y + boxes[i].h - scale, col);
            }
            if (n > 0) {
                const double t = now_ms();
                if (1 == f->server_events && t - f->time_events_ms > f->events_ms) {
                    f->time_events_ms = t;
                    g_signal_emit(G_OBJECT(f), face_signal, 0, faces_str.c_str());
                }
            }
        }
    }
    return GST_FLOW_OK;                                     /* always: FACE/kmsfacedetect.cpp:897 */
}
NVCA_GST_CATCH("nvca_face_transform_frame_ip", GST_FLOW_OK)

static void nvca_face_finalize(GObject *o)
try {
    NvcaFace *f = (NvcaFace *)o;
    if (getenv("NVCA_GST_STATS") && f->slot) {
        std::lock_guard<std::mutex> lk(f->slot->face_q.m);
        fprintf(stderr, "nubovca: largest combined face batch %d (slot %d)\n", f->slot->face_q.max_batch, f->slot->index);
        fprintf(stderr, "nubovca: page-locked pool memories %d (slot %d)\n", f->slot->registered.load(), f->slot->index);
    }
    if (f->stream) nvca_face_stream_destroy(f->stream);
    if (f->cascade) release_cascade(f->slot, f->cascade);
    if (f->image_to_overlay) gst_structure_free(f->image_to_overlay);
    g_queue_free_full(f->events_queue, (GDestroyNotify)gst_structure_free);
    g_rec_mutex_clear(&f->mutex);
    G_OBJECT_CLASS(nvca_face_parent_class)->finalize(o);
}
NVCA_GST_CATCH_VOID("nvca_face_finalize")

static void nvca_face_init(NvcaFace *f)
try {
    nvca_face_params_default(&f->p);                        /* defaults of FACE/kmsfacedetect.cpp:979-999 */
    f->view_faces = 0; f->send_meta_data = 0; f->server_events = 0; f->events_ms = 30001; f->time_events_ms = 0;
    f->cascade = nullptr; f->stream = nullptr; f->image_to_overlay = nullptr;
    f->events_queue = g_queue_new();
    g_rec_mutex_init(&f->mutex);
}
NVCA_GST_CATCH_VOID("nvca_face_init")

#define INT_PROP(klass, id, name, nick, blurb, lo, hi) \
    g_object_class_install_property(klass, id, g_param_spec_int(name, nick, blurb, lo, hi, 0, (GParamFlags)G_PARAM_READWRITE))

static void nvca_face_class_init(NvcaFaceClass *klass)
{
    GObjectClass *go = G_OBJECT_CLASS(klass);
    GstVideoFilterClass *vf = GST_VIDEO_FILTER_CLASS(klass);
    GstCaps *caps = gst_caps_from_string(GST_VIDEO_CAPS_MAKE("{ BGR }"));
    gst_element_class_add_pad_template(GST_ELEMENT_CLASS(klass), gst_pad_template_new("src", GST_PAD_SRC, GST_PAD_ALWAYS, caps));
    gst_element_class_add_pad_template(GST_ELEMENT_CLASS(klass), gst_pad_template_new("sink", GST_PAD_SINK, GST_PAD_ALWAYS, caps));
    gst_caps_unref(caps);
    gst_element_class_set_static_metadata(GST_ELEMENT_CLASS(klass), "face detection filter element", "Video/Filter",
                                          "Haar face detector (MI355X / HIP implementation of NuboFaceDetector)", "nubovca-hip");
    go->set_property = nvca_face_set_property; go->get_property = nvca_face_get_property; go->finalize = nvca_face_finalize;
    INT_PROP(go, FP_VIEW, "view-faces", "view faces", "draw or hide the detected faces on the stream", 0, 1);
    INT_PROP(go, FP_DETECT_EVENT, "detect-event", "detect event", "0 => always; 1 => only after a motion event", 0, 1);
    INT_PROP(go, FP_META, "send-meta-data", "send meta data", "0 (default) => no meta data; 1 => send the face boxes as metadata", 0, 1);
    INT_PROP(go, FP_WIDTH, "width-to-process", "width to process", "width of the image the algorithm processes", 0, 640);
    INT_PROP(go, FP_X_EVERY_4, "process-x-every-4-frames", "process x every 4 frames", "1,2,3,4 (default)", 0, 4);
    INT_PROP(go, FP_EUCLID, "euclidean-distance", "euclidean distance", "0 - 20 (8 default)", 0, 20);
    INT_PROP(go, FP_TRACK, "track-threshold", "track threshold", "0 - 100 (30 default)", 0, 100);
    INT_PROP(go, FP_AREA, "area-threshold", "area threshold", "0 - 1000 (500 default)", 0, 1000);
    INT_PROP(go, FP_SCALE, "multi-scale-factor", "multi scale factor", "5-50 (25 default)", 0, 51);
    INT_PROP(go, FP_EVENTS, "activate-events", "Activate Events", "0 (default) => no events to the server; 1 => send events", 0, 1);
    INT_PROP(go, FP_EVENTS_MS, "events-ms", "Activate Events", "the time, it takes to send events to the servers", 0, 30000);
    g_object_class_install_property(go, FP_OVERLAY, g_param_spec_boxed("image-to-overlay", "image to overlay",
                                    "set the url of the image to overlay the faces", GST_TYPE_STRUCTURE,
                                    (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS)));
    vf->transform_frame_ip = GST_DEBUG_FUNCPTR(nvca_face_transform_frame_ip);
    GST_BASE_TRANSFORM_CLASS(klass)->sink_event = GST_DEBUG_FUNCPTR(nvca_face_sink_event);
    face_signal = g_signal_new("face-event", G_TYPE_FROM_CLASS(klass), G_SIGNAL_RUN_LAST, 0, NULL, NULL, NULL, G_TYPE_NONE, 1,
                               G_TYPE_STRING);
}

// =====================================================================================
// nubotracker
// =====================================================================================
struct NvcaTrk {
    GstVideoFilter base;
    GRecMutex mutex;
    GpuSlot *slot;
    nvca_tracker *trk;
    nvca_tracker_params p;
    int visual_mode, server_events, events_ms;
    double time_events_ms;
};
struct NvcaTrkClass { GstVideoFilterClass parent; };
G_DEFINE_TYPE(NvcaTrk, nvca_trk, GST_TYPE_VIDEO_FILTER)
enum { TP_0, TP_THRESHOLD, TP_MIN_AREA, TP_MAX_AREA, TP_DISTANCE, TP_VISUAL, TP_EVENTS, TP_EVENTS_MS };
static guint trk_signal = 0;

static void nvca_trk_set_property(GObject *o, guint id, const GValue *v, GParamSpec *ps)
try {
    NvcaTrk *t = (NvcaTrk *)o;
    NvcaRecLock nvca_lock_(&t->mutex);
    switch (id) {
    case TP_THRESHOLD: t->p.threshold = g_value_get_int(v); break;
    case TP_MIN_AREA: t->p.min_area = g_value_get_int(v); break;
    case TP_MAX_AREA: t->p.max_area = g_value_get_long(v); break;
    case TP_DISTANCE: t->p.distance = g_value_get_int(v); break;
    case TP_VISUAL: t->visual_mode = g_value_get_int(v); break;
    case TP_EVENTS: t->server_events = g_value_get_int(v); t->time_events_ms = now_ms(); break;
    case TP_EVENTS_MS: t->events_ms = g_value_get_int(v); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps); break;
    }
    if (t->trk) nvca_tracker_set_params(t->trk, &t->p);
}
NVCA_GST_CATCH_VOID("nvca_trk_set_property")
static void nvca_trk_get_property(GObject *o, guint id, GValue *v, GParamSpec *ps)
try {
    NvcaTrk *t = (NvcaTrk *)o;
    NvcaRecLock nvca_lock_(&t->mutex);
    switch (id) {
    case TP_THRESHOLD: g_value_set_int(v, t->p.threshold); break;
    case TP_MIN_AREA: g_value_set_int(v, t->p.min_area); break;
    case TP_MAX_AREA: g_value_set_long(v, t->p.max_area); break;
    case TP_DISTANCE: g_value_set_int(v, t->p.distance); break;
    case TP_VISUAL: g_value_set_int(v, t->visual_mode); break;
    case TP_EVENTS: g_value_set_int(v, t->server_events); break;
    case TP_EVENTS_MS: g_value_set_int(v, t->events_ms); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps); break;
    }
}
NVCA_GST_CATCH_VOID("nvca_trk_get_property")

static GstFlowReturn nvca_trk_transform_frame_ip(GstVideoFilter *filter, GstVideoFrame *frame)
try {
    NvcaTrk *t = (NvcaTrk *)filter;
    NvcaRecLock nvca_lock_(&t->mutex);
    if (!t->trk) { if (!t->slot) t->slot = take_slot(); nvca_ctx *ctx = t->slot->ctx; if (ctx && nvca_tracker_create(ctx, &t->p, &t->trk) != NVCA_OK) t->trk = nullptr; }
    if (t->trk) {
        note_frame_memory(t->slot, frame);
        nvca_frame nf;
        nf.data = GST_VIDEO_FRAME_PLANE_DATA(frame, 0);
        nf.width = GST_VIDEO_FRAME_WIDTH(frame); nf.height = GST_VIDEO_FRAME_HEIGHT(frame);
        nf.stride = GST_VIDEO_FRAME_PLANE_STRIDE(frame, 0);
        nf.mem = NVCA_MEM_HOST; nf.pts = GST_BUFFER_PTS(frame->buffer);
        const double timestamp = 1000.0 * clock() / CLOCKS_PER_SEC;          /* TRK/gstnubotracker.cpp:349 */
        int n = 0;
        nvca_rect *bx = (nvca_rect *)g_malloc(sizeof(nvca_rect) * kTrkCap);
        TrkReq req{t->trk, nf, timestamp, bx, 0, NVCA_OK, false};
        nvca_ctx *ctx = t->slot->ctx;
        const int rc = t->slot->trk_q.process(&req, [ctx](std::vector<TrkReq *> &round) { trk_run_round(ctx, round); });
        n = req.n;
        if (rc != NVCA_OK) GST_ERROR("nvca_tracker_process: %d", rc);
        else if (n > 0) {
            if (n > 4096) n = 4096;
            if (t->visual_mode > 0) {                        /* TRK/gstnubotracker.cpp:389: tl() .. br(), Scalar(0,0,255) */
                static const guint8 col[4] = {0, 0, 255, 0};
                const Canvas cv = canvas_of(t->slot->ctx, frame);
                for (int i = 0; i < n; i++) draw_rect3(cv, bx[i].x, bx[i].y, bx[i].x + bx[i].w, bx[i].y + bx[i].h, col);
            }
            std::string s;
            if (1 == t->server_events)
                for (int i = 0; i < n; i++)
                    s += "x:" + std::to_string((guint)bx[i].x) + ",y:" + std::to_string((guint)bx[i].y) + ",width:" +
                         std::to_string((guint)bx[i].w) + ",height:" + std::to_string((guint)bx[i].h) + ";";
            const double now = now_ms();
            if (1 == t->server_events && now - t->time_events_ms > t->events_ms) {   /* :402-414 */
                t->time_events_ms = now;
                g_signal_emit(G_OBJECT(t), trk_signal, 0, s.c_str());
            }
        }
        g_free(bx);
    }
    return GST_FLOW_OK;
}
NVCA_GST_CATCH("nvca_trk_transform_frame_ip", GST_FLOW_OK)

static void nvca_trk_finalize(GObject *o)
try {
    NvcaTrk *t = (NvcaTrk *)o;
    if (getenv("NVCA_GST_STATS") && t->slot) {
        std::lock_guard<std::mutex> lk(t->slot->trk_q.m);
        fprintf(stderr, "nubovca: largest combined tracker batch %d (slot %d)\n", t->slot->trk_q.max_batch, t->slot->index);
    }
    if (t->trk) nvca_tracker_destroy(t->trk);
    g_rec_mutex_clear(&t->mutex);
    G_OBJECT_CLASS(nvca_trk_parent_class)->finalize(o);
}
NVCA_GST_CATCH_VOID("nvca_trk_finalize")
static void nvca_trk_init(NvcaTrk *t)
try {
    nvca_tracker_params_default(&t->p);                     /* TRK/gstnubotracker.cpp:457-466 */
    t->visual_mode = 0; t->server_events = 0; t->events_ms = 30001; t->time_events_ms = 0; t->trk = nullptr;
    g_rec_mutex_init(&t->mutex);
}
NVCA_GST_CATCH_VOID("nvca_trk_init")
static void nvca_trk_class_init(NvcaTrkClass *klass)
{
    GObjectClass *go = G_OBJECT_CLASS(klass);
    GstCaps *caps = gst_caps_from_string(GST_VIDEO_CAPS_MAKE("{ BGRA }"));
    gst_element_class_add_pad_template(GST_ELEMENT_CLASS(klass), gst_pad_template_new("src", GST_PAD_SRC, GST_PAD_ALWAYS, caps));
    gst_element_class_add_pad_template(GST_ELEMENT_CLASS(klass), gst_pad_template_new("sink", GST_PAD_SINK, GST_PAD_ALWAYS, caps));
    gst_caps_unref(caps);
    gst_element_class_set_static_metadata(GST_ELEMENT_CLASS(klass), "motion tracker filter element", "Video/Filter",
                                          "Motion-history tracker (MI355X / HIP implementation of NuboTracker)", "nubovca-hip");
    go->set_property = nvca_trk_set_property; go->get_property = nvca_trk_get_property; go->finalize = nvca_trk_finalize;
    INT_PROP(go, TP_THRESHOLD, "set_threshold", "threshold", "motion threshold (20 default)", 0, 255);
    INT_PROP(go, TP_MIN_AREA, "set_min_area", "min area", "minimum object area (50 default)", 0, 10000);
    g_object_class_install_property(go, TP_MAX_AREA, g_param_spec_long("set_max_area", "max area", "maximum object area (30000 default)",
                                                                       0, 300000, 0, (GParamFlags)G_PARAM_READWRITE));
    INT_PROP(go, TP_DISTANCE, "set_distance", "distance", "merge distance (35 default)", 0, 2000);
    INT_PROP(go, TP_VISUAL, "set_visual_mode", "visual mode", "draw the objects (0 default)", 0, 4);
    INT_PROP(go, TP_EVENTS, "activate-events", "Activate Events", "0 (default) => no events to the server", 0, 1);
    INT_PROP(go, TP_EVENTS_MS, "events-ms", "Activate Events", "the time, it takes to send events to the servers", 0, 30000);
    GST_VIDEO_FILTER_CLASS(klass)->transform_frame_ip = GST_DEBUG_FUNCPTR(nvca_trk_transform_frame_ip);
    trk_signal = g_signal_new("tracker-event", G_TYPE_FROM_CLASS(klass), G_SIGNAL_RUN_LAST, 0, NULL, NULL, NULL, G_TYPE_NONE, 1,
                              G_TYPE_STRING);
}


// =====================================================================================
// nuboeyedetector / nubonosedetector / nubomouthdetector / nuboeardetector  (one implementation, four GTypes)
// =====================================================================================
struct PartDesc {
    int kind; const char *factory, *type_name, *view_prop, *meta_prop, *signal, *face_file, *a_file, *b_file, *longname;
    guint signal_id; gpointer parent_class;
};
static PartDesc part_descs[4] = {
    {NVCA_PART_EYE, "nuboeyedetector", "NvcaEyeDetect", "view-eyes", "send-meta-data", "eye-event", "haarcascade_frontalface_alt.xml",
     "haarcascade_mcs_righteye.xml", "haarcascade_mcs_lefteye.xml", "eye detection filter element", 0, NULL},
    {NVCA_PART_NOSE, "nubonosedetector", "NvcaNoseDetect", "view-noses", "send-meta-data", "nose-event", "haarcascade_frontalface_alt.xml",
     "haarcascade_mcs_nose.xml", NULL, "nose detection filter element", 0, NULL},
    {NVCA_PART_MOUTH, "nubomouthdetector", "NvcaMouthDetect", "view-mouths", "send-meta-data", "mouth-event", "haarcascade_frontalface_alt.xml",
     "haarcascade_mcs_mouth.xml", NULL, "mouth detection filter element", 0, NULL},
    /* EAR: LEFT_SIDE uses mcs_rightear.xml, RIGHT_SIDE mcs_leftear.xml (sic, EAR/kmseardetect.cpp:29-31,796,801); property "meta-data" */
    {NVCA_PART_EAR, "nuboeardetector", "NvcaEarDetect", "view-ears", "meta-data", "ear-event", "haarcascade_profileface.xml",
     "haarcascade_mcs_rightear.xml", "haarcascade_mcs_leftear.xml", "ear detection filter element", 0, NULL},
};
struct NvcaPart {
    GstVideoFilter base;
    GRecMutex mutex;
    PartDesc *desc;
    GpuSlot *slot;
    nvca_cascade *cf, *ca, *cb; nvca_part_stream *stream;
    nvca_part_params p;
    int view, meta_data, server_events, events_ms;
    double time_events_ms;
    GstStructure *image_to_overlay;
};
struct NvcaPartClass { GstVideoFilterClass parent; PartDesc *desc; };
enum { PP_0, PP_VIEW, PP_DETECT_EVENT, PP_META, PP_WIDTH, PP_X_EVERY_4, PP_SCALE, PP_EVENTS, PP_EVENTS_MS, PP_OVERLAY };

static void nvca_part_set_property(GObject *o, guint id, const GValue *v, GParamSpec *ps)
try {
    NvcaPart *f = (NvcaPart *)o;
    NvcaRecLock nvca_lock_(&f->mutex);
    switch (id) {
    case PP_VIEW: f->view = g_value_get_int(v); break;
    case PP_DETECT_EVENT: f->p.detect_event = g_value_get_int(v); break;
    case PP_META: f->meta_data = g_value_get_int(v); break;
    case PP_WIDTH: f->p.width_to_process = g_value_get_int(v); break;
    case PP_X_EVERY_4: f->p.process_x_every_4 = g_value_get_int(v); break;
    case PP_SCALE: f->p.scale_factor_pct = g_value_get_int(v); break;
    case PP_EVENTS: f->server_events = g_value_get_int(v); f->time_events_ms = now_ms(); break;
    case PP_EVENTS_MS: f->events_ms = g_value_get_int(v); break;
    case PP_OVERLAY:
        if (f->image_to_overlay) gst_structure_free(f->image_to_overlay);
        f->image_to_overlay = (GstStructure *)g_value_dup_boxed(v);
        break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps); break;
    }
    if (f->stream) nvca_part_stream_set_params(f->stream, &f->p);
}
NVCA_GST_CATCH_VOID("nvca_part_set_property")
static void nvca_part_get_property(GObject *o, guint id, GValue *v, GParamSpec *ps)
try {
    NvcaPart *f = (NvcaPart *)o;
    NvcaRecLock nvca_lock_(&f->mutex);
    switch (id) {
    case PP_VIEW: g_value_set_int(v, f->view); break;
    case PP_DETECT_EVENT: g_value_set_int(v, f->p.detect_event); break;
    case PP_META: g_value_set_int(v, f->meta_data); break;
    case PP_WIDTH: g_value_set_int(v, f->p.width_to_process); break;
    case PP_X_EVERY_4: g_value_set_int(v, f->p.process_x_every_4); break;
    case PP_SCALE: g_value_set_int(v, f->p.scale_factor_pct); break;
    case PP_EVENTS: g_value_set_int(v, f->server_events); break;
    case PP_EVENTS_MS: g_value_set_int(v, f->events_ms); break;
    case PP_OVERLAY: g_value_set_boxed(v, f->image_to_overlay); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps); break;
    }
}
NVCA_GST_CATCH_VOID("nvca_part_get_property")

static void part_lazy_init(NvcaPart *f)
{
    if (f->stream) return;
    if (!f->slot) f->slot = take_slot();
    nvca_ctx *ctx = f->slot->ctx;
    if (!ctx) return;
    struct { nvca_cascade **c; const char *file; } need[3] = {{&f->cf, f->desc->face_file}, {&f->ca, f->desc->a_file}, {&f->cb, f->desc->b_file}};
    for (auto &n : need) {
        if (!n.file || *n.c) continue;
        const std::string path = cascade_path(n.file);
        *n.c = acquire_cascade(f->slot, path);
        if (!*n.c) {
            GST_ERROR("Error charging cascade %s: %s", path.c_str(), nvca_last_error(ctx));
            return;
        }
    }
    if (nvca_part_stream_create(ctx, &f->p, f->cf, f->ca, f->cb, &f->stream) != NVCA_OK) f->stream = nullptr;
}

// faces handed over by an upstream element (EYE/kmseyedetect.cpp:680-724): sub-structures whose type == "face"
static gboolean nvca_part_sink_event(GstBaseTransform *trans, GstEvent *event)
try {
    NvcaPart *f = (NvcaPart *)trans;
    if (GST_EVENT_TYPE(event) == GST_EVENT_CUSTOM_DOWNSTREAM && f->desc->kind != NVCA_PART_EAR) {
        const GstStructure *m = gst_event_get_structure(event);
        NvcaRecLock nvca_lock_(&f->mutex);
        if (m && f->p.detect_event) {
            part_lazy_init(f);
            nvca_rect faces[64]; int n = 0;
            const gint len = gst_structure_n_fields(m);
            for (gint i = 0; i < len && n < 64; i++) {
                const gchar *name = gst_structure_nth_field_name(m, i);
                if (g_strcmp0(name, "timestamp") == 0) continue;
                GstStructure *d = NULL;
                if (!gst_structure_get(m, name, GST_TYPE_STRUCTURE, &d, NULL) || !d) continue;
                const gchar *type = gst_structure_get_string(d, "type");
                if (g_strcmp0(type, "face") == 0) {
                    guint x = 0, y = 0, w = 0, h = 0;
                    gst_structure_get(d, "x", G_TYPE_UINT, &x, "y", G_TYPE_UINT, &y, "width", G_TYPE_UINT, &w, "height", G_TYPE_UINT, &h, NULL);
                    faces[n++] = nvca_rect{(int)x, (int)y, (int)w, (int)h};
                }
                gst_structure_free(d);
            }
            if (f->stream) nvca_part_stream_push_faces(f->stream, faces, n);
        }
    }
    return GST_BASE_TRANSFORM_CLASS(f->desc->parent_class)->sink_event(trans, event);
}
NVCA_GST_CATCH("nvca_part_sink_event", FALSE)

static void add_box(GstStructure *message, const char *sname, const char *type, int idx, const nvca_rect &r, int mul)
{
    GstStructure *st = gst_structure_new(sname, "type", G_TYPE_STRING, type, "x", G_TYPE_UINT, (guint)r.x * mul, "y", G_TYPE_UINT,
                                         (guint)r.y * mul, "width", G_TYPE_UINT, (guint)r.w * mul, "height", G_TYPE_UINT, (guint)r.h * mul, NULL);
    char id[16]; snprintf(id, sizeof(id), "%d", idx);
    gst_structure_set(message, id, GST_TYPE_STRUCTURE, st, NULL);
    gst_structure_free(st);
}
static std::string box_str(const nvca_rect &r)
{
    return "x:" + std::to_string((guint)r.x) + ",y:" + std::to_string((guint)r.y) + ",width:" + std::to_string((guint)r.w) +
           ",height:" + std::to_string((guint)r.h) + ";";
}

static GstFlowReturn nvca_part_transform_frame_ip(GstVideoFilter *filter, GstVideoFrame *frame)
try {
    NvcaPart *f = (NvcaPart *)filter;
    NvcaRecLock nvca_lock_(&f->mutex);
    part_lazy_init(f);
    if (f->stream && f->p.width_to_process > 0) {
        note_frame_memory(f->slot, frame);
        nvca_frame nf;
        nf.data = GST_VIDEO_FRAME_PLANE_DATA(frame, 0);
        nf.width = GST_VIDEO_FRAME_WIDTH(frame); nf.height = GST_VIDEO_FRAME_HEIGHT(frame);
        nf.stride = GST_VIDEO_FRAME_PLANE_STRIDE(frame, 0);
        nf.mem = NVCA_MEM_HOST; nf.pts = GST_BUFFER_PTS(frame->buffer);
        nvca_rect a[64], b[64], faces[64]; int na = 0, nb = 0, nfaces = 0;
        // part detectors of other streams (and other kinds) that arrive meanwhile share the call: nvca_part_batch_process
        PartReq req{f->stream, nf, a, b, 0, 0, NVCA_OK, false};
        nvca_ctx *ctx = f->slot->ctx;
        const int rc = f->slot->part_q.process(&req, [ctx](std::vector<PartReq *> &round) { part_run_round(ctx, round); });
        na = req.na; nb = req.nb;
        if (rc != NVCA_OK) GST_ERROR("nvca_part_stream_process: %d", rc);
        else {
            na = MIN(na, 64); nb = MIN(nb, 64);
            nvca_part_stream_faces(f->stream, faces, 64, &nfaces); nfaces = MIN(nfaces, 64);
            const int kind = f->desc->kind;
            std::string str; int i = 0;
            GstStructure *message = gst_structure_new_empty(kind == NVCA_PART_NOSE ? "noses" : "message");
            if (kind != NVCA_PART_NOSE) {
                GstStructure *ts = gst_structure_new("time", "pts", G_TYPE_UINT64, GST_BUFFER_PTS(frame->buffer), NULL);
                gst_structure_set(message, "timestamp", GST_TYPE_STRUCTURE, ts, NULL);
                gst_structure_free(ts);
            }
            if (kind == NVCA_PART_EYE) {                 // left eyes first, then right eyes (EYE :245-284)
                for (int k = 0; k < nb; k++, i++) { add_box(message, "eye_left", "eye", i, b[k], 1); str += box_str(b[k]); }
                for (int k = 0; k < na; k++, i++) { add_box(message, "eye_right", "eye", i, a[k], 1); str += box_str(a[k]); }
            } else if (kind == NVCA_PART_NOSE) {
                for (int k = 0; k < na; k++, i++) { add_box(message, "noses", "nose", i, a[k], 1); str += box_str(a[k]); }
            } else if (kind == NVCA_PART_MOUTH) {        // faces x norm_faces (int of scale_o2f), then mouths (MOUTH :221-255)
                const int norm_faces = f->p.detect_event ? 1 : (int)(((float)nf.width) / ((float)160));
                for (int k = 0; k < nfaces; k++, i++) add_box(message, "face", "face", i, faces[k], norm_faces);
                for (int k = 0; k < na; k++, i++) { add_box(message, "mouth", "mouth", i, a[k], 1); str += box_str(a[k]); }
            } else {                                     // EAR: right list then left list; the event is never pushed (:196-290)
                for (int k = 0; k < nb; k++) str += box_str(b[k]);
                for (int k = 0; k < na; k++) str += box_str(a[k]);
            }
            if (1 == f->view && !f->image_to_overlay) {
                const Canvas cv = canvas_of(f->slot->ctx, frame);
                if (kind == NVCA_PART_EYE) {                 // one circle per side, first box only (EYE :1069-1099)
                    static const guint8 col[4] = {255, 0, 0, 0};              /* Scalar(255, 0, 0) */
                    int radius = -1;
                    if (na > 0) {
                        radius = (int)lrint((a[0].w + a[0].h) * 0.25);
                        draw_ring4(cv, a[0].x + a[0].w / 2, a[0].y + a[0].h / 2, radius, col);
                    }
                    if (nb > 0) {
                        if (radius < 0) radius = (int)lrint((b[0].w + b[0].h) * 0.25);
                        draw_ring4(cv, b[0].x + b[0].w / 2, b[0].y + b[0].h / 2, radius, col);
                    }
                } else {
                    // palettes and corner arithmetic as written in NOSE :808-815,902-905 (x + w, y + h - 1),
                    // MOUTH :814-821,900-903 (x + w - 1, y + h - 1), EAR :736-743,754-758 (x + w, y + h - 1)
                    static const guint8 nose_col[8][4] = {BGR_OF_RGB(255, 0, 255), BGR_OF_RGB(255, 0, 0), BGR_OF_RGB(255, 255, 0), BGR_OF_RGB(255, 128, 0),
                                                          BGR_OF_RGB(0, 255, 0), BGR_OF_RGB(0, 255, 255), BGR_OF_RGB(0, 128, 255), BGR_OF_RGB(0, 0, 255)};
                    static const guint8 mouth_col[8][4] = {BGR_OF_RGB(255, 255, 0), BGR_OF_RGB(255, 128, 0), BGR_OF_RGB(255, 0, 0), BGR_OF_RGB(255, 0, 255),
                                                           BGR_OF_RGB(0, 128, 255), BGR_OF_RGB(0, 0, 255), BGR_OF_RGB(0, 255, 255), BGR_OF_RGB(0, 255, 0)};
                    static const guint8 ear_col[8][4] = {BGR_OF_RGB(0, 0, 255), BGR_OF_RGB(0, 128, 255), BGR_OF_RGB(0, 255, 255), BGR_OF_RGB(0, 255, 0),
                                                         BGR_OF_RGB(255, 128, 0), BGR_OF_RGB(255, 255, 0), BGR_OF_RGB(255, 0, 0), BGR_OF_RGB(255, 0, 255)};
                    const guint8 (*pal)[4] = kind == NVCA_PART_NOSE ? nose_col : kind == NVCA_PART_MOUTH ? mouth_col : ear_col;
                    const int dx = kind == NVCA_PART_MOUTH ? -1 : 0;
                    int j = 0;
                    if (kind == NVCA_PART_EAR) for (int k = 0; k < nb; k++, j++) draw_rect3(cv, b[k].x, b[k].y, b[k].x + b[k].w + dx, b[k].y + b[k].h - 1, pal[j % 8]);
                    for (int k = 0; k < na; k++, j++) draw_rect3(cv, a[k].x, a[k].y, a[k].x + a[k].w + dx, a[k].y + a[k].h - 1, pal[j % 8]);
                }
            }
            if (kind == NVCA_PART_EAR) gst_structure_free(message);
            else gst_pad_push_event(GST_BASE_TRANSFORM(f)->srcpad, gst_event_new_custom(GST_EVENT_CUSTOM_DOWNSTREAM, message));
            if (na > 0 || nb > 0) {
                const double t = now_ms();
                if (1 == f->server_events && t - f->time_events_ms > f->events_ms) {
                    f->time_events_ms = t;
                    g_signal_emit(G_OBJECT(f), f->desc->signal_id, 0, str.c_str());
                }
            }
        }
    }
    return GST_FLOW_OK;
}
NVCA_GST_CATCH("nvca_part_transform_frame_ip", GST_FLOW_OK)

static void nvca_part_finalize(GObject *o)
try {
    NvcaPart *f = (NvcaPart *)o;
    if (getenv("NVCA_GST_STATS") && f->slot) {
        std::lock_guard<std::mutex> lk(f->slot->part_q.m);
        fprintf(stderr, "nubovca: largest combined part-detector batch %d (slot %d)\n", f->slot->part_q.max_batch, f->slot->index);
    }
    if (f->stream) nvca_part_stream_destroy(f->stream);
    if (f->cf) release_cascade(f->slot, f->cf);
    if (f->ca) release_cascade(f->slot, f->ca);
    if (f->cb) release_cascade(f->slot, f->cb);
    if (f->image_to_overlay) gst_structure_free(f->image_to_overlay);
    g_rec_mutex_clear(&f->mutex);
    G_OBJECT_CLASS(f->desc->parent_class)->finalize(o);
}
NVCA_GST_CATCH_VOID("nvca_part_finalize")
static void nvca_part_instance_init(GTypeInstance *inst, gpointer klass)
try {
    NvcaPart *f = (NvcaPart *)inst;
    f->desc = ((NvcaPartClass *)klass)->desc;
    nvca_part_params_default(&f->p, f->desc->kind);      /* width-to-process 320, process 4 of 4, scale factor 25 */
    f->view = 0; f->meta_data = 0; f->server_events = 0; f->events_ms = 30001; f->time_events_ms = 0;
    f->cf = f->ca = f->cb = nullptr; f->stream = nullptr; f->image_to_overlay = nullptr;
    g_rec_mutex_init(&f->mutex);
}
NVCA_GST_CATCH_VOID("nvca_part_instance_init")
static void nvca_part_class_init(gpointer klass, gpointer class_data)
{
    PartDesc *d = (PartDesc *)class_data;
    ((NvcaPartClass *)klass)->desc = d;
    d->parent_class = g_type_class_peek_parent(klass);
    GObjectClass *go = G_OBJECT_CLASS(klass);
    GstCaps *caps = gst_caps_from_string(GST_VIDEO_CAPS_MAKE("{ BGR }"));
    gst_element_class_add_pad_template(GST_ELEMENT_CLASS(klass), gst_pad_template_new("src", GST_PAD_SRC, GST_PAD_ALWAYS, caps));
    gst_element_class_add_pad_template(GST_ELEMENT_CLASS(klass), gst_pad_template_new("sink", GST_PAD_SINK, GST_PAD_ALWAYS, caps));
    gst_caps_unref(caps);
    gst_element_class_set_static_metadata(GST_ELEMENT_CLASS(klass), d->longname, "Video/Filter",
                                          "Haar part detector (MI355X / HIP implementation)", "nubovca-hip");
    go->set_property = nvca_part_set_property; go->get_property = nvca_part_get_property; go->finalize = nvca_part_finalize;
    INT_PROP(go, PP_VIEW, d->view_prop, "view", "draw or hide the detections on the stream", 0, 1);
    INT_PROP(go, PP_DETECT_EVENT, "detect-event", "detect event", "0 => own face pass; 1 => faces come from upstream events", 0, 1);
    INT_PROP(go, PP_META, d->meta_prop, "send meta data", "0 (default) => no meta data", 0, 1);
    INT_PROP(go, PP_WIDTH, "width-to-process", "width to process", "width of the image the part cascade processes (320 default)", 0, 640);
    INT_PROP(go, PP_X_EVERY_4, "process-x-every-4-frames", "process x every 4 frames", "1,2,3,4 (default)", 0, 4);
    INT_PROP(go, PP_SCALE, "multi-scale-factor", "multi scale factor", "5-50 (25 default)", 0, 51);
    INT_PROP(go, PP_EVENTS, "activate-events", "Activate Events", "0 (default) => no events to the server", 0, 1);
    INT_PROP(go, PP_EVENTS_MS, "events-ms", "Activate Events", "the time, it takes to send events to the servers", 0, 30000);
    g_object_class_install_property(go, PP_OVERLAY, g_param_spec_boxed("image-to-overlay", "image to overlay", "set the url of the image to overlay",
                                    GST_TYPE_STRUCTURE, (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS)));
    GST_VIDEO_FILTER_CLASS(klass)->transform_frame_ip = GST_DEBUG_FUNCPTR(nvca_part_transform_frame_ip);
    if (d->kind != NVCA_PART_EAR) GST_BASE_TRANSFORM_CLASS(klass)->sink_event = GST_DEBUG_FUNCPTR(nvca_part_sink_event);   /* ear: not overridden */
    d->signal_id = g_signal_new(d->signal, G_TYPE_FROM_CLASS(klass), G_SIGNAL_RUN_LAST, 0, NULL, NULL, NULL, G_TYPE_NONE, 1, G_TYPE_STRING);
}
static GType nvca_part_type(PartDesc *d)
{
    GTypeInfo info; memset(&info, 0, sizeof(info));
    info.class_size = sizeof(NvcaPartClass); info.class_init = nvca_part_class_init; info.class_data = d;
    info.instance_size = sizeof(NvcaPart); info.instance_init = nvca_part_instance_init;
    return g_type_register_static(GST_TYPE_VIDEO_FILTER, d->type_name, &info, (GTypeFlags)0);
}

// =====================================================================================
static void debug_category_once()
{
    static gsize once = 0;
    if (g_once_init_enter(&once)) {
        GST_DEBUG_CATEGORY_INIT(nubovca_debug, "nubovca", 0, "NUBOMEDIA-VCA Haar path on MI355X");
        g_once_init_leave(&once, 1);
    }
}
static gboolean plugin_init(GstPlugin *plugin)
{
    debug_category_once();
    gboolean ok = gst_element_register(plugin, "nubofacedetector", GST_RANK_NONE, nvca_face_get_type()) &&
                  gst_element_register(plugin, "nubotracker", GST_RANK_NONE, nvca_trk_get_type());
    for (PartDesc &d : part_descs) ok = ok && gst_element_register(plugin, d.factory, GST_RANK_NONE, nvca_part_type(&d));
    return ok;
}
// One element under a plugin of its own: the reference ships six plugins (libnubofacedetector.so with plugin name
// nubofacedetector, libnuboeyedetector.so / eyefilter, libnubonosedetector.so / nubonosedetector, libnubomouthdetector.so /
// nubomouthdetector, libnuboeardetector.so / earfilter, libnubotracker.so / nubotracker -- modules/nubo_*/.../src/gst-plugins/
// nubo*.c, GST_PLUGIN_DEFINE), and deployments that name them (GST_PLUGIN_PATH entries, `gst-inspect-1.0 <plugin>`, registry
// checks) find the same six here: gst_reference_names/lib<name>.so are stubs (plugin_alias.cpp) that register their element
// through this entry point.  The element types, the batching state and the contexts live in THIS library once, whichever plugin
// registered the factory: branches of different elements are still combined into batched calls.
extern "C" __attribute__((visibility("default"))) gboolean nvca_gst_register_element(GstPlugin *plugin, const char *factory)
{
    debug_category_once();
    if (!plugin || !factory) return FALSE;
    if (!strcmp(factory, "nubofacedetector")) return gst_element_register(plugin, factory, GST_RANK_NONE, nvca_face_get_type());
    if (!strcmp(factory, "nubotracker")) return gst_element_register(plugin, factory, GST_RANK_NONE, nvca_trk_get_type());
    for (PartDesc &d : part_descs)
        if (!strcmp(factory, d.factory)) return gst_element_register(plugin, factory, GST_RANK_NONE, nvca_part_type(&d));
    return FALSE;
}

#ifndef PACKAGE
#define PACKAGE "nubovca"
#endif
GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, nubovca, "NUBOMEDIA-VCA detection filters on MI355X (HIP)", plugin_init,
                  "0.1", "LGPL", "nubovca-hip", "https://github.com/nubomedia/NUBOMEDIA-VCA")
