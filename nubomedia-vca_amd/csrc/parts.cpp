// parts.cpp -- NuboEyeDetector / NuboNoseDetector / NuboMouthDetector / NuboEarDetector stream objects.
// Host glue of kms_{eye,nose,mouth,ear}_detect_process_frame (EYE/kmseyedetect.cpp:915-1064,
// NOSE/kmsnosedetect.cpp:792-868, MOUTH/kmsmouthdetect.cpp:798-873, EAR/kmseardetect.cpp:644-729,767-812):
// ROI geometry, the per-frame / frame-to-frame merging heuristics and the frame gating are O(#faces) integer
// code and stay on the host; every cv:: call (cvtColor, equalizeHist, resize, flip, detectMultiScale) is the
// device implementation behind the public ABI, working on device-resident intermediates owned by the stream.
// std::vector idioms of the reference that rely on libstdc++ behaviour (erase through a reverse iterator,
// erase(end()-i) inside a counting loop) are written out as the index operations they perform.
#include "nvca_internal.h"
#include "host_logic.h"
#include <chrono>
#include <algorithm>
#include <cmath>
#include <climits>
#include <cstring>
#include <deque>

using namespace nvca;

typedef std::vector<nvca_rect> RectV;

struct nvca_part_stream {
    nvca_ctx *ctx;
    nvca_part_params p;
    const nvca_cascade *face, *a, *b;
    RectV faces, la, lb;
    int num_frame = 0, num_frames_to_process = 0, no_det_a = 0, no_det_b = 0;
    std::deque<RectV> queue;
};

namespace nvca { extern double g_jobs_enqueue_s, g_jobs_wait_s, g_jobs_advance_s, g_jobs_fine_s[6]; }
namespace {
inline double mono_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline int cv_round(double v)
{
    if (!(v > -2147483648.5 && v < 2147483647.5)) return INT_MIN;
    return (int)lrint(v);
}
inline int area(const nvca_rect &r) { return r.w * r.h; }

// One frame of one part stream on its way through a batched call: what the three phases hand to each other.
struct RoiJob { DetectJob *job = nullptr; nvca_rect roi{0, 0, 0, 0}; int side = 0; };
// Streams of one batched call that were handed the same frame (the part detectors of one video stream all see the buffer the
// face detector saw) share what they compute identically from it: the upload, the working images and the face pass.  And the
// frames of the call share launches: every working image of one (frame geometry, size, chain) is made by one launch set, every
// face pass of one kind over those images is one job.  Same arithmetic on the same bytes: the results are those of per-stream calls.
struct FrameGroup {
    const void *data = nullptr; int w = 0, h = 0, stride = 0, mem = 0;
    const void *bgr = nullptr;               // device BGR (the caller's, or the one upload of a host frame)
    int eye_index = -1;                      // an eye detector looks at it: index of its full-size gray image / LUT
    size_t upload_at = 0, gray_at = 0;       // arena offsets
};
struct ImageRef { int batch = -1, k = 0; };
struct ImageBatch {                          // the working images [equalizeHist](resize(gray or equalized gray)) of one size
    int W = 0, H = 0, stride = 0, dw = 0, dh = 0; bool eye = false, post_eq = true, flips = false;
    std::vector<int> members;                // frame groups, image k belongs to members[k]
    size_t at = 0, slot = 0; uint8_t *base = nullptr;        // image k at base + k * slot (pitch dw), its mirror image at base + (count + k) * slot
};
struct FacePass {                            // type 0: EYE's plain scan, 1: NOSE / MOUTH SCALE_IMAGE pass, 2: EAR's pass over images and mirror images
    int type = 0; const nvca_cascade *c = nullptr; int batch = 0; double sf = 0;
    std::vector<int> members;                // images of the batch that some stream wants searched
    std::vector<DetectJob *> jobs;           // kJobImages images per job
    ~FacePass() { for (DetectJob *j : jobs) detect_job_free(j); }
};
struct PartWork {
    nvca_part_stream *s = nullptr; const nvca_frame *f = nullptr;
    bool early_return = false, run = false;
    int W = 0, H = 0, fw = 0, fh = 0, pw = 0, ph = 0;
    double scale_o2f = 1, scale_x2o = 1, scale_f2x = 1;
    int lane = 0, group = -1, pass = -1;
    ImageRef small, part_ref;
    std::vector<RoiJob> rois;                // per face, in face order (EYE: right then left; EAR: side 0 then side 1)
    size_t n_side0 = 0;                      // EAR: how many of `rois` belong to side 0
    RectV faces;                             // a stream without a face pass of its own (detect-event): the faces its gate took for THIS frame -- the stream's own
                                             // list may belong to the next frame by the time the back half runs (two calls in flight)
    ~PartWork() { for (RoiJob &r : rois) detect_job_free(r.job); }
};

// detectMultiScale on a sub-matrix of a device image (pitch == cols), as a queued job; job == nullptr: cv::Mat's ROI
// constructor would have thrown (the reference never gets there with such a rectangle: nothing is detected)
int make_roi_job(nvca_part_stream *s, const nvca_cascade *c, const uint8_t *img, int cols, int rows, const nvca_rect &roi,
                 double sf, int mn, int flags, int minw, int minh, RoiJob &out)
{
    out.roi = roi; out.job = nullptr;
    if (roi.x < 0 || roi.y < 0 || roi.w <= 0 || roi.h <= 0 || roi.x + roi.w > cols || roi.y + roi.h > rows) return NVCA_OK;
    out.job = detect_job_new();
    if (!out.job) return NVCA_ERR_NOMEM;
    return make_detect_job(s->ctx, *out.job, c, img + (size_t)roi.y * cols + roi.x, roi.w, roi.h, cols, NVCA_MEM_DEVICE, sf, mn, flags, minw, minh, 0, 0, false);
}
RectV roi_result(const RoiJob &r)
{
    if (!r.job) return RectV();
    const std::vector<nvca_rect> &v = detect_job_out(r.job, 0);
    return RectV(v.begin(), v.begin() + std::min<size_t>(v.size(), 256));
}

// the per-frame / frame-to-frame merging heuristics live in host_logic.cpp (pure host code: also built under the sanitizers)
// kms_ear_detect_find_ears EAR/kmseardetect.cpp:644-729, in two halves around the (queued) ear searches.
// `profile_faces`: the profile-face pass on this side's image (the image itself / its mirror), :656-659.
// First half: the bookkeeping the reference does before it searches, and one FIND_BIGGEST job per profile face.
int find_ears_begin(nvca_part_stream *s, PartWork &w, const std::vector<nvca_rect> &profile_faces, const uint8_t *ear_img,
                    const nvca_cascade *ear_cascade, int side)
{
    int rc;
    const int fcols = w.fw, ecols = w.pw, erows = w.ph;
    const double scale_f2e = w.scale_f2x;
    s->faces.assign(profile_faces.begin(), profile_faces.begin() + std::min<size_t>(profile_faces.size(), 256));
    if (s->faces.empty()) return NVCA_OK;
    RectV &ears = side == 0 ? s->la : s->lb;
    if (!ears.empty()) ears.clear();
    else if (s->no_det_a < 4) s->no_det_a += 1;            // MAX_NUM_FPS_WITH_NO_DETECTION 4, one counter for both sides
    else { s->no_det_a = 0; ears.clear(); }
    for (nvca_rect &r : s->faces) {
        const int top_height = cv_round((float)r.h * 20 / 100), down_height = cv_round((float)r.h * 20 / 100);
        if (side == 0) {
            r.y = (int)((r.y + top_height) * scale_f2e);
            r.x = (int)((r.x + (r.w / 2)) * scale_f2e);
            r.h = (int)((r.h - down_height) * scale_f2e);
            r.w = (int)((r.w / 2) * scale_f2e + 50);        // EXTRA_ROI
            if (r.x + r.w > ecols) r.w = ecols - r.x - 1;
        } else {
            r.y = (int)((r.y + top_height) * scale_f2e);
            r.x = (int)((fcols - r.x - r.w) * scale_f2e - 50);
            r.h = (int)((r.h - down_height) * scale_f2e);
            r.w = (int)((r.w / 2) * scale_f2e);
            if (r.x < 0) r.x = 0;
        }
        RoiJob rj; rj.side = side;
        if ((rc = make_roi_job(s, ear_cascade, ear_img, ecols, erows, r, 1.1, 3, NVCA_HAAR_FIND_BIGGEST_OBJECT, 1, 1, rj))) { detect_job_free(rj.job); return rc; }
        w.rois.push_back(rj);
    }
    return NVCA_OK;
}
// second half: the ears found in each profile face's region, in face order
void find_ears_end(nvca_part_stream *s, const PartWork &w, size_t first, size_t last, int side)
{
    RectV &ears = side == 0 ? s->la : s->lb;
    for (size_t k = first; k < last; k++) {
        const nvca_rect &r = w.rois[k].roi;
        for (const nvca_rect &e : roi_result(w.rois[k])) {
            nvca_rect o;
            o.x = cv_round((r.x + e.x) * w.scale_x2o); o.y = cv_round((r.y + e.y) * w.scale_x2o);
            o.w = (int)((e.w - 1) * w.scale_x2o); o.h = (int)((e.h - 1) * w.scale_x2o);
            ears.push_back(o);
        }
    }
}
} // namespace

static void part_calls_abandon_for(nvca_ctx *ctx, const nvca_part_stream *s);

extern "C" {

void nvca_part_params_default(nvca_part_params *p, int kind)
try {
    if (!p) return;
    p->kind = kind; p->width_to_process = 320; p->process_x_every_4 = 4; p->scale_factor_pct = 25; p->detect_event = 0;
}
NVCA_API_CATCH_VOID

int nvca_part_stream_create(nvca_ctx *ctx, const nvca_part_params *params, const nvca_cascade *face, const nvca_cascade *a,
                            const nvca_cascade *b, nvca_part_stream **out)
try {
    if (!ctx || !params || !face || !a || !out || params->kind < NVCA_PART_EYE || params->kind > NVCA_PART_EAR) return NVCA_ERR_ARG;
    if ((params->kind == NVCA_PART_EYE || params->kind == NVCA_PART_EAR) && !b) return NVCA_ERR_ARG;
    nvca_part_stream *s = new (std::nothrow) nvca_part_stream();
    if (!s) return NVCA_ERR_NOMEM;
    s->ctx = ctx; s->p = *params; s->face = face; s->a = a; s->b = b;
    *out = s;
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)
void nvca_part_stream_destroy(nvca_part_stream *s)
try {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    {
        // a submitted, not yet collected batch holds this stream (its gates' snapshot, its jobs): such calls are abandoned first --
        // the newer one before the older one, each rolled back and drained -- and their tickets become unknown
        std::lock_guard<std::recursive_mutex> lk(s->ctx->mu);
        part_calls_abandon_for(s->ctx, s);
    }
    (void)hipStreamSynchronize(s->ctx->stream);
    delete s;
}
NVCA_API_CATCH_VOID
int nvca_part_stream_set_params(nvca_part_stream *s, const nvca_part_params *params)
try {
    if (!s || !params || params->kind != s->p.kind) return NVCA_ERR_ARG;
    s->p = *params;
    return NVCA_OK;
}
NVCA_API_CATCH((s ? s->ctx : nullptr))
int nvca_part_stream_push_faces(nvca_part_stream *s, const nvca_rect *faces, int n)
try {
    if (!s || n < 0 || (n > 0 && !faces)) return NVCA_ERR_ARG;
    if (s->queue.size() < 16) s->queue.emplace_back(faces, faces + n);
    return NVCA_OK;
}
NVCA_API_CATCH((s ? s->ctx : nullptr))

int nvca_part_stream_faces(const nvca_part_stream *s, nvca_rect *out, int cap, int *n_out)
try {
    if (!s || !n_out || cap < 0 || (cap > 0 && !out)) return NVCA_ERR_ARG;
    *n_out = (int)s->faces.size();
    for (int i = 0; i < std::min(*n_out, cap); i++) out[i] = s->faces[i];
    return NVCA_OK;
}
NVCA_API_CATCH((s ? s->ctx : nullptr))

} // extern "C"

// ---- a batched call in two halves -------------------------------------------------------------------------------------------------
// front: the gates of every stream, the working images and the face passes QUEUED (nothing is waited for); back: the face passes'
// results, the part searches in every face's region, the merging heuristics.  nvca_part_batch_process runs them back to back;
// nvca_part_batch_submit / _collect let the caller queue the next frames' front half before it collects this frames' back half, so
// that the image chains and face passes of tick k + 1 fill the GPU while tick k's searches are advanced on the host (two calls may be
// in flight: each uses the working-image set, candidate buffers and lanes of its ticket's parity).
namespace {
struct StreamSnap { nvca_part_stream *s; RectV faces, la, lb; int num_frame, to_process, no_a, no_b; bool popped; RectV front; };
static double g_total_acc = 0, g_phase3_acc = 0;          // NVCA_PART_STATS (diagnostic, one context at a time)
static constexpr int kCallLanes = 3;                      // lanes of one call: images on the first, face passes and part searches side by side on all three
struct PartCall {
    nvca_ctx *ctx = nullptr; int n = 0, parity = 0, seq = 0;
    std::vector<nvca_part_stream *> streams; std::vector<nvca_frame> frames;
    std::vector<FrameGroup> groups;
    std::vector<ImageBatch> batches;
    std::deque<FacePass> passes;
    std::vector<PartWork> work;
    std::vector<DetectJob *> jobs;
    std::vector<int> job_lane;
    int n_eye = 0;
    JobRound *round = nullptr; bool queued = false;       // the face passes' first round, left in flight by the front half
    double t0 = 0, t1 = 0;
    // The gates of phase 1a (and find_ears_begin in phase 2) advance per-stream state; plans, buffers and launches come after
    // them and may still fail (too many scales, allocation, a refused launch).  Whatever the error, the call leaves every
    // stream as it found it -- the GStreamer shim re-runs the streams one by one after a refused batch, and a gate that had
    // already advanced would then advance twice and drop a queued face event.  Nothing of the call may stay in flight either:
    // the caller's frames (H2D copies) and the arena are only safe to reuse once the lanes have drained.
    std::vector<StreamSnap> snaps; bool armed = true;
    ~PartCall()
    {
        if (armed && ctx) {
            (void)hipDeviceSynchronize();
            for (StreamSnap &g : snaps) {
                nvca_part_stream *s = g.s;
                s->faces.swap(g.faces); s->la.swap(g.la); s->lb.swap(g.lb);
                s->num_frame = g.num_frame; s->num_frames_to_process = g.to_process; s->no_det_a = g.no_a; s->no_det_b = g.no_b;
                if (g.popped) s->queue.push_front(std::move(g.front));
            }
        }
        job_round_free(round);
    }
};
struct CallSets {            // the context's per-call selections, put back when the half is over
    nvca_ctx *c; int part_set, roi_set;
    CallSets(nvca_ctx *x, int parity) : c(x), part_set(x->part_set), roi_set(x->roi_set) { c->part_set = parity; c->roi_set = 1 + parity; }
    ~CallSets() { c->part_set = part_set; c->roi_set = roi_set; c->cur_lane = 0; }
};

int part_front(nvca_ctx *ctx, PartCall &c, int n, nvca_part_stream *const *streams, const nvca_frame *frames)
{
    if (n < 0 || (n > 0 && (!streams || !frames))) return NVCA_ERR_ARG;
    for (int i = 0; i < n; i++) {
        const nvca_part_stream *s = streams[i]; const nvca_frame *f = &frames[i];
        if (!s || s->ctx != ctx || !f->data || f->width <= 0 || f->height <= 0 || f->stride < f->width * 3 || s->p.width_to_process <= 0 ||
            (f->mem != NVCA_MEM_HOST && f->mem != NVCA_MEM_DEVICE)) return NVCA_ERR_ARG;
        for (int j = 0; j < i; j++) if (streams[j] == s) { ctx->set_error("a part stream may appear once per batch"); return NVCA_ERR_ARG; }
        // every frame is validated before any stream's gate advances: a refused call leaves all streams as they were
        const float o2f = (s->p.kind != NVCA_PART_EAR && s->p.detect_event) ? 1.f : ((float)f->width) / ((float)160);
        const float x2o = ((float)f->width) / ((float)s->p.width_to_process);
        if (cv_round(f->width / (double)o2f) <= 0 || cv_round(f->height / (double)o2f) <= 0 || cv_round(f->width / (double)x2o) <= 0 || cv_round(f->height / (double)x2o) <= 0) {
            ctx->set_error("part stream: frame too small"); return NVCA_ERR_ARG;
        }
    }
    (void)hipSetDevice(ctx->device);
    c.ctx = ctx; c.n = n;
    c.streams.assign(streams, streams + n); c.frames.assign(frames, frames + n);
    c.work.resize(n);
    std::vector<FrameGroup> &groups = c.groups;
    std::vector<ImageBatch> &batches = c.batches;
    std::deque<FacePass> &passes = c.passes;
    std::vector<PartWork> &work = c.work;
    std::vector<DetectJob *> &jobs = c.jobs;
    std::vector<int> &job_lane = c.job_lane;
    int &n_eye = c.n_eye;
    const int lane_base = c.parity ? 1 + kCallLanes : 1;           // lanes 1 .. 3 / 4 .. 6; calls with several streams stay off lane 0, where a face detector's batch may be in flight
    CallSets sets(ctx, c.parity);
    // image k of the batch of (frame geometry, size, chain): asked for by frame group gi
    auto request = [&](int gi, int dw, int dh, bool eye, bool post_eq) {
        const FrameGroup &fg = groups[gi];
        ImageRef r;
        for (size_t bi = 0; bi < batches.size() && r.batch < 0; bi++) {
            const ImageBatch &b = batches[bi];
            if (b.W == fg.w && b.H == fg.h && b.stride == fg.stride && b.dw == dw && b.dh == dh && b.eye == eye && b.post_eq == post_eq) r.batch = (int)bi;
        }
        if (r.batch < 0) {
            batches.emplace_back();
            ImageBatch &b = batches.back();
            b.W = fg.w; b.H = fg.h; b.stride = fg.stride; b.dw = dw; b.dh = dh; b.eye = eye; b.post_eq = post_eq;
            r.batch = (int)batches.size() - 1;
        }
        ImageBatch &b = batches[r.batch];
        const auto it = std::find(b.members.begin(), b.members.end(), gi);
        r.k = (int)(it - b.members.begin());
        if (it == b.members.end()) b.members.push_back(gi);
        return r;
    };
    c.snaps.reserve(n);
    for (int i = 0; i < n; i++) {
        nvca_part_stream *s = streams[i];
        c.snaps.push_back(StreamSnap{s, s->faces, s->la, s->lb, s->num_frame, s->num_frames_to_process, s->no_det_a, s->no_det_b, false, RectV()});
    }
    int rc = NVCA_OK;
    const int D = NVCA_MEM_DEVICE;
#define CK(e) do { if ((rc = (e))) return rc; } while (0)
    const bool stats = ctx->sw.part_stats > 0;
    c.t0 = stats ? mono_s() : 0;
    // ---- phase 1a: gating of every stream, in stream order; what the streams that run need is only noted down here
    for (int i = 0; i < n; i++) {
        PartWork &w = work[i];
        nvca_part_stream *s = w.s = streams[i]; const nvca_frame *f = w.f = &c.frames[i];
        const int kind = s->p.kind, W = w.W = f->width, H = w.H = f->height;
        // conf_images: float arithmetic (EYE/kmseyedetect.cpp:331-339 and siblings)
        const float o2f = (kind != NVCA_PART_EAR && s->p.detect_event) ? ((float)W) / ((float)W) : ((float)W) / ((float)160);
        const float x2o = ((float)W) / ((float)s->p.width_to_process);
        const float f2x = ((float)o2f) / ((float)x2o);
        w.scale_o2f = o2f; w.scale_x2o = x2o; w.scale_f2x = f2x;
        bool received = true;
        if (kind != NVCA_PART_EAR) {                                            // __receive_event
            if (s->p.detect_event) {
                received = false;
                if (!s->queue.empty()) {
                    s->faces = s->queue.front(); s->queue.pop_front();
                    c.snaps[i].popped = true; c.snaps[i].front = s->faces;
                    received = true;
                    s->num_frames_to_process = 10 / (5 - s->p.process_x_every_4);
                }
            }
            if (!received && s->num_frames_to_process <= 0) w.early_return = true;
        }
        if (w.early_return) continue;
        if (4 == s->num_frame) s->num_frame = 0;                                // GOP (the reference resets at the end of the frame before: nothing reads the counter in between)
        s->num_frame++;
        const int px = s->p.process_x_every_4;
        w.run = (2 == px && (1 == s->num_frame % 2)) || ((2 != px) && (s->num_frame <= px));
        if (!w.run) continue;
        s->num_frames_to_process--;
        const int fw = w.fw = cv_round(W / w.scale_o2f), fh = w.fh = cv_round(H / w.scale_o2f);
        const int pw = w.pw = cv_round(W / w.scale_x2o), ph = w.ph = cv_round(H / w.scale_x2o);
        if (fw <= 0 || fh <= 0 || pw <= 0 || ph <= 0) { ctx->set_error("part stream: frame too small"); return NVCA_ERR_ARG; }
        for (size_t gi = 0; gi < groups.size(); gi++) {
            const FrameGroup &fg = groups[gi];
            if (fg.data == f->data && fg.w == W && fg.h == H && fg.stride == f->stride && fg.mem == f->mem) w.group = (int)gi;
        }
        if (w.group < 0) {
            groups.emplace_back();
            w.group = (int)groups.size() - 1;
            FrameGroup &fg = groups.back();
            fg.data = f->data; fg.w = W; fg.h = H; fg.stride = f->stride; fg.mem = f->mem;
        }
        w.lane = n > 1 ? lane_base + w.group % kCallLanes : 0;     // the part searches of one frame's streams share a lane
        // the images this stream works on: requested here, computed below for all streams at once
        if (kind == NVCA_PART_EYE) {
            FrameGroup &fg = groups[w.group];
            if (fg.eye_index < 0) fg.eye_index = n_eye++;
            if (0 == s->p.detect_event) w.small = request(w.group, fw, fh, true, false);
            w.part_ref = request(w.group, pw, ph, true, true);
        } else {
            if (kind == NVCA_PART_EAR || 0 == s->p.detect_event) w.small = request(w.group, fw, fh, false, true);
            w.part_ref = request(w.group, pw, ph, false, true);
            if (kind == NVCA_PART_EAR) batches[w.small.batch].flips = true;
        }
        // its face pass: one job per (kind of pass, cascade, image set, scale factor), however many streams ask for it
        if (w.small.batch >= 0) {
            const double sf_face = 1 + s->p.scale_factor_pct * 1.0 / 100;
            const int type = kind == NVCA_PART_EYE ? 0 : (kind == NVCA_PART_EAR ? 2 : 1);
            for (size_t pi = 0; pi < passes.size(); pi++)
                if (passes[pi].type == type && passes[pi].c == s->face && passes[pi].batch == w.small.batch && passes[pi].sf == sf_face) w.pass = (int)pi;
            if (w.pass < 0) { passes.emplace_back(); w.pass = (int)passes.size() - 1; FacePass &fp = passes.back(); fp.type = type; fp.c = s->face; fp.batch = w.small.batch; fp.sf = sf_face; }
            FacePass &fp = passes[w.pass];
            if (std::find(fp.members.begin(), fp.members.end(), w.small.k) == fp.members.end()) fp.members.push_back(w.small.k);
        }
        if (w.pass < 0 && kind != NVCA_PART_EAR) w.faces = s->faces;
    }
    // ---- phase 1b: every image the call needs, in a handful of launches
    {
        ctx->cur_lane = n > 1 ? lane_base : 0;
        // arena: uploads of host frames | full-size gray images of the eye detectors' frames | the working images, batch by batch
        size_t need = 0;
        auto carve = [&](size_t bytes) { const size_t at = need; need += (bytes + 255) & ~(size_t)255; return at; };
        for (FrameGroup &fg : groups) if (fg.mem == NVCA_MEM_HOST) fg.upload_at = carve((size_t)fg.stride * fg.h);
        for (int e = 0; e < n_eye; e++)              // in LUT order: frames of one geometry then sit at equal distances
            for (FrameGroup &fg : groups) if (fg.eye_index == e) fg.gray_at = carve((size_t)fg.w * fg.h);
        for (ImageBatch &b : batches) { b.slot = ((size_t)b.dw * b.dh + 255) & ~(size_t)255; b.at = carve(b.slot * b.members.size() * (b.flips ? 2 : 1)); }
        uint8_t *arena = nullptr, *eye_luts = nullptr;
        CK(part_arena(ctx, need, &arena));
        size_t max_members = 1;
        for (const ImageBatch &b : batches) max_members = std::max(max_members, b.members.size());
        CK(part_luts(ctx, n_eye, (int)max_members, &eye_luts));
        for (FrameGroup &fg : groups) {
            fg.bgr = fg.data;
            if (fg.mem == NVCA_MEM_HOST) {
                CK(caller_h2d(ctx, arena + fg.upload_at, fg.data, (size_t)fg.stride * (fg.h - 1) + (size_t)fg.w * 3, ctx->cs()));
                fg.bgr = arena + fg.upload_at;
            }
        }
        // EYE :948-950: cvtColor + equalizeHist of the whole frame -- gray images + LUTs here, the LUT is applied where the resizes read
        {
            std::vector<char> done(groups.size(), 0);
            for (size_t gi = 0; gi < groups.size(); gi++) {
                if (groups[gi].eye_index < 0 || done[gi]) continue;
                // frames of one geometry whose gray slots and LUT indices run on: one launch set
                std::vector<const void *> srcs; const FrameGroup &g0 = groups[gi];
                const size_t slot = ((size_t)g0.w * g0.h + 255) & ~(size_t)255;
                for (size_t gj = gi; gj < groups.size(); gj++) {
                    const FrameGroup &fg = groups[gj];
                    if (fg.eye_index < 0 || done[gj] || fg.w != g0.w || fg.h != g0.h || fg.stride != g0.stride) continue;
                    if (fg.eye_index != g0.eye_index + (int)srcs.size() || fg.gray_at != g0.gray_at + slot * srcs.size()) continue;
                    srcs.push_back(fg.bgr); done[gj] = 1;
                }
                CK(part_gray_eq(ctx, srcs.data(), (int)srcs.size(), g0.w, g0.h, g0.stride, arena + g0.gray_at, slot, eye_luts + (size_t)g0.eye_index * 256));
            }
        }
        for (ImageBatch &b : batches) {
            PartImageBatch ib;
            ib.bgr = !b.eye; ib.post_eq = b.post_eq; ib.sw = b.W; ib.sh = b.H; ib.sstride = b.eye ? b.W : b.stride; ib.dw = b.dw; ib.dh = b.dh;
            ib.dst = b.base = arena + b.at; ib.slot = b.slot;
            for (int gi : b.members) {
                const FrameGroup &fg = groups[gi];
                ib.src.push_back(b.eye ? (const void *)(arena + fg.gray_at) : fg.bgr);
                if (b.eye) ib.lut_idx.push_back(fg.eye_index);
            }
            CK(part_image_batch(ctx, ib, eye_luts));
            if (b.flips) CK(part_flip_batch(ctx, b.base, b.base + b.slot * b.members.size(), b.dw, b.dh, (int)b.members.size(), b.slot));     // EAR :800
        }
        // the face passes: members in image order, so that a pass over all images of a batch reads them in place
        int next_lane = lane_base + 1;
        std::vector<int> used_lanes;
        for (FacePass &fp : passes) {
            std::sort(fp.members.begin(), fp.members.end());
            const ImageBatch &b = batches[fp.batch];
            const int per_job = fp.type == 2 ? kJobImages / 2 : kJobImages;
            for (size_t m0 = 0; m0 < fp.members.size(); m0 += per_job) {
                const size_t m1 = std::min(fp.members.size(), m0 + per_job);
                DetectJob *job = detect_job_new();
                if (!job) return NVCA_ERR_NOMEM;
                fp.jobs.push_back(job);
                const uint8_t *first = b.base + b.slot * fp.members[m0];
                if (fp.type == 0) CK(make_detect_job(ctx, *job, fp.c, first, b.dw, b.dh, b.dw, D, fp.sf, 3, 0, 30, 30, 0, 0, false));                               // EYE :958-960
                else if (fp.type == 1) CK(make_detect_job(ctx, *job, fp.c, first, b.dw, b.dh, b.dw, D, fp.sf, 2, NVCA_HAAR_SCALE_IMAGE, 3, 3, 0, 0, false));       // NOSE :843-846, MOUTH :845-848
                else CK(make_detect_job(ctx, *job, fp.c, first, b.dw, b.dh, b.dw, D, fp.sf, 2, NVCA_HAAR_SCALE_IMAGE, 3, 3, b.dw, b.dh, false));                    // EAR :656-659
                for (size_t m = m0 + 1; m < m1; m++) if (detect_job_add_image(job, b.base + b.slot * fp.members[m]) < 0) return NVCA_ERR_ARG;
                if (fp.type == 2)          // ... and the mirrored images (EAR :796-803): results k + count
                    for (size_t m = m0; m < m1; m++) if (detect_job_add_image(job, b.base + b.slot * (b.members.size() + fp.members[m])) < 0) return NVCA_ERR_ARG;
                const int lane = n > 1 ? next_lane : 0;
                next_lane = next_lane + 1 < lane_base + kCallLanes ? next_lane + 1 : lane_base + 1;
                jobs.push_back(job); job_lane.push_back(lane); used_lanes.push_back(lane);
            }
        }
        for (const PartWork &w : work) if (w.run) used_lanes.push_back(w.lane);     // the part searches of phase 2 read these images on the streams' lanes
        std::sort(used_lanes.begin(), used_lanes.end());
        used_lanes.erase(std::unique(used_lanes.begin(), used_lanes.end()), used_lanes.end());
        CK(part_images_done(ctx, used_lanes.data(), (int)used_lanes.size()));
    }
    c.t1 = stats ? mono_s() : 0;
    // the face passes' launch sets are queued here and collected by the back half (small-image jobs: one k_roi launch for all of them)
    if (!jobs.empty()) {
        c.round = job_round_new();
        if (!c.round) return NVCA_ERR_NOMEM;
        CK(detect_jobs_begin(ctx, jobs.data(), (int)jobs.size(), job_lane.data(), c.round, &c.queued));
    }
#undef CK
    return NVCA_OK;
}

int part_back(nvca_ctx *ctx, PartCall &c, nvca_rect *out_a, int cap_a, int *n_a, nvca_rect *out_b, int cap_b, int *n_b)
{
    const int n = c.n;
    if ((n > 0 && (!n_a || !n_b)) || cap_a < 0 || cap_b < 0 || (cap_a > 0 && !out_a) || (cap_b > 0 && !out_b)) return NVCA_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    std::vector<FrameGroup> &groups = c.groups;
    std::vector<ImageBatch> &batches = c.batches;
    std::deque<FacePass> &passes = c.passes;
    std::vector<PartWork> &work = c.work;
    std::vector<DetectJob *> &jobs = c.jobs;
    std::vector<int> &job_lane = c.job_lane;
    const int lane_base = c.parity ? 1 + kCallLanes : 1;
    CallSets sets(ctx, c.parity);
    int rc = NVCA_OK;
#define CK(e) do { if ((rc = (e))) return rc; } while (0)
    const bool stats = ctx->sw.part_stats > 0;    // diagnostic: the host's time per phase of calls with n (default 8) or more streams, every 8 such calls
    static double acc[4] = {0, 0, 0, 0}; static int calls = 0;
    const int stats_min = stats ? ctx->sw.part_stats : 8;
    const bool whole_on = stats && n >= ctx->sw.part_stats;
    const double ts0 = c.t0, ts1 = c.t1, tb0 = stats ? mono_s() : 0;
    struct Whole { bool on; double t0, front; ~Whole() { if (on) g_total_acc += mono_s() - t0 + front; } } whole{whole_on, tb0, ts1 - ts0};
    // Every face pass waits for the images (part_images_done), so draining the passes' lanes drains the image lane's work
    // too.  A call without any face pass (detect-event streams: the faces were pushed) has nobody waiting for it: the H2D
    // copies of the caller's frames and the image kernels are drained here, before the call can return -- the caller may
    // recycle its buffers, and the next call carves the same arena on another lane.
    if (jobs.empty() && !groups.empty()) {
        const hipError_t he = hipStreamSynchronize(ctx->lane_streams[n > 1 ? lane_base : 0]);
        if (he != hipSuccess) { ctx->set_error(std::string("hipStreamSynchronize: ") + hipGetErrorString(he)); return NVCA_ERR_HIP; }
    }
    if (!jobs.empty()) CK(detect_jobs_finish(ctx, jobs.data(), (int)jobs.size(), job_lane.data(), c.round, c.queued));          // wait 1: every face pass
    c.queued = false;
    // a stream's faces: result k of its pass's job
    auto pass_result = [&](const PartWork &w, bool mirrored) -> const std::vector<nvca_rect> & {
        const FacePass &fp = passes[w.pass];
        const size_t pos = std::find(fp.members.begin(), fp.members.end(), w.small.k) - fp.members.begin();
        const size_t per_job = fp.type == 2 ? kJobImages / 2 : kJobImages, ji = pos / per_job, in_job = std::min(fp.members.size() - ji * per_job, per_job);
        return detect_job_out(fp.jobs[ji], (int)(pos % per_job + (mirrored ? in_job : 0)));
    };
    const double ts2 = stats ? mono_s() : 0;
    // ---- phase 2: the part searches of every face of every stream
    jobs.clear(); job_lane.clear();
    for (int i = 0; i < n; i++) {
        PartWork &w = work[i];
        if (!w.run) continue;
        nvca_part_stream *s = w.s;
        const int kind = s->p.kind;
        const uint8_t *part = batches[w.part_ref.batch].base + batches[w.part_ref.batch].slot * w.part_ref.k;
        if (kind == NVCA_PART_EAR) {
            CK(find_ears_begin(s, w, pass_result(w, false), part, s->a, 0));
            w.n_side0 = w.rois.size();
            CK(find_ears_begin(s, w, pass_result(w, true), part, s->b, 1));
        } else {
            if (w.pass >= 0) { const std::vector<nvca_rect> &fv = pass_result(w, false); s->faces.assign(fv.begin(), fv.begin() + std::min<size_t>(fv.size(), 256)); }
            const double scale_f2x = w.scale_f2x;
            const RectV &faces_now = w.pass >= 0 ? s->faces : w.faces;
            for (const nvca_rect &r : faces_now) {
                if (kind == NVCA_PART_EYE) {
                    nvca_rect ra, fr, fl;
                    ra.x = (int)(r.x * scale_f2x); ra.y = (int)(r.y * scale_f2x); ra.w = (int)(r.w * scale_f2x); ra.h = (int)(r.h * scale_f2x);
                    const int down_height = cv_round((float)ra.h * 40 / 100), top_height = cv_round((float)ra.h * 25 / 100);
                    fr.x = ra.x; fr.y = ra.y + top_height; fr.h = ra.h - top_height - down_height; fr.w = ra.w / 2;
                    fl.x = ra.x + ra.w / 2; fl.y = ra.y + top_height; fl.h = ra.h - top_height - down_height; fl.w = ra.w / 2;
                    RoiJob jr, jl; jl.side = 1;
                    rc = make_roi_job(s, s->a, part, w.pw, w.ph, fr, 1.1, 2, NVCA_HAAR_SCALE_IMAGE, 20, 20, jr);
                    if (!rc) rc = make_roi_job(s, s->b, part, w.pw, w.ph, fl, 1.1, 2, NVCA_HAAR_SCALE_IMAGE, 20, 20, jl);
                    w.rois.push_back(jr); w.rois.push_back(jl);
                    if (rc) return rc;
                } else {
                    nvca_rect ra;
                    if (kind == NVCA_PART_NOSE) {                   // NOSE :858-868
                        const int top = cv_round((float)r.h * 25 / 100), down = cv_round((float)r.h * 10 / 100);
                        const int side = cv_round((float)r.w * 25 / 100);
                        ra.y = (int)((r.y + top) * scale_f2x); ra.x = (int)((r.x + side) * scale_f2x);
                        ra.h = (int)((r.h - down - top) * scale_f2x); ra.w = (int)((r.w - side) * scale_f2x);
                    } else {                                        // MOUTH :859-865
                        const int half = cv_round((float)r.h / 1.8);
                        ra.y = (int)((r.y + half) * scale_f2x); ra.x = (int)(r.x * scale_f2x);
                        ra.h = (int)(half * scale_f2x); ra.w = (int)(r.w * scale_f2x);
                    }
                    RoiJob jr;
                    rc = make_roi_job(s, s->a, part, w.pw, w.ph, ra, 1.1, 3, NVCA_HAAR_FIND_BIGGEST_OBJECT, 1, 1, jr);
                    w.rois.push_back(jr);
                    if (rc) return rc;
                }
            }
        }
        for (RoiJob &r : w.rois) if (r.job) { jobs.push_back(r.job); job_lane.push_back(w.lane); }
    }
    const double ts3 = stats ? mono_s() : 0;
    CK(run_detect_jobs(ctx, jobs.data(), (int)jobs.size(), job_lane.data()));          // wait 2 (+ one more for searches that narrowed)
    if (stats) {
        const double ts4 = mono_s();
        if (n >= stats_min) { acc[0] += ts1 - ts0; acc[1] += ts2 - tb0; acc[2] += ts3 - ts2; acc[3] += ts4 - ts3; }
        else { g_jobs_enqueue_s = g_jobs_wait_s = g_jobs_advance_s = 0; for (double &v : g_jobs_fine_s) v = 0; }
        if (n >= stats_min && ++calls % 8 == 0) {
            fprintf(stderr, "nubovca part batch (ms per call): image chains %.3f, face passes %.3f, roi set-up %.3f, roi searches %.3f | in the job rounds: enqueue %.3f, wait %.3f, advance %.3f | merging (previous calls) %.3f, whole call (previous 8) %.3f\n",
                    acc[0] / 8 * 1e3, acc[1] / 8 * 1e3, acc[2] / 8 * 1e3, acc[3] / 8 * 1e3, g_jobs_enqueue_s / 8 * 1e3, g_jobs_wait_s / 8 * 1e3, g_jobs_advance_s / 8 * 1e3,
                    g_phase3_acc / 8 * 1e3, g_total_acc / 8 * 1e3);
            fprintf(stderr, "nubovca part batch, job rounds in detail (ms per call): adding jobs %.3f (%.0f jobs), launch %.3f, collect %.3f, advance on the helpers %.3f, advance serial %.3f\n",
                    g_jobs_fine_s[0] / 8 * 1e3, g_jobs_fine_s[5] / 8, g_jobs_fine_s[1] / 8 * 1e3, g_jobs_fine_s[2] / 8 * 1e3, g_jobs_fine_s[3] / 8 * 1e3, g_jobs_fine_s[4] / 8 * 1e3);
            for (double &v : g_jobs_fine_s) v = 0;
            acc[0] = acc[1] = acc[2] = acc[3] = 0; g_jobs_enqueue_s = g_jobs_wait_s = g_jobs_advance_s = 0; g_phase3_acc = 0; g_total_acc = 0;
        }
    }
#undef CK
    c.armed = false;                // nothing below can fail short of an exception -- which the containers' strong guarantee
                                           // and the ABI barrier turn into an error code; the device work is complete
    // ---- phase 3: merging heuristics, hysteresis, emission -- in stream order
    struct P3 { bool on; double t0; ~P3() { if (on) g_phase3_acc += mono_s() - t0; } } p3{whole_on, whole_on ? mono_s() : 0};
    for (int i = 0; i < n; i++) {
        PartWork &w = work[i];
        nvca_part_stream *s = w.s;
        const int kind = s->p.kind;
        if (!w.early_return) {
            RectV res_a, res_b;
            if (w.run) {
                const int iscale = (int)w.scale_x2o;                      // the merge helpers take `int scale`
                if (kind == NVCA_PART_EAR) {
                    find_ears_end(s, w, 0, w.n_side0, 0);
                    find_ears_end(s, w, w.n_side0, w.rois.size(), 1);
                } else if (kind == NVCA_PART_EYE) {
                    for (size_t k = 0; k + 1 < w.rois.size(); k += 2) {
                        const nvca_rect &fr = w.rois[k].roi, &fl = w.rois[k + 1].roi;
                        RectV eye_r = roi_result(w.rois[k]), eye_l = roi_result(w.rois[k + 1]), aux;
                        to_global(eye_r, fr, iscale); to_global(eye_l, fl, iscale);
                        if (!eye_r.empty()) {
                            merge_eyes_current(fr, eye_r, eye_r, iscale, false);
                            merge_eyes_consecutive(eye_r, s->la, aux);
                            res_a.insert(res_a.end(), aux.begin(), aux.end());
                        }
                        if (!eye_l.empty()) {
                            merge_eyes_current(fl, res_a, eye_l, iscale, true);
                            merge_eyes_consecutive(eye_l, s->lb, aux);
                            res_b.insert(res_b.end(), aux.begin(), aux.end());
                        }
                    }
                } else {
                    const int dis = kind == NVCA_PART_NOSE ? 6 : 4;
                    for (const RoiJob &rj : w.rois) {
                        RectV cn = roi_result(rj), aux;
                        if (!cn.empty()) {
                            merge_consecutive_nm(cn, s->la, rj.roi, iscale, dis, aux);
                            res_a.insert(res_a.end(), aux.begin(), aux.end());
                        }
                    }
                }
                if (kind == NVCA_PART_EYE) {                                // per-side hysteresis EYE :1034-1064
                    if (res_a.empty()) { if (s->no_det_a < 1) s->no_det_a += 1; else { s->no_det_a = 0; s->la.clear(); } }
                    else { s->no_det_a = 0; s->la = res_a; }
                    if (res_b.empty()) { if (s->no_det_b < 1) s->no_det_b += 1; else { s->no_det_b = 0; s->lb.clear(); } }
                    else { s->no_det_b = 0; s->lb = res_b; }
                }
            }
            if (kind == NVCA_PART_NOSE || kind == NVCA_PART_MOUTH) s->la = res_a;   // rebuilt on every call that gets here
        }
        n_a[i] = (int)s->la.size(); n_b[i] = (int)s->lb.size();
        for (int k = 0; k < std::min(n_a[i], cap_a); k++) out_a[(size_t)i * cap_a + k] = s->la[k];
        for (int k = 0; k < std::min(n_b[i], cap_b); k++) out_b[(size_t)i * cap_b + k] = s->lb[k];
    }
    return NVCA_OK;
}

} // namespace
// outstanding calls that hold stream s (nullptr: any) are given up: rolled back (newest first), drained, deleted
static void part_calls_abandon_for(nvca_ctx *ctx, const nvca_part_stream *s)
{
    bool hit = false;
    for (void *o : ctx->part_calls)
        if (o) { const PartCall *c = (const PartCall *)o; if (!s || std::find(c->streams.begin(), c->streams.end(), s) != c->streams.end()) hit = true; }
    if (!hit) return;
    // (both go: the newer call's gates were taken on top of the older one's)
    PartCall *a = (PartCall *)ctx->part_calls[0], *b = (PartCall *)ctx->part_calls[1];
    if (a && b && a->seq > b->seq) std::swap(a, b);          // a: older, b: newer
    ctx->part_calls[0] = ctx->part_calls[1] = nullptr;
    delete b;
    delete a;
}
namespace {
// the slot of the next call, or -1 when two are in flight
int part_call_slot(nvca_ctx *ctx)
{
    const int parity = ctx->part_seq & 1;
    return ctx->part_calls[parity] ? -1 : parity;
}
} // namespace

extern "C" {

// One transform_frame_ip of every stream of the batch.  The streams' device work is queued together and waited for three
// times per call, however many streams there are: (1) the working images of all frames (a launch set per image size) and the
// face passes (an N-image job per kind of pass), (2) every part search in every face's region (FIND_BIGGEST searches that
// narrow their scan take one more round), (3) nothing -- the merging heuristics that follow are host code on the collected boxes.
int nvca_part_batch_process(nvca_ctx *ctx, int n, nvca_part_stream *const *streams, const nvca_frame *frames, nvca_rect *out_a, int cap_a,
                            int *n_a, nvca_rect *out_b, int cap_b, int *n_b)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (n < 0 || (n > 0 && (!streams || !frames || !n_a || !n_b)) || cap_a < 0 || cap_b < 0 || (cap_a > 0 && !out_a) || (cap_b > 0 && !out_b)) return NVCA_ERR_ARG;
    const int slot = part_call_slot(ctx);
    if (slot < 0) { ctx->set_error("two part batches are in flight: collect one first"); return NVCA_ERR_ARG; }
    // a stream of an outstanding ticket would have its back halves run out of order (this call's before the ticket's): refused
    for (void *o : ctx->part_calls)
        if (o)
            for (nvca_part_stream *s : ((PartCall *)o)->streams)
                for (int i = 0; i < n; i++)
                    if (streams[i] == s) { ctx->set_error("part stream has a submitted batch outstanding: collect it first"); return NVCA_ERR_ARG; }
    PartCall call; call.parity = slot;
    int rc = part_front(ctx, call, n, streams, frames);
    if (!rc) rc = part_back(ctx, call, out_a, cap_a, n_a, out_b, cap_b, n_b);
    return rc;
}
NVCA_API_CATCH(ctx)

// The same call in two halves (see part_front / part_back).  submit: gates, working images and face passes are queued, *ticket names
// the call; the frames must stay valid until the ticket is collected.  collect: the oldest outstanding ticket only (the streams' state
// machines advance in submit order).  At most two tickets are outstanding.  A failed collect rolls its streams back as a failed
// nvca_part_batch_process does -- and abandons a newer outstanding ticket with it (that ticket's streams are rolled back first).
int nvca_part_batch_submit(nvca_ctx *ctx, int n, nvca_part_stream *const *streams, const nvca_frame *frames, int *ticket)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (!ticket) return NVCA_ERR_ARG;
    const int slot = part_call_slot(ctx);
    if (slot < 0) { ctx->set_error("two part batches are in flight: collect one first"); return NVCA_ERR_ARG; }
    // a stream may not sit in two outstanding calls whose order the library cannot see: it may (tick k, tick k + 1), in submit order
    std::unique_ptr<PartCall> call(new PartCall());
    call->parity = slot; call->seq = ctx->part_seq;
    const int rc = part_front(ctx, *call, n, streams, frames);
    if (rc) return rc;                                   // (~PartCall rolls the gates back)
    ctx->part_calls_abandon = [](nvca_ctx *c) { part_calls_abandon_for(c, nullptr); };
    ctx->part_calls[slot] = call.release();
    *ticket = ctx->part_seq++;
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)
int nvca_part_batch_collect(nvca_ctx *ctx, int ticket, nvca_rect *out_a, int cap_a, int *n_a, nvca_rect *out_b, int cap_b, int *n_b)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    const int slot = ticket & 1, other = slot ^ 1;
    PartCall *call = ticket >= 0 ? (PartCall *)ctx->part_calls[slot] : nullptr;
    if (!call || call->seq != ticket) { ctx->set_error("unknown part-batch ticket"); return NVCA_ERR_ARG; }
    PartCall *newer = (PartCall *)ctx->part_calls[other];
    if (newer && newer->seq < ticket) { ctx->set_error("part-batch tickets are collected in submit order"); return NVCA_ERR_ARG; }
    const int rc = part_back(ctx, *call, out_a, cap_a, n_a, out_b, cap_b, n_b);
    if (rc && rc != NVCA_ERR_ARG) {                      // (bad output arguments: the ticket stays collectable)
        if (newer) { ctx->part_calls[other] = nullptr; delete newer; }       // its gates were taken on top of this call's: back first
        ctx->part_calls[slot] = nullptr; delete call;
        return rc;
    }
    if (rc) return rc;
    ctx->part_calls[slot] = nullptr; delete call;
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)

int nvca_part_stream_process(nvca_part_stream *s, const nvca_frame *f, nvca_rect *out_a, int cap_a, int *n_a, nvca_rect *out_b,
                             int cap_b, int *n_b)
try {
    if (!s || !f) return NVCA_ERR_ARG;
    nvca_part_stream *arr[1] = {s};
    return nvca_part_batch_process(s->ctx, 1, arr, f, out_a, cap_a, n_a, out_b, cap_b, n_b);
}
NVCA_API_CATCH((s ? s->ctx : nullptr))

} // extern "C"
