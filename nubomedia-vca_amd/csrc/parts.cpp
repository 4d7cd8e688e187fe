// parts.cpp -- NuboEyeDetector / NuboNoseDetector / NuboMouthDetector / NuboEarDetector stream objects.
// Host glue of kms_{eye,nose,mouth,ear}_detect_process_frame (EYE/kmseyedetect.cpp:915-1064,
// NOSE/kmsnosedetect.cpp:792-868, MOUTH/kmsmouthdetect.cpp:798-873, EAR/kmseardetect.cpp:644-729,767-812):
// ROI geometry, the per-frame / frame-to-frame merging heuristics and the frame gating are O(#faces) integer
// code and stay on the host; every cv:: call (cvtColor, equalizeHist, resize, flip, detectMultiScale) is the
// device implementation behind the public ABI, working on device-resident intermediates owned by the stream.
// std::vector idioms of the reference that rely on libstdc++ behaviour (erase through a reverse iterator,
// erase(end()-i) inside a counting loop) are written out as the index operations they perform.
#include "nvca_internal.h"
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
    DevBuf d_frame, d_gray, d_small, d_part, d_flip;
};

namespace {
inline int cv_round(double v)
{
    if (!(v > -2147483648.5 && v < 2147483647.5)) return INT_MIN;
    return (int)lrint(v);
}
inline int area(const nvca_rect &r) { return r.w * r.h; }

// One frame of one part stream on its way through a batched call: what the three phases hand to each other.
struct RoiJob { DetectJob *job = nullptr; nvca_rect roi{0, 0, 0, 0}; int side = 0; };
struct PartWork {
    nvca_part_stream *s = nullptr; const nvca_frame *f = nullptr;
    bool early_return = false, run = false;
    int W = 0, H = 0, fw = 0, fh = 0, pw = 0, ph = 0;
    double scale_o2f = 1, scale_x2o = 1, scale_f2x = 1;
    DetectJob *face_job = nullptr;           // the face pass (EAR: profile faces of the image and of its mirror)
    std::vector<RoiJob> rois;                // per face, in face order (EYE: right then left; EAR: side 0 then side 1)
    size_t n_side0 = 0;                      // EAR: how many of `rois` belong to side 0
    ~PartWork() { detect_job_free(face_job); for (RoiJob &r : rois) detect_job_free(r.job); }
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

void merge_consecutive_nm(RectV &cn, const RectV &old, const nvca_rect &face, int scale, int dis, RectV &res)
{   // __merge_noses_consecutives_frames NOSE/kmsnosedetect.cpp:745-790 (mouth :750-796 identical but for the distance)
    res.clear();
    for (const nvca_rect &o : old) {
        const int ocx = o.x + o.w / 2, ocy = o.y + o.h / 2;
        for (size_t j = 0; j < cn.size(); j++) {
            const int ncx = (cn[j].x + face.x) * scale + ((cn[j].w * scale) / 2);
            const int ncy = (cn[j].y + face.y) * scale + ((cn[j].h * scale) / 2);
            const double h2 = std::sqrt(std::pow((double)(ncx - ocx), 2) + std::pow((double)(ncy - ocy), 2));
            if (h2 < dis) { res.push_back(o); cn.erase(cn.begin() + j); break; }
        }
    }
    for (nvca_rect r : cn) {
        r.x = cv_round((face.x + r.x) * scale); r.y = cv_round((face.y + r.y) * scale);
        r.w = (r.w - 1) * scale; r.h = (r.h - 1) * scale;
        res.push_back(r);
    }
}

bool contain_bb(int px, int py, const nvca_rect &r) { return (py >= r.y && py <= r.y + r.h) && (px >= r.x && px <= r.x + r.w); }

void merge_eyes_current(const nvca_rect &face_bb, const RectV &eye_r, RectV &eyes, int scale, bool eye_left)
{   // __merge_eyes_current_frame EYE/kmseyedetect.cpp:778-862
    for (int i = (int)eyes.size() - 1; i > 0; i--) {
        int cx = eyes[i].x + eyes[i].w / 2, cy = eyes[i].y + eyes[i].h / 2;
        if (contain_bb(cx, cy, eyes[i - 1]) && area(eyes[i]) < area(eyes[i - 1])) eyes.erase(eyes.end() - i - 1);
        else {
            cx = eyes[i - 1].x + eyes[i - 1].w / 2; cy = eyes[i - 1].y + eyes[i - 1].h / 2;
            if (contain_bb(cx, cy, eyes[i]) && area(eyes[i - 1]) < area(eyes[i])) eyes.erase(eyes.end() - i);
        }
    }
    for (int i = (int)eyes.size() - 1; i >= 0; i--) {
        const int y_aux = face_bb.y * scale + face_bb.h * scale * 60 / 100;
        if (face_bb.y * scale + eyes[i].y < y_aux) {
            if (i == 0 && eyes.size() == 1) { if (!eye_r.empty() && eye_left) eyes[i].y = eye_r[0].y; }
            else eyes.erase(eyes.begin() + i);
        }
    }
    if (eyes.size() > 1) {
        const int middle_y = face_bb.x * scale + face_bb.h * scale / 2;      // sic (x / y swapped in the reference)
        const int middle_x = face_bb.y * scale + face_bb.w * scale / 2;
        for (int i = (int)eyes.size() - 1; i > 0; i--) {
            const int cy = eyes[i].y + eyes[i].h / 2, cx = eyes[i].x + eyes[i].w / 2;
            const int cy2 = eyes[i - 1].y + eyes[i - 1].h / 2, cx2 = eyes[i - 1].x + eyes[i - 1].w / 2;
            const float s1 = (float)std::sqrt(std::pow((double)(middle_x - cx), 2) + std::pow((double)(middle_y - cy), 2));
            const float s2 = (float)std::sqrt(std::pow((double)(middle_x - cx2), 2) + std::pow((double)(middle_y - cy2), 2));
            if (s1 < s2) eyes.erase(eyes.end() - i - 1); else eyes.erase(eyes.end() - i);
        }
    }
    if (eye_left && !eye_r.empty() && !eyes.empty()) eyes[0].y = eye_r[0].y;
}

void merge_eyes_consecutive(RectV &ce, const RectV &old, RectV &res)
{   // __merge_eyes_consecutives_frames EYE/kmseyedetect.cpp:864-900, DEFAULT_EUCLIDEAN_DIS 7
    res.clear();
    for (const nvca_rect &o : old) {
        const int ocx = o.x + o.w / 2, ocy = o.y + o.h / 2;
        for (size_t j = 0; j < ce.size(); j++) {
            const int ncx = ce[j].x + ce[j].w / 2, ncy = ce[j].y + ce[j].h / 2;
            const double h2 = std::sqrt(std::pow((double)(ncx - ocx), 2) + std::pow((double)(ncy - ocy), 2));
            if (h2 < 7) { res.push_back(o); ce.erase(ce.begin() + j); break; }
        }
    }
    res.insert(res.end(), ce.begin(), ce.end());
}

void to_global(RectV &v, const nvca_rect &face, int scale)
{   // transform_2_global_coordinates EYE/kmseyedetect.cpp:902-913
    for (nvca_rect &r : v) { r.x = (face.x + r.x) * scale; r.y = (face.y + r.y) * scale; r.w = (r.w - 1) * scale; r.h = (r.h - 1) * scale; }
}

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

extern "C" {

void nvca_part_params_default(nvca_part_params *p, int kind)
{
    if (!p) return;
    p->kind = kind; p->width_to_process = 320; p->process_x_every_4 = 4; p->scale_factor_pct = 25; p->detect_event = 0;
}

int nvca_part_stream_create(nvca_ctx *ctx, const nvca_part_params *params, const nvca_cascade *face, const nvca_cascade *a,
                            const nvca_cascade *b, nvca_part_stream **out)
{
    if (!ctx || !params || !face || !a || !out || params->kind < NVCA_PART_EYE || params->kind > NVCA_PART_EAR) return NVCA_ERR_ARG;
    if ((params->kind == NVCA_PART_EYE || params->kind == NVCA_PART_EAR) && !b) return NVCA_ERR_ARG;
    nvca_part_stream *s = new (std::nothrow) nvca_part_stream();
    if (!s) return NVCA_ERR_NOMEM;
    s->ctx = ctx; s->p = *params; s->face = face; s->a = a; s->b = b;
    *out = s;
    return NVCA_OK;
}
void nvca_part_stream_destroy(nvca_part_stream *s)
{
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    s->d_frame.release(); s->d_gray.release(); s->d_small.release(); s->d_part.release(); s->d_flip.release();
    delete s;
}
int nvca_part_stream_set_params(nvca_part_stream *s, const nvca_part_params *params)
{
    if (!s || !params || params->kind != s->p.kind) return NVCA_ERR_ARG;
    s->p = *params;
    return NVCA_OK;
}
int nvca_part_stream_push_faces(nvca_part_stream *s, const nvca_rect *faces, int n)
{
    if (!s || n < 0 || (n > 0 && !faces)) return NVCA_ERR_ARG;
    if (s->queue.size() < 16) s->queue.emplace_back(faces, faces + n);
    return NVCA_OK;
}

int nvca_part_stream_faces(const nvca_part_stream *s, nvca_rect *out, int cap, int *n_out)
{
    if (!s || !n_out || cap < 0 || (cap > 0 && !out)) return NVCA_ERR_ARG;
    *n_out = (int)s->faces.size();
    for (int i = 0; i < std::min(*n_out, cap); i++) out[i] = s->faces[i];
    return NVCA_OK;
}

// One transform_frame_ip of every stream of the batch.  The streams' device work is queued together and waited for three
// times per call, however many streams there are: (1) gray / equalize / resize / flip of every frame and every face pass,
// (2) every part search in every face's region (FIND_BIGGEST searches that narrow their scan take one more round),
// (3) nothing -- the merging heuristics that follow are host code on the collected boxes.
int nvca_part_batch_process(nvca_ctx *ctx, int n, nvca_part_stream *const *streams, const nvca_frame *frames, nvca_rect *out_a, int cap_a,
                            int *n_a, nvca_rect *out_b, int cap_b, int *n_b)
{
    NVCA_LOCK_OR_FAIL(ctx);
    if (n < 0 || (n > 0 && (!streams || !frames || !n_a || !n_b)) || cap_a < 0 || cap_b < 0 || (cap_a > 0 && !out_a) || (cap_b > 0 && !out_b)) return NVCA_ERR_ARG;
    for (int i = 0; i < n; i++) {
        const nvca_part_stream *s = streams[i]; const nvca_frame *f = &frames[i];
        if (!s || s->ctx != ctx || !f->data || f->width <= 0 || f->height <= 0 || f->stride < f->width * 3 || s->p.width_to_process <= 0) return NVCA_ERR_ARG;
        for (int j = 0; j < i; j++) if (streams[j] == s) { ctx->set_error("a part stream may appear once per batch"); return NVCA_ERR_ARG; }
    }
    (void)hipSetDevice(ctx->device);
    // the image primitives below hand device buffers to each other on the context's stream: no drain in between
    struct Defer { nvca_ctx *c; Defer(nvca_ctx *x) : c(x) { c->defer_device_sync++; } ~Defer() { c->defer_device_sync--; } } defer(ctx);
    std::vector<PartWork> work(n);
    std::vector<DetectJob *> jobs;
    std::vector<int> job_lane;
    // stream i's image chain and searches run on lane i mod kLanes: in order on that lane, side by side with the other lanes
    struct LaneGuard { nvca_ctx *c; ~LaneGuard() { c->cur_lane = 0; } } lane_guard{ctx};
    int rc = NVCA_OK;
    const int D = NVCA_MEM_DEVICE;
#define CK(e) do { if ((rc = (e))) return rc; } while (0)
    // ---- phase 1: gating, image chains and face passes of every stream, in stream order
    for (int i = 0; i < n; i++) {
        PartWork &w = work[i];
        ctx->cur_lane = n > 1 ? i % kLanes : 0;
        nvca_part_stream *s = w.s = streams[i]; const nvca_frame *f = w.f = &frames[i];
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
                    received = true;
                    s->num_frames_to_process = 10 / (5 - s->p.process_x_every_4);
                }
            }
            if (!received && s->num_frames_to_process <= 0) w.early_return = true;
        }
        if (w.early_return) continue;
        s->num_frame++;
        const int px = s->p.process_x_every_4;
        w.run = (2 == px && (1 == s->num_frame % 2)) || ((2 != px) && (s->num_frame <= px));
        if (!w.run) continue;
        s->num_frames_to_process--;
        const int fw = w.fw = cv_round(W / w.scale_o2f), fh = w.fh = cv_round(H / w.scale_o2f);
        const int pw = w.pw = cv_round(W / w.scale_x2o), ph = w.ph = cv_round(H / w.scale_x2o);
        if (fw <= 0 || fh <= 0 || pw <= 0 || ph <= 0) { ctx->set_error("part stream: frame too small"); return NVCA_ERR_ARG; }
        if (s->d_gray.ensure((size_t)W * H + 64) || s->d_small.ensure((size_t)fw * fh + 64) || s->d_part.ensure((size_t)pw * ph + 64) ||
            s->d_flip.ensure((size_t)fw * fh + 64)) { ctx->set_error("part stream: allocation failed"); return NVCA_ERR_NOMEM; }
        const void *src = f->data; const int sstride = f->stride;
        if (f->mem == NVCA_MEM_HOST) {
            if (s->d_frame.ensure((size_t)f->stride * H + 64)) { ctx->set_error("part stream: allocation failed"); return NVCA_ERR_NOMEM; }
            NVCA_HIP_CHECK(ctx, hipMemcpyAsync(s->d_frame.p, f->data, (size_t)f->stride * (H - 1) + (size_t)W * 3, hipMemcpyHostToDevice, ctx->cs()));
            src = s->d_frame.p;
        }
        uint8_t *gray = s->d_gray.as<uint8_t>(), *small = s->d_small.as<uint8_t>(), *part = s->d_part.as<uint8_t>();
        const double sf_face = 1 + s->p.scale_factor_pct * 1.0 / 100;
        CK(nvca_bgr2gray(ctx, src, W, H, sstride, 3, D, gray, W));
        if (kind == NVCA_PART_EYE) CK(nvca_equalize_hist(ctx, gray, W, H, W, D, gray, W));          // EYE :950
        if (kind == NVCA_PART_EAR) {
            CK(nvca_resize_linear(ctx, gray, W, H, W, 1, D, small, fw, fh, fw));
            CK(nvca_equalize_hist(ctx, small, fw, fh, fw, D, small, fw));
            CK(nvca_resize_linear(ctx, gray, W, H, W, 1, D, part, pw, ph, pw));
            CK(nvca_equalize_hist(ctx, part, pw, ph, pw, D, part, pw));
            CK(nvca_flip_horizontal(ctx, small, fw, fh, fw, D, s->d_flip.p, fw));                    // EAR :800
            // profile faces in the image and in its mirror: one launch set (EAR :656-659, :796-803)
            if (!(w.face_job = detect_job_new())) return NVCA_ERR_NOMEM;
            CK(make_detect_job(ctx, *w.face_job, s->face, small, fw, fh, fw, D, sf_face, 2, NVCA_HAAR_SCALE_IMAGE, 3, 3, fw, fh, false));
            detect_job_pair(w.face_job, s->d_flip.p);
        } else {
            if (0 == s->p.detect_event) {
                CK(nvca_resize_linear(ctx, gray, W, H, W, 1, D, small, fw, fh, fw));
                if (!(w.face_job = detect_job_new())) return NVCA_ERR_NOMEM;
                if (kind == NVCA_PART_EYE)
                    CK(make_detect_job(ctx, *w.face_job, s->face, small, fw, fh, fw, D, sf_face, 3, 0, 30, 30, 0, 0, false));
                else {
                    CK(nvca_equalize_hist(ctx, small, fw, fh, fw, D, small, fw));
                    CK(make_detect_job(ctx, *w.face_job, s->face, small, fw, fh, fw, D, sf_face, 2, NVCA_HAAR_SCALE_IMAGE, 3, 3, 0, 0, false));
                }
            }
            CK(nvca_resize_linear(ctx, gray, W, H, W, 1, D, part, pw, ph, pw));
            CK(nvca_equalize_hist(ctx, part, pw, ph, pw, D, part, pw));
        }
        if (w.face_job) { jobs.push_back(w.face_job); job_lane.push_back(ctx->cur_lane); }
    }
    ctx->cur_lane = 0;
    CK(run_detect_jobs(ctx, jobs.data(), (int)jobs.size(), job_lane.data()));          // wait 1: every face pass
    // ---- phase 2: the part searches of every face of every stream
    jobs.clear(); job_lane.clear();
    for (int i = 0; i < n; i++) {
        PartWork &w = work[i];
        if (!w.run) continue;
        nvca_part_stream *s = w.s;
        const int kind = s->p.kind;
        const uint8_t *part = s->d_part.as<uint8_t>();
        if (kind == NVCA_PART_EAR) {
            CK(find_ears_begin(s, w, detect_job_out(w.face_job, 0), part, s->a, 0));
            w.n_side0 = w.rois.size();
            CK(find_ears_begin(s, w, detect_job_out(w.face_job, 1), part, s->b, 1));
        } else {
            if (w.face_job) { const std::vector<nvca_rect> &fv = detect_job_out(w.face_job, 0); s->faces.assign(fv.begin(), fv.begin() + std::min<size_t>(fv.size(), 256)); }
            const double scale_f2x = w.scale_f2x;
            for (const nvca_rect &r : s->faces) {
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
        for (RoiJob &r : w.rois) if (r.job) { jobs.push_back(r.job); job_lane.push_back(n > 1 ? i % kLanes : 0); }
    }
    CK(run_detect_jobs(ctx, jobs.data(), (int)jobs.size(), job_lane.data()));          // wait 2 (+ one more for searches that narrowed)
#undef CK
    // ---- phase 3: merging heuristics, hysteresis, emission -- in stream order
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
            if (4 == s->num_frame) s->num_frame = 0;                                // GOP
        }
        n_a[i] = (int)s->la.size(); n_b[i] = (int)s->lb.size();
        for (int k = 0; k < std::min(n_a[i], cap_a); k++) out_a[(size_t)i * cap_a + k] = s->la[k];
        for (int k = 0; k < std::min(n_b[i], cap_b); k++) out_b[(size_t)i * cap_b + k] = s->lb[k];
    }
    return NVCA_OK;
}

int nvca_part_stream_process(nvca_part_stream *s, const nvca_frame *f, nvca_rect *out_a, int cap_a, int *n_a, nvca_rect *out_b,
                             int cap_b, int *n_b)
{
    if (!s || !f) return NVCA_ERR_ARG;
    nvca_part_stream *arr[1] = {s};
    return nvca_part_batch_process(s->ctx, 1, arr, f, out_a, cap_a, n_a, out_b, cap_b, n_b);
}

} // extern "C"
