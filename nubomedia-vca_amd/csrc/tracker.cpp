// tracker.cpp -- NuboTracker stream objects: replaces gst_nubo_tracker_img_conf +
// gst_nubo_tracker_process (TRK/gstnubotracker.cpp:202-237, 339-421).  Per stream the
// device keeps the previous gray frame and the motion-history image; every frame runs
// k_trk_pixel + the connected-component kernels, then __join_objects on the host.
#include "nvca_internal.h"
#include "host_logic.h"
#include <chrono>
#include <algorithm>
#include <cstring>

using namespace nvca;

struct nvca_tracker {
    nvca_ctx *ctx;
    nvca_tracker_params p;
    int w = 0, h = 0, num_frames = 0;
    DevBuf prev, mhi;
};

namespace {
constexpr int kCompCap = 1 << 18;
}

extern "C" {

int nvca_tracker_create(nvca_ctx *ctx, const nvca_tracker_params *params, nvca_tracker **out)
try {
    if (!ctx || !out) return NVCA_ERR_ARG;
    nvca_tracker *t = new (std::nothrow) nvca_tracker();
    if (!t) return NVCA_ERR_NOMEM;
    t->ctx = ctx;
    if (params) t->p = *params; else nvca_tracker_params_default(&t->p);
    *out = t;
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)

void nvca_tracker_destroy(nvca_tracker *t)
try {
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->lane_streams[kTrackerLane]);
    t->prev.release(); t->mhi.release();
    delete t;
}
NVCA_API_CATCH_VOID

int nvca_tracker_set_params(nvca_tracker *t, const nvca_tracker_params *params)
try {
    if (!t || !params) return NVCA_ERR_ARG;
    t->p = *params;
    return NVCA_OK;
}
NVCA_API_CATCH((t ? t->ctx : nullptr))

int nvca_tracker_batch_process(nvca_ctx *ctx, int n, nvca_tracker *const *trackers, const nvca_frame *frames,
                               const double *ts, nvca_rect *out, int cap, int *n_out)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (!ctx || n < 0 || (n > 0 && (!trackers || !frames || !ts || !n_out)) || cap < 0 || (cap > 0 && !out)) return NVCA_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    // the trackers' lane: a stream of its own, so the call neither queues behind a face batch in flight nor waits for it
    struct UseLane { nvca_ctx *c; int old; UseLane(nvca_ctx *x, int l) : c(x), old(x->cur_lane) { c->cur_lane = l; } ~UseLane() { c->cur_lane = old; } } use_lane(ctx, kTrackerLane);
    ctx->timer.tick(1);
    for (int i = 0; i < n; i++) {
        n_out[i] = 0;
        const nvca_frame &f = frames[i];
        if (!trackers[i] || trackers[i]->ctx != ctx || !f.data || f.width <= 0 || f.height <= 0 || f.stride < f.width * 4 ||
            (f.mem != NVCA_MEM_HOST && f.mem != NVCA_MEM_DEVICE)) return NVCA_ERR_ARG;
        for (int j = 0; j < i; j++) if (trackers[j] == trackers[i]) { ctx->set_error("a tracker may appear once per batch"); return NVCA_ERR_ARG; }
    }
    TrkWorkspace &ws = ctx->trk;
    const bool hostprof = ctx->sw.host_profile;
    auto tp0 = std::chrono::steady_clock::now(), tp1 = tp0, tp2 = tp0;
    int total_comps = 0;
    std::vector<char> done(n, 0);
    for (int i0 = 0; i0 < n; i0++) {
        if (done[i0]) continue;
        const int W = frames[i0].width, H = frames[i0].height;
        std::vector<int> idx;
        for (int j = i0; j < n; j++) if (!done[j] && frames[j].width == W && frames[j].height == H) { idx.push_back(j); done[j] = 1; }
        const int batch = (int)idx.size();
        const size_t N = (size_t)W * H;
        // per-stream state (gst_nubo_tracker_img_conf: MHI re-created zeroed on a size change)
        size_t stage_bytes = 0;
        bool any_ccl = false, vec4 = (W % 4) == 0;
        for (int b = 0; b < batch; b++) {
            nvca_tracker *t = trackers[idx[b]];
            if (t->w != W || t->h != H) {
                if (t->prev.ensure(N + 64) || t->mhi.ensure(N * sizeof(float) + 64)) { ctx->set_error("tracker state allocation failed"); return NVCA_ERR_NOMEM; }
                NVCA_HIP_CHECK(ctx, hipMemsetAsync(t->mhi.p, 0, N * sizeof(float), ctx->cs()));
                NVCA_HIP_CHECK(ctx, hipMemsetAsync(t->prev.p, 0, N, ctx->cs()));
                t->w = W; t->h = H;
            }
            const nvca_frame &f = frames[idx[b]];
            if (f.mem == NVCA_MEM_HOST) stage_bytes += ((size_t)f.stride * H + 255) / 256 * 256;
            if ((f.stride & 15) || (f.mem == NVCA_MEM_DEVICE && ((uintptr_t)f.data & 15))) vec4 = false;
            if (t->num_frames > 0) any_ccl = true;
        }
        static const int roots_div = [] { const char *e = getenv("NVCA_TRK_ROOTS_DIV"); const int v = e ? atoi(e) : 8; return v >= 1 ? v : 8; }();      // diagnostic: the share of the pixels the root list holds (1 / n)
        const int roots_cap = (int)std::min<size_t>(N * batch / roots_div, (size_t)1 << 30);
        const size_t tiles_per_slot = (size_t)((W + 255) / 256) * ((H + 7) / 8), tile_bytes = sizeof(int) * (2 * tiles_per_slot + 2) * (size_t)batch;
        const void *tiles_before = ws.tiles.p;
        if (ws.slots.ensure(sizeof(TrkSlot) * batch) || ws.h_slots.ensure(sizeof(TrkSlot) * batch) ||
            ws.labels.ensure(sizeof(int) * N * batch) || ws.acc.ensure(sizeof(CompAcc) * N * batch) || ws.flags.ensure(tracker_flag_bytes(W, H, batch) + 64) ||
            ws.out.ensure(sizeof(int) * (2 + 6 * (size_t)kCompCap)) || ws.h_out.ensure(sizeof(int) * (2 + 6 * (size_t)kCompCap)) ||
            ws.roots.ensure(sizeof(int) * (8 + (size_t)roots_cap + 64)) || ws.tiles.ensure(tile_bytes) ||
            (stage_bytes && ws.staging.ensure(stage_bytes))) { ctx->set_error("tracker workspace allocation failed"); return NVCA_ERR_NOMEM; }
        if (ws.tiles.p != tiles_before || ws.tiles_w != W || ws.tiles_h != H || ws.tiles_batch != batch || ws.tick >= 0x7ffffffe) {
            NVCA_HIP_CHECK(ctx, hipMemsetAsync(ws.tiles.p, 0, tile_bytes, ctx->cs()));       // marks of another layout (or of two billion ticks ago) mean nothing
            ws.tiles_w = W; ws.tiles_h = H; ws.tiles_batch = batch; ws.tick = 0;
        }
        const int tick = ++ws.tick;
        TrkSlot *hs = ws.h_slots.as<TrkSlot>();
        size_t off = 0;
        for (int b = 0; b < batch; b++) {
            nvca_tracker *t = trackers[idx[b]];
            const nvca_frame &f = frames[idx[b]];
            TrkSlot &s = hs[b];
            memset(&s, 0, sizeof(s));
            if (f.mem == NVCA_MEM_HOST) {
                uint8_t *d = ws.staging.as<uint8_t>() + off;
                if (int rc = caller_h2d(ctx, d, f.data, (size_t)f.stride * (H - 1) + (size_t)W * 4, ctx->cs())) return rc;
                s.src = d; off += ((size_t)f.stride * H + 255) / 256 * 256;
            } else s.src = (const uint8_t *)f.data;
            s.prev = t->prev.as<uint8_t>(); s.mhi = t->mhi.as<float>();
            const double timestamp = ts[idx[b]];
            s.ts = (float)timestamp; s.delbound = (float)(timestamp - t->p.mhi_duration);    // cvUpdateMotionHistory
            s.seg = (float)t->p.seg_thresh; s.threshold = t->p.threshold;
            s.has_prev = t->num_frames > 0; s.sstride = f.stride;
            s.min_area = t->p.min_area; s.max_area = (long long)t->p.max_area;
        }
        NVCA_HIP_CHECK(ctx, hipMemcpyAsync(ws.slots.p, hs, sizeof(TrkSlot) * batch, hipMemcpyHostToDevice, ctx->cs()));
        // (the result header and the root list's header are cleared by the pixel kernel; the live-segment counts only steer the per-pixel
        // kernels' visiting order)
        if (!ctx->sw.trk_fold) NVCA_HIP_CHECK(ctx, hipMemsetAsync(ws.flags.as<uint8_t>() + tracker_count_offset(W, H, batch), 0, sizeof(int) * (size_t)batch, ctx->cs()));
        const bool fold = ctx->sw.trk_fold;
        { TimedLaunch tl(ctx, NVCA_K_TRACKER);
          launch_tracker(ctx->cs(), ws.slots.p, batch, W, H, vec4, ws.labels.as<int>(), ws.acc.p, ws.out.as<int>(), kCompCap, any_ccl, ws.flags.as<uint8_t>(), ctx->sw.trk_order,
                         ws.roots.as<int>(), roots_cap, fold ? 0 : 1, ws.tiles.as<int>(), tick); }
        NVCA_LAUNCH_CHECK(ctx);
        tp1 = std::chrono::steady_clock::now();
        int *ho = ws.h_out.as<int>();
        int total = 0;
        if (any_ccl) {
            const int first = 1024;
            NVCA_HIP_CHECK(ctx, hipMemcpyAsync(ho, ws.out.p, sizeof(int) * (2 + 6 * (size_t)first), hipMemcpyDeviceToHost, ctx->cs()));
            NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->cs()));
            auto label_error = [&]() {
                int dbg[4] = {0, 0, 0, 0};
                (void)hipMemcpy(dbg, ws.roots.as<int>() + 2, 3 * sizeof(int), hipMemcpyDeviceToHost);
                ctx->set_error("internal: tracker label walk met an unwritten word (" + std::to_string(dbg[0]) + " times; first at " + std::to_string(dbg[1]) + ", kernel " + std::to_string(dbg[2]) + ")");
                return NVCA_ERR_INTERNAL;
            };
            if (ho[1] & 2) return label_error();
            if (fold && (ho[1] & 1)) {
                // more tile roots than the list holds (a frame of single-pixel components): the same motion history through the per-pixel kernels
                NVCA_HIP_CHECK(ctx, hipMemsetAsync(ws.out.p, 0, 2 * sizeof(int), ctx->cs()));
                NVCA_HIP_CHECK(ctx, hipMemsetAsync(ws.flags.as<uint8_t>() + tracker_count_offset(W, H, batch), 0, sizeof(int) * (size_t)batch, ctx->cs()));
                { TimedLaunch tl(ctx, NVCA_K_TRACKER);
                  launch_tracker(ctx->cs(), ws.slots.p, batch, W, H, vec4, ws.labels.as<int>(), ws.acc.p, ws.out.as<int>(), kCompCap, true, ws.flags.as<uint8_t>(), ctx->sw.trk_order,
                                 ws.roots.as<int>(), roots_cap, 2, ws.tiles.as<int>(), tick); }
                NVCA_LAUNCH_CHECK(ctx);
                NVCA_HIP_CHECK(ctx, hipMemcpyAsync(ho, ws.out.p, sizeof(int) * (2 + 6 * (size_t)first), hipMemcpyDeviceToHost, ctx->cs()));
                NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->cs()));
                if (ho[1] & 2) return label_error();
            }
            total = ho[0];
            if (total < 0) { ctx->set_error("internal: negative component count"); return NVCA_ERR_INTERNAL; }
            if (total > kCompCap) { ctx->set_error("tracker: more motion components than the list holds"); return NVCA_ERR_OVERFLOW; }
            if (total > first) {
                NVCA_HIP_CHECK(ctx, hipMemcpyAsync(ho + 2 + 6 * (size_t)first, ws.out.as<int>() + 2 + 6 * (size_t)first,
                                                   sizeof(int) * 6 * (size_t)(total - first), hipMemcpyDeviceToHost, ctx->cs()));
                NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->cs()));
            }
        } else
            NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->cs()));
        tp2 = std::chrono::steady_clock::now(); total_comps += total;
        // seed order (raster order of each component's first seed pixel) = cvSegmentMotion's output order
        std::vector<std::vector<std::pair<int, nvca_rect>>> comps(batch);
        for (int k = 0; k < total; k++) {
            const int *o = ho + 2 + (size_t)k * 6;
            if (o[0] < 0 || o[0] >= batch) { ctx->set_error("internal: motion component of an unknown slot (device result rejected)"); return NVCA_ERR_INTERNAL; }
            comps[o[0]].push_back({o[1], nvca_rect{o[2], o[3], o[4], o[5]}});
        }
        for (int b = 0; b < batch; b++) {
            nvca_tracker *t = trackers[idx[b]];
            if (t->num_frames > 0) {
                auto &c = comps[b];
                std::sort(c.begin(), c.end(), [](const std::pair<int, nvca_rect> &x, const std::pair<int, nvca_rect> &y) { return x.first < y.first; });
                std::vector<nvca_rect> sb;
                sb.reserve(c.size());
                for (auto &e : c) sb.push_back(e.second);
                join_objects(sb, t->p.min_area, t->p.max_area, t->p.distance);      // __join_objects :380
                n_out[idx[b]] = (int)sb.size();
                for (int k = 0; k < std::min<int>(cap, (int)sb.size()); k++) out[(size_t)idx[b] * cap + k] = sb[k];
            }
            t->num_frames++;
        }
    }
    if (hostprof) {
        auto tp3 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        fprintf(stderr, "[nvca host] tracker: enqueue %ld us, wait %ld us, host logic %ld us, %d components\n", us(tp0, tp1), us(tp1, tp2), us(tp2, tp3), total_comps);
    }
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)

int nvca_tracker_process(nvca_tracker *t, const nvca_frame *f, double ts, nvca_rect *out, int cap, int *n_out)
try {
    if (!t || !f) return NVCA_ERR_ARG;
    nvca_tracker *arr[1] = {t};
    return nvca_tracker_batch_process(t->ctx, 1, arr, f, &ts, out, cap, n_out);
}
NVCA_API_CATCH((t ? t->ctx : nullptr))

} // extern "C"
