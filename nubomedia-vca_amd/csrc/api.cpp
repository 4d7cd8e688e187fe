// api.cpp -- C ABI of libnubovca_hip (include/nubovca.h): context, workspace,
// launch sequencing, the stream objects that mirror the reference's elements.
#include "nvca_internal.h"
#include "plan.h"
#include "host_logic.h"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <climits>
#include <algorithm>
#include <fstream>
#include <sstream>
#include <chrono>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <stdexcept>
#include <new>
#include <thread>
#include <condition_variable>
#include <atomic>

using namespace nvca;

// =========================================================================
// buffers, timing
// =========================================================================
namespace nvca {

// NVCA_ALLOC_LOG=1 (diagnostic): every device allocation and release on stderr -- a GPU memory fault names an address, this says whose
static bool alloc_log() { static const bool on = getenv("NVCA_ALLOC_LOG") != nullptr; return on; }
// NVCA_ALLOC_GUARD=1 (diagnostic, "electric fence"): every device buffer is mapped through the virtual-memory API with an
// unmapped guard range before and behind it and ends (to 256 bytes) where its mapping ends, with no head-room: a kernel that
// reads or writes past a buffer faults at that access, every time, instead of now and then when the neighbouring pages happen
// to be unmapped.  Costs an allocation granule (2 MiB) per buffer; never on in production.
// NVCA_ALLOC_GUARD=1: released buffers stay mapped (leaked: a test run allocates a few GB in all); =2: they are unmapped and their
// address range stays reserved (a use after release faults too); =3: unmapped, released and the range freed.
static int alloc_guard_mode() { static const int m = getenv("NVCA_ALLOC_GUARD") ? std::max(1, atoi(getenv("NVCA_ALLOC_GUARD"))) : 0; return m; }
static bool alloc_guard() { return alloc_guard_mode() > 0; }
namespace {
struct GuardRec { void *va; size_t total, mapped, lead; hipMemGenericAllocationHandle_t h; };
std::map<void *, GuardRec> g_guard;
std::mutex g_guard_mu;
hipError_t guard_alloc(void **out, size_t n)
{
    int dev = 0; (void)hipGetDevice(&dev);
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess || gran == 0) return e != hipSuccess ? e : hipErrorUnknown;
    const size_t body = (n + 255) & ~(size_t)255, mapped = (body + gran - 1) / gran * gran;
    GuardRec r; r.total = mapped + 2 * gran; r.mapped = mapped; r.lead = gran; r.va = nullptr;
    if ((e = hipMemAddressReserve(&r.va, r.total, gran, nullptr, 0)) != hipSuccess) return e;
    if ((e = hipMemCreate(&r.h, mapped, &prop, 0)) != hipSuccess) { (void)hipMemAddressFree(r.va, r.total); return e; }
    char *base = (char *)r.va + gran;
    if ((e = hipMemMap(base, mapped, 0, r.h, 0)) != hipSuccess) { (void)hipMemRelease(r.h); (void)hipMemAddressFree(r.va, r.total); return e; }
    hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(base, mapped, &acc, 1)) != hipSuccess) { (void)hipMemUnmap(base, mapped); (void)hipMemRelease(r.h); (void)hipMemAddressFree(r.va, r.total); return e; }
    *out = base + (mapped - body);                       // the buffer ends where the mapping ends
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard[*out] = r;
    return hipSuccess;
}
void guard_free(void *p)
{
    GuardRec r;
    { std::lock_guard<std::mutex> lk(g_guard_mu);
      auto it = g_guard.find(p);
      if (it == g_guard.end()) { (void)hipFree(p); return; }
      r = it->second; g_guard.erase(it); }
    (void)hipDeviceSynchronize();
    if (alloc_guard_mode() >= 2) (void)hipMemUnmap((char *)r.va + r.lead, r.mapped);
    if (alloc_guard_mode() >= 3) { (void)hipMemRelease(r.h); (void)hipMemAddressFree(r.va, r.total); }
}
}
int DevBuf::ensure(size_t n)
{
    if (n <= bytes) return 0;
    static bool guard_broken = false;          // the runtime refused the virtual-memory calls: said once, plain allocations from then on
    if (alloc_guard() && !guard_broken) {
        if (p) { (void)hipDeviceSynchronize(); if (alloc_log()) fprintf(stderr, "[nvca alloc] free  %p (%zu bytes, grows)\n", p, bytes); guard_free(p); p = nullptr; bytes = 0; }
        const hipError_t ge = guard_alloc(&p, n);
        if (ge == hipSuccess) {
            bytes = n;
            if (alloc_log()) fprintf(stderr, "[nvca alloc] alloc %p .. %p (%zu bytes, guarded)\n", p, (void *)((char *)p + n), n);
            return 0;
        }
        (void)hipGetLastError(); p = nullptr; bytes = 0; guard_broken = true;
        fprintf(stderr, "[nvca alloc] guard unavailable (%s): plain allocations\n", hipGetErrorString(ge));
    }
    if (p) { (void)hipDeviceSynchronize(); if (alloc_log()) fprintf(stderr, "[nvca alloc] free  %p (%zu bytes, grows)\n", p, bytes); (void)hipFree(p); p = nullptr; bytes = 0; }
    size_t want = n + n / 4;                                  // head-room: batches grow
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; e = hipMalloc(&p, n); want = n; }    // the refused head-room attempt must not surface later as a launch error
    if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; bytes = 0; return (int)e; }
    bytes = want;
    if (alloc_log()) fprintf(stderr, "[nvca alloc] alloc %p .. %p (%zu bytes, %zu asked)\n", p, (void *)((char *)p + want), want, n);
    return 0;
}
void DevBuf::release() { if (p && bytes) { if (alloc_log()) fprintf(stderr, "[nvca alloc] free  %p (%zu bytes)\n", p, bytes); if (alloc_guard()) guard_free(p); else (void)hipFree(p); } p = nullptr; bytes = 0; }     // bytes == 0: a view into another buffer
int PinnedBuf::ensure(size_t n)
{
    if (n <= bytes) return 0;
    if (p) { (void)hipDeviceSynchronize(); (void)hipHostFree(p); p = nullptr; bytes = 0; }
    hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; return (int)e; }
    bytes = n;
    return 0;
}
void PinnedBuf::release() { if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; } }

static thread_local TimedLaunch *g_scope = nullptr;
TimedLaunch::TimedLaunch(nvca_ctx *c, int kind) : ctx(c), k(kind)
{
    if (!ctx->timer.on || !ctx->timer.sample) return;
    active = true; prev = g_scope; g_scope = this;
}
TimedLaunch::~TimedLaunch()
{
    if (active) g_scope = prev;
}
static thread_local hipError_t g_launch_err = hipSuccess;
static thread_local const char *g_launch_kernel = nullptr;
void note_launch(const char *kernel)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && g_launch_err == hipSuccess) { g_launch_err = e; g_launch_kernel = kernel; }
}
hipError_t take_launch_error(const char **kernel)
{
    const hipError_t e = g_launch_err;
    if (kernel) *kernel = g_launch_kernel;
    g_launch_err = hipSuccess; g_launch_kernel = nullptr;
    return e;
}
bool launch_events(hipEvent_t *a, hipEvent_t *b)
{
    TimedLaunch *sc = g_scope;
    if (!sc) return false;
    KernelTimer &t = sc->ctx->timer;
    auto get = [&]() {
        hipEvent_t e = nullptr;
        if (!t.pool.empty()) { e = t.pool.back(); t.pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    *a = get(); *b = get();
    if (!*a || !*b) return false;
    t.pending.push_back(KernelTimer::Ev{*a, *b, sc->k, sc->n++ == 0});
    return true;
}
static void drain_timer_now(nvca_ctx *ctx)
{
    KernelTimer &t = ctx->timer;
    std::vector<KernelTimer::Ev> later;
    bool stop = false;
    for (auto &e : t.pending) {
        float ms = 0;
        // kernels finish in launch order: after the first pair that is not ready (a batch still in flight between
        // submit and collect) nothing later is either, and asking again for every one of them is not free
        if (stop || hipEventQuery(e.b) == hipErrorNotReady) { stop = true; later.push_back(e); continue; }
        const hipError_t r = hipEventElapsedTime(&ms, e.a, e.b);
        if (r == hipErrorNotReady) { stop = true; later.push_back(e); continue; }
        if (r == hipSuccess) { t.total_ms[e.k] += ms; if (e.first) t.launches[e.k]++; }
        t.pool.push_back(e.a); t.pool.push_back(e.b);
    }
    t.pending.swap(later);
}
// Event pairs are turned into times when somebody asks (nvca_ctx_kernel_timing) or when many have piled up: querying them
// after every batch costs more than it looks while another batch is executing.
static void drain_timer(nvca_ctx *ctx)
{
    if (ctx->timer.pending.size() > 4096) drain_timer_now(ctx);
}

// handler of the ABI's function-try-blocks (NVCA_API_CATCH): called inside a catch (...) clause
int api_catch(nvca_ctx *ctx) noexcept
{
    int code = NVCA_ERR_INTERNAL;
    const char *what = "unknown exception";
    char buf[160];
    try { throw; }
    catch (const std::bad_alloc &) { code = NVCA_ERR_NOMEM; what = "out of host memory (std::bad_alloc)"; }
    catch (const std::length_error &e) { code = NVCA_ERR_NOMEM; snprintf(buf, sizeof(buf), "container size limit exceeded (%s)", e.what()); what = buf; }
    catch (const std::exception &e) { snprintf(buf, sizeof(buf), "internal error: %s", e.what()); what = buf; }
    catch (...) { }
    if (ctx) {
        try { std::lock_guard<std::recursive_mutex> lk(ctx->mu); ctx->err.assign(what); } catch (...) { }
    }
    return code;
}

static Switches read_switches()
{
    Switches w;
    auto num = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    auto set = [](const char *name) { return getenv(name) != nullptr; };
    w.group_zero_copy = num("NVCA_GROUP_ZEROCOPY", 1) != 0;
    w.skip_cascade = set("NVCA_SKIP_CASCADE");
    w.host_group = set("NVCA_HOST_GROUP");
    w.band_map = num("NVCA_BAND_MAP", 0);
    w.band = num("NVCA_BAND", -1);
    w.host_profile = set("NVCA_HOST_PROFILE");
    w.sparse_ingest = num("NVCA_SPARSE_INGEST", 1) != 0;
    w.pyr_off = set("NVCA_PYR_OFF");
    if (set("NVCA_PART_STATS")) { const int n = num("NVCA_PART_STATS", 0); w.part_stats = n > 0 ? n : 8; }
    w.ingest_chunk = num("NVCA_INGEST_CHUNK", 8);
    w.stage_order = num("NVCA_STAGE_ORDER", 0) != 0;
    w.trk_fold = num("NVCA_TRK_FOLD", 1) != 0;
    w.spec_pairs = std::max(1, num("NVCA_SPEC_PAIRS", 1536));
    w.pair_max = num("NVCA_PAIR_MAX", 32);
    w.deep_stage = set("NVCA_DEEP_STAGE") ? std::max(1, num("NVCA_DEEP_STAGE", 0)) : 0;
    w.tiles = num("NVCA_TILES", 1) != 0;
    w.plan_debug = set("NVCA_PLAN_DEBUG");
    w.deep_lds = !set("NVCA_DEEP_LDS_OFF");
    w.trk_order = num("NVCA_TRK_ORDER", -1);
    w.host_threads = num("NVCA_HOST_THREADS", -1);
    w.two_lanes = num("NVCA_TWO_LANES", 1) != 0;
    w.roi = num("NVCA_ROI", 1) != 0;
    w.fb_dense = num("NVCA_FB_DENSE", 1) != 0;
    w.pre_cus = num("NVCA_PRE_CUS", 0);
    w.quiet = set("NVCA_QUIET");
    w.stamps_out = getenv("NVCA_STAMPS_OUT");
    return w;
}
const Switches &switches()
{
    static const Switches w = read_switches();      // first use: nvca_ctx_create
    return w;
}

static inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cv_round(double v)
{
    if (!(v > -2147483648.5 && v < 2147483647.5)) return INT_MIN;   // _mm_cvtsd_si32 on overflow / inf
    return (int)lrint(v);
}

// what a cascade job leaves behind for the host: candidate list, box table, thresholds.  Three sets: [0] the synchronous
// entry points, [1] / [2] the two batches that may be in flight through nvca_face_batch_submit / _collect
struct ResultBufs {
    DevBuf hits, grp, gthr, staging, srcptrs;        // staging / srcptrs: host frames on their way in, frame pointer table
    PinnedBuf h_hits, h_grp, h_gthr, h_srcptrs;   // h_srcptrs: frame pointers on their way to the device array
    std::vector<int> gthr_last;       // thresholds currently resident in gthr
    void release() { hits.release(); grp.release(); gthr.release(); staging.release(); srcptrs.release(); h_hits.release(); h_grp.release(); h_gthr.release(); h_srcptrs.release(); gthr_last.clear(); }
};
// Working memory of the kernels, per LANE.  A lane is a HIP stream with its own planes and cascade scratch: whatever runs on a
// lane is ordered by its stream, different lanes run side by side.  Everything uses lane 0 (the context's stream) except
// the batched part detectors, which spread their streams' small, launch-bound jobs over all lanes (parts.cpp): the GPU then
// holds several of those tiny kernels at a time instead of one.  What a job leaves for the host lives in ResultBufs regions
// of its own, shared by all lanes.
struct Lane {
    DevBuf gray, hist, lut, bandsum, bandsq, sum, sqsum, tilted, staging, aux, failbits, vnf, deep;
    int hist_clean = 0;               // leading histogram slots known to be all zero
    void release_all()
    {
        gray.release(); hist.release(); lut.release(); bandsum.release(); bandsq.release(); sum.release();
        sqsum.release(); tilted.release(); staging.release(); aux.release();
        failbits.release(); vnf.release(); deep.release();
    }
};
struct Workspace {
    Lane lanes[kLanes];
    int *cur_lane = nullptr;          // the context's current lane index
    Lane &ln() { return lanes[*cur_lane]; }
    ResultBufs res[3];
    int cur_res = 0;
    void release_all()
    {
        for (Lane &l : lanes) l.release_all();
        for (ResultBufs &r : res) r.release();
    }
};

// one cached geometry: source frame -> working image -> scan tables
struct PyrLevel { double f; int szw, szh, winw, winh; size_t gray_off; int gpitch; int plane_off; };
// Source rows a shrinking bilinear resize reads, when they form equal runs at a fixed period (integer ratios: a 1080p
// frame shrunk by 12 reads rows 12k + 5 and 12k + 6 only).  Host frames then cross PCIe as one strided 2-D copy of those
// rows -- into their natural places of the staged frame, so the kernels are unchanged -- instead of whole.
struct RowCopy {
    bool on = false;
    int first = 0, period = 0, run = 0, count = 0;
};
static RowCopy make_rowcopy(const ResizeTab &t)
{
    RowCopy rc;
    if (t.mode != 1 || t.dh <= 0) return rc;
    std::vector<int> rows;
    auto clampr = [&](int r) { return r >= 0 ? (r < t.sh ? r : t.sh - 1) : 0; };
    for (int dy = 0; dy < t.dh; dy++) { rows.push_back(clampr(t.yofs[dy])); rows.push_back(clampr(t.yofs[dy] + 1)); }
    std::sort(rows.begin(), rows.end());
    rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
    if (rows.size() * 2 > (size_t)t.sh) return rc;              // no saving worth a strided copy
    std::vector<std::pair<int, int>> runs;                          // maximal runs of consecutive rows
    for (int r : rows) { if (!runs.empty() && runs.back().first + runs.back().second == r) runs.back().second++; else runs.push_back({r, 1}); }
    const int period = runs.size() > 1 ? runs[1].first - runs[0].first : t.sh;
    for (size_t i = 0; i < runs.size(); i++)
        if (runs[i].second != runs[0].second || runs[i].first != runs[0].first + (int)i * period) return rc;
    rc.on = true; rc.first = runs[0].first; rc.period = period; rc.run = runs[0].second; rc.count = (int)runs.size();
    return rc;
}

struct GeomPlan {
    ResizeTab tab;
    RowCopy rowcopy;
    DevBuf d_xofs, d_yofs, d_ialpha, d_ibeta;
    DetectPlan det;
    PreGeom g;
    bool has_det = false;
    uint64_t last_use = 0;                                    // plan cache is LRU-bounded (store_plan)
    int inflight = 0;                                         // batches in flight that reference this plan: never evicted
    // CV_HAAR_SCALE_IMAGE: pyramid levels, their resize tables, plane layout
    std::vector<PyrLevel> lv;
    std::vector<std::unique_ptr<GeomPlan>> level_tabs;
    size_t gray_total = 0, plane_total = 0; int P = 0;
    std::vector<int> fb_ladder;                               // FIND_BIGGEST: ladder position of each scale of the full-grid plan
    DevBuf d_pyr, d_level_tabs; int pyr_maxw = 0, pyr_maxh = 0; bool pyr_ok = false;   // device level table (one-launch pyramid kernels)
    ~GeomPlan() { level_tabs.clear(); d_xofs.release(); d_yofs.release(); d_ialpha.release(); d_ibeta.release(); d_pyr.release(); d_level_tabs.release(); }
};

DetectPlan::~DetectPlan()
{
    release_tables();
    d_scales.release(); d_stages.release(); d_strips.release(); d_pos.release(); d_order.release(); d_tasks.release(); d_tiles.release(); d_tile_order.release(); d_tcoords.release(); d_bands.release(); d_band_order.release(); d_deeprecs.release(); d_stage_hint.release(); d_stage_first.release(); d_stage_thr.release(); d_blob.release();
}

int DetectPlan::upload(nvca_ctx *ctx)
{
    struct Item { DevBuf *d; const void *h; size_t n; } items[] = {
        {&d_scales, scales.data(), scales.size() * sizeof(ScaleRec)},
        {&d_stages, stages.data(), stages.size() * sizeof(StageRec)},
        {&d_strips, strips.data(), strips.size() * sizeof(StripRec)},
        {&d_pos, pos.data(), pos.size() * sizeof(int)},
        {&d_order, order.data(), order.size() * sizeof(int)},
        {&d_tasks, tasks.data(), tasks.size() * sizeof(unsigned)},
        {&d_tiles, tiles.data(), tiles.size() * sizeof(TileRec)},
        {&d_tile_order, tile_order.data(), tile_order.size() * sizeof(int)},
        {&d_tcoords, tcoords.data(), tcoords.size() * sizeof(unsigned short)},
        {&d_bands, bands.data(), bands.size() * sizeof(BandRec)},
        {&d_band_order, band_order.data(), band_order.size() * sizeof(int)},
        {&d_deeprecs, deeprecs.data(), deeprecs.size() * sizeof(DeepRec)},
        {&d_stage_hint, nullptr, tiles.empty() ? (size_t)0 : 8 * sizeof(int)},          // zero: nothing known yet
        {&d_stage_first, stage_first.data(), stage_first.size() * sizeof(int)},
        {&d_stage_thr, stage_thr.data(), stage_thr.size() * sizeof(float)},
    };
    // one device allocation and one copy for all tables (a FIND_BIGGEST scan builds a plan per scale, per call)
    size_t total = 0;
    for (auto &it : items) total += (it.n + 255) & ~(size_t)255;
    if (total == 0) return NVCA_OK;
    std::vector<unsigned char> blob(total);
    size_t off = 0;
    for (auto &it : items) { if (it.n && it.h) memcpy(blob.data() + off, it.h, it.n); off += (it.n + 255) & ~(size_t)255; }
    for (auto &it : items) it.d->release();
    if (d_blob.ensure(total)) { ctx->set_error("hipMalloc failed for plan tables"); return NVCA_ERR_NOMEM; }
    NVCA_HIP_CHECK(ctx, hipMemcpy(d_blob.p, blob.data(), total, hipMemcpyHostToDevice));
    off = 0;
    for (auto &it : items) {
        it.d->p = it.n ? (unsigned char *)d_blob.p + off : nullptr; it.d->bytes = 0;       // views
        off += (it.n + 255) & ~(size_t)255;
    }
    return NVCA_OK;
}

static void make_geom(PreGeom &g, int sw, int sh, int sstride, int cn, int w, int h)
{
    memset(&g, 0, sizeof(g));
    g.sw = sw; g.sh = sh; g.sstride = sstride; g.cn = cn;
    g.w = w; g.h = h;
    g.gpitch = (int)round_up(w, 64);
    g.spitch = (int)round_up(w + 1, 8);
    g.nbands = (h + kIntegralBand - 1) / kIntegralBand;
    g.gray_slot = round_up((size_t)g.gpitch * h, 256);
    g.sum_slot = round_up((size_t)g.spitch * (h + 1), 64);
    g.band_slot = (size_t)g.nbands * round_up(w, 8);
}

static int ensure_ws(nvca_ctx *ctx, const PreGeom &g, int batch)
{
    Workspace &ws = *ctx->ws;
    int e = 0;
    e |= ws.ln().gray.ensure(g.gray_slot * batch + 64);
    { void *old = ws.ln().hist.p; e |= ws.ln().hist.ensure((size_t)batch * 256 * sizeof(unsigned)); if (ws.ln().hist.p != old) ws.ln().hist_clean = 0; }
    e |= ws.ln().lut.ensure((size_t)batch * 256);
    e |= ws.ln().bandsum.ensure(g.band_slot * batch * sizeof(unsigned));
    e |= ws.ln().bandsq.ensure(g.band_slot * batch * sizeof(unsigned));
    e |= ws.ln().sum.ensure((g.sum_slot * batch + 4 * (size_t)g.spitch) * sizeof(int));      // a few spare rows: a scaled feature corner may round past the window by a pixel or two
    e |= ws.ln().sqsum.ensure(g.sum_slot * batch * sizeof(unsigned long long));
    e |= ws.res[ws.cur_res].srcptrs.ensure((size_t)batch * sizeof(void *));
    e |= ws.res[ws.cur_res].h_srcptrs.ensure((size_t)batch * sizeof(void *));
    if (e) { ctx->set_error("device/pinned allocation failed for the workspace"); return NVCA_ERR_NOMEM; }
    return NVCA_OK;
}

static int upload_tab(nvca_ctx *ctx, GeomPlan &gp)
{
    const ResizeTab &t = gp.tab;
    if (t.mode != 1) return NVCA_OK;
    if (gp.d_xofs.ensure(t.xofs.size() * 4) || gp.d_yofs.ensure(t.yofs.size() * 4) ||
        gp.d_ialpha.ensure(t.ialpha.size() * 2) || gp.d_ibeta.ensure(t.ibeta.size() * 2)) {
        ctx->set_error("hipMalloc failed for resize tables"); return NVCA_ERR_NOMEM;
    }
    NVCA_HIP_CHECK(ctx, hipMemcpy(gp.d_xofs.p, t.xofs.data(), t.xofs.size() * 4, hipMemcpyHostToDevice));
    NVCA_HIP_CHECK(ctx, hipMemcpy(gp.d_yofs.p, t.yofs.data(), t.yofs.size() * 4, hipMemcpyHostToDevice));
    NVCA_HIP_CHECK(ctx, hipMemcpy(gp.d_ialpha.p, t.ialpha.data(), t.ialpha.size() * 2, hipMemcpyHostToDevice));
    NVCA_HIP_CHECK(ctx, hipMemcpy(gp.d_ibeta.p, t.ibeta.data(), t.ibeta.size() * 2, hipMemcpyHostToDevice));
    return NVCA_OK;
}

// the resize tables of several levels in one allocation (owned by `blob`) and one copy; the levels' buffers become views
static int upload_tabs(nvca_ctx *ctx, std::vector<std::unique_ptr<GeomPlan>> &levels, DevBuf &blob)
{
    auto al = [](size_t n) { return (n + 255) & ~(size_t)255; };
    size_t total = 0;
    for (auto &gp : levels) { const ResizeTab &t = gp->tab; if (t.mode != 1) continue; total += al(t.xofs.size() * 4) + al(t.yofs.size() * 4) + al(t.ialpha.size() * 2) + al(t.ibeta.size() * 2); }
    if (!total) return NVCA_OK;
    std::vector<unsigned char> h(total);
    if (blob.ensure(total)) { ctx->set_error("hipMalloc failed for resize tables"); return NVCA_ERR_NOMEM; }
    size_t off = 0;
    auto put = [&](DevBuf &d, const void *src, size_t n) { memcpy(h.data() + off, src, n); d.release(); d.p = (unsigned char *)blob.p + off; d.bytes = 0; off += al(n); };
    for (auto &gp : levels) {
        const ResizeTab &t = gp->tab;
        if (t.mode != 1) continue;
        put(gp->d_xofs, t.xofs.data(), t.xofs.size() * 4); put(gp->d_yofs, t.yofs.data(), t.yofs.size() * 4);
        put(gp->d_ialpha, t.ialpha.data(), t.ialpha.size() * 2); put(gp->d_ibeta, t.ibeta.data(), t.ibeta.size() * 2);
    }
    NVCA_HIP_CHECK(ctx, hipMemcpy(blob.p, h.data(), total, hipMemcpyHostToDevice));
    return NVCA_OK;
}

// ---- launch sequences ----------------------------------------------------

// integral planes for `batch` slots (lut == nullptr -> identity); gray / sum / sq default to the workspace planes
static void run_integral(nvca_ctx *ctx, const PreGeom &g, const uint8_t *lut, int batch, const uint8_t *gray = nullptr,
                         int *sum = nullptr, unsigned long long *sq = nullptr)
{
    Workspace &ws = *ctx->ws;
    if (!gray) gray = ws.ln().gray.as<uint8_t>();
    if (!sum) sum = ws.ln().sum.as<int>();
    if (!sq) sq = ws.ln().sqsum.as<unsigned long long>();
    if (batch <= 64 && small_integral_fits(g)) {         // small images (ROI searches and working images of the part detectors): one launch, a workgroup per image
        TimedLaunch t(ctx, NVCA_K_INTEGRAL);
        launch_small_integral(ctx->cs(), gray, lut, 256, g, sum, sq, batch);
        return;
    }
    { TimedLaunch t(ctx, NVCA_K_COLSUM);
      launch_colsum(ctx->cs(), gray, lut, 256, g, ws.ln().bandsum.as<unsigned>(), ws.ln().bandsq.as<unsigned>(), batch); }
    { TimedLaunch t(ctx, NVCA_K_BANDSCAN);
      launch_bandscan(ctx->cs(), g, ws.ln().bandsum.as<unsigned>(), ws.ln().bandsq.as<unsigned>(), batch); }
    { TimedLaunch t(ctx, NVCA_K_INTEGRAL);
      launch_integral(ctx->cs(), gray, lut, 256, g, ws.ln().bandsum.as<unsigned>(), ws.ln().bandsq.as<unsigned>(), sum, sq, batch); }
}

// tilted integral planes for `batch` slots (cascades with tilted features only); same geometry and equalisation LUT as run_integral
static int run_tilted(nvca_ctx *ctx, const PreGeom &g, const uint8_t *lut, int batch, const uint8_t *gray = nullptr, int *tilted = nullptr)
{
    Workspace &ws = *ctx->ws;
    if (g.w + 1 > 8 * 1024 || (size_t)2 * (g.w + g.h + 2) * sizeof(int) > 64 * 1024) { ctx->set_error("image too large for the tilted integral"); return NVCA_ERR_ARG; }
    if (!tilted) {
        if (ws.ln().tilted.ensure((g.sum_slot * batch + 4 * (size_t)g.spitch) * sizeof(int))) { ctx->set_error("device allocation failed (tilted integral)"); return NVCA_ERR_NOMEM; }
        tilted = ws.ln().tilted.as<int>();
    }
    if (!gray) gray = ws.ln().gray.as<uint8_t>();
    TimedLaunch t(ctx, NVCA_K_INTEGRAL);
    launch_tilted(ctx->cs(), gray, lut, 256, g, tilted, batch);
    return NVCA_OK;
}

// cascade scan over the integral planes of slots [0, n); fills raw[b] (canonical scale,y,x order)
// group_thr (optional, [n]): cv::groupRectangles thresholds; when given and the plan allows it the grouping
// runs on the device (k_group) and raw[b] comes back already grouped -- grouped[b] says which.
// A job owns a result region (`r0` = index of its first frame in the caller's batch of `total` frames): its candidate
// list and box table stay untouched while later jobs are enqueued, so several jobs can be queued before one sync.
static constexpr int kMaxHitCap = 1 << 22;   // raw candidates per frame the lists are ever sized for (nvca_ctx_set_hit_capacity's limit)
static constexpr int kGroupOutCap = 64;      // final boxes per frame returned by k_group (more -> host grouping)
struct CascadeJob {
    int r0 = 0, n = 0, total = 0;
    bool dev_group = false;
    bool counters_zeroed = false;   // the caller's k_lut launch reset the two list counters (cascade_counters())
    unsigned cap = 0;
    size_t first = 0;         // raw candidates fetched with the count (raw mode)
    unsigned long long *d_hits = nullptr, *h_hits = nullptr;
    int *d_grp = nullptr, *h_grp = nullptr;
};

// the two list counters a job's kernels append to (so that the caller's k_lut launch can reset them); sizes the lists
static int cascade_counters(nvca_ctx *ctx, DetectPlan &dp, const CascadeJob &job, unsigned long long **hits, unsigned long long **deep)
{
    Workspace &ws = *ctx->ws;
    ResultBufs &rb = ws.res[ws.cur_res];
    const int total = std::max(job.total, job.r0 + job.n);
    const size_t hits_stride = (size_t)ctx->hit_cap + 1;
    const unsigned deep_cap = (unsigned)std::min<size_t>((size_t)dp.tasks.size() * 64 * job.n + 64, 1u << 28);
    if (ws.ln().deep.ensure(((size_t)deep_cap + 1) * sizeof(unsigned long long)) ||
        rb.hits.ensure(hits_stride * total * sizeof(unsigned long long)) || rb.h_hits.ensure(hits_stride * total * sizeof(unsigned long long))) {
        ctx->set_error("device allocation failed for the cascade workspace"); return NVCA_ERR_NOMEM;
    }
    *hits = rb.hits.as<unsigned long long>() + hits_stride * job.r0;
    *deep = ws.ln().deep.as<unsigned long long>();
    return NVCA_OK;
}

static int cascade_enqueue(nvca_ctx *ctx, DetectPlan &dp, size_t sum_slot, int spitch, CascadeJob &job, const int *group_thr, bool want_group,
                           hipEvent_t early_done = nullptr /* recorded behind the band / tile kernels, ahead of the late stages */)
{
    Workspace &ws = *ctx->ws;
    ResultBufs &rb = ws.res[ws.cur_res];
    const bool grp_zero_copy = ctx->sw.group_zero_copy;
    const int batch = job.n, total = std::max(job.total, job.r0 + job.n);
    const size_t hits_stride = (size_t)ctx->hit_cap + 1;                 // u64 words per result slot
    const unsigned cap = (unsigned)ctx->hit_cap * (unsigned)batch;
    const unsigned deep_cap = (unsigned)std::min<size_t>((size_t)dp.tasks.size() * 64 * batch + 64, 1u << 28);   // every window may survive
    if (ws.ln().failbits.ensure(dp.tasks.size() * sizeof(unsigned long long) * batch + 8) ||
        ws.ln().vnf.ensure(dp.tasks.size() * 64 * sizeof(double) * batch + 8) ||
        ws.ln().deep.ensure(((size_t)deep_cap + 1) * sizeof(unsigned long long)) ||
        rb.hits.ensure(hits_stride * total * sizeof(unsigned long long)) || rb.h_hits.ensure(hits_stride * total * sizeof(unsigned long long))) {
        ctx->set_error("device allocation failed for the cascade workspace"); return NVCA_ERR_NOMEM;
    }
    job.cap = cap;
    job.d_hits = rb.hits.as<unsigned long long>() + hits_stride * job.r0;
    job.h_hits = rb.h_hits.as<unsigned long long>() + hits_stride * job.r0;
    if (!job.counters_zeroed) {
        NVCA_HIP_CHECK(ctx, hipMemsetAsync(job.d_hits, 0, sizeof(unsigned long long), ctx->cs()));
        NVCA_HIP_CHECK(ctx, hipMemsetAsync(ws.ln().deep.p, 0, sizeof(unsigned long long), ctx->cs()));
    }
    const bool skip_cascade = ctx->sw.skip_cascade;
    const bool host_group = ctx->sw.host_group;
    const bool dev_group = group_thr && want_group && dp.device_group_ok && !host_group && !dp.tasks.empty() && !skip_cascade;
    job.dev_group = dev_group;
    const size_t rec = 2 + 4 * kGroupOutCap;
    const size_t grp_stride = rec + 2;                                       // per result slot: a job's table is followed by the 64-bit raw count
    if (dev_group) {
        const void *old_gthr = rb.gthr.p;
        if (rb.grp.ensure((size_t)total * grp_stride * sizeof(int)) || rb.h_grp.ensure((size_t)total * grp_stride * sizeof(int)) ||
            rb.gthr.ensure((size_t)total * sizeof(int)) || rb.h_gthr.ensure((size_t)total * sizeof(int))) {
            ctx->set_error("device allocation failed for the grouping workspace"); return NVCA_ERR_NOMEM;
        }
        if (rb.gthr.p != old_gthr) rb.gthr_last.clear();          // a new buffer holds no thresholds yet
        job.d_grp = rb.grp.as<int>() + grp_stride * job.r0; job.h_grp = rb.h_grp.as<int>() + grp_stride * job.r0;
        if (rb.gthr_last.size() < (size_t)total) rb.gthr_last.resize(total, -1);
        if (memcmp(rb.gthr_last.data() + job.r0, group_thr, batch * sizeof(int)) != 0) {
            NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->cs()));             // h_gthr may still feed an earlier copy
            memcpy(rb.h_gthr.as<int>() + job.r0, group_thr, batch * sizeof(int));
            NVCA_HIP_CHECK(ctx, hipMemcpyAsync(rb.gthr.as<int>() + job.r0, rb.h_gthr.as<int>() + job.r0, batch * sizeof(int), hipMemcpyHostToDevice, ctx->cs()));
            std::copy(group_thr, group_thr + batch, rb.gthr_last.begin() + job.r0);
        }
    }
    if (!dp.tasks.empty() && !skip_cascade) {
        CascadeArgs a;
        a.sum = ws.ln().sum.as<int>(); a.sqsum = ws.ln().sqsum.as<unsigned long long>();
        a.sum_slot = sum_slot; a.spitch = spitch;
        a.scales = dp.d_scales.as<ScaleRec>();
        a.stages = dp.d_stages.as<StageRec>(); a.strips = dp.d_strips.as<StripRec>(); a.pos = dp.d_pos.as<int>();
        a.order = dp.d_order.as<int>(); a.blocks_per_frame = dp.blocks_per_frame;
        a.tasks = dp.d_tasks.as<unsigned>(); a.ntasks = (int)dp.tasks.size();
        a.failbits = ws.ln().failbits.as<unsigned long long>(); a.vnf = ws.ln().vnf.as<double>();
        a.nstages = (int)dp.stages.size(); a.pair_policy = ctx->policy == NVCA_SUM_F32PAIR; a.stage_order = ctx->sw.stage_order ? 1 : 0; a.stage_hint = dp.d_stage_hint.as<int>(); a.stage_first = dp.d_stage_first.as<int>(); a.stage_thr = dp.d_stage_thr.as<float>(); a.spec_pairs = ctx->sw.spec_pairs; a.pair_max = std::min(32, std::max(0, ctx->sw.pair_max));
        a.deep_stage = dp.deep_stage; a.deep = ws.ln().deep.as<unsigned long long>(); a.deep_cap = deep_cap;
        a.hits = job.d_hits; a.hit_cap = cap;
        a.tiles = dp.d_tiles.as<TileRec>(); a.tile_order = dp.d_tile_order.as<int>();
        a.tile_blocks_per_frame = dp.tile_blocks_per_frame;
        a.tcoords = dp.d_tcoords.as<unsigned short>(); a.tile_lds = dp.tile_lds;
        a.nscales = (int)dp.scales.size(); a.key_sy = dp.key_sy; a.key_ss = dp.key_ss;
        a.bands = dp.d_bands.as<BandRec>(); a.band_order = dp.d_band_order.as<int>(); a.band_blocks_per_frame = dp.band_blocks_per_frame; a.batch = batch;
        { const int bm = ctx->sw.band_map; a.band_map = (bm > 0 && batch % (8 * bm) == 0) ? bm : 0; }
        a.deeprecs = dp.deeprecs.empty() ? nullptr : dp.d_deeprecs.as<DeepRec>(); a.deep_lds = dp.deep_lds;
        a.tilted = dp.needs_tilted ? ws.ln().tilted.as<int>() : nullptr;
        a.galpha = dp.tabs.empty() ? nullptr : dp.tabs[0]->d_galpha; a.gcls_first = dp.tabs.empty() ? nullptr : dp.tabs[0]->d_gcls_first;
        a.stump_based = dp.generic_stumps ? 1 : 0;
        if (dp.generic) {
            // tree weak classifiers / tilted features: stage-0 pre-pass for every window, then the remaining stages on the
            // visited survivors, window per lane (kernels_cascade.hip, "general cascades")
            if (dp.needs_tilted && !a.tilted) { ctx->set_error("internal: tilted integral missing"); return NVCA_ERR_ARG; }
            { TimedLaunch t(ctx, NVCA_K_STAGE0); launch_generic(ctx->cs(), a, batch, 0); }
            { TimedLaunch t(ctx, NVCA_K_STRIP); launch_generic(ctx->cs(), a, batch, 1); }
        } else {
#ifdef NVCA_STAMPS
        {   // diagnostic build: the stamps of the last band launch are written to $NVCA_STAMPS_OUT when the context synchronises
            static DevBuf dbgbuf;
            a.dbg = nullptr;
            if (switches().stamps_out && !dbgbuf.ensure(64 * 16 * 64 * 8)) { a.dbg = dbgbuf.as<unsigned long long>(); (void)hipMemsetAsync(a.dbg, 0, 64 * 16 * 64 * 8, ctx->cs()); ctx->stamps = a.dbg; }
        }
#endif
        // one workgroup per band of window rows (k_band) when the batch offers enough bands to fill the 512 workgroup slots (>= 540 bands); otherwise stage-0 pre-pass + one workgroup per tile.  NVCA_BAND=0/1 forces the choice.
        const int band_env = ctx->sw.band;
        const bool use_band = !dp.bands.empty() && (band_env >= 0 ? band_env != 0 : (long long)dp.bands.size() * batch >= 540);     // measured crossover at 1080p (68 bands per frame): 4 frames -21 %, 8 frames +5 %, 12 frames +24 %
        auto launch = [&](int which) {
            const int e = launch_cascade_sc(ctx->cs(), a, batch, which, ctx->lds_grant);
            if (e) ctx->set_error(std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString((hipError_t)e));
            return e;
        };
        if (!use_band) { TimedLaunch t(ctx, NVCA_K_STAGE0); if (launch(0)) return NVCA_ERR_HIP; }
        if (use_band) {
            TimedLaunch t(ctx, NVCA_K_BAND); if (launch(5)) return NVCA_ERR_HIP;
        } else {
            { TimedLaunch t(ctx, NVCA_K_TILE); if (launch(3)) return NVCA_ERR_HIP; }
            { TimedLaunch t(ctx, NVCA_K_STRIP); if (launch(1)) return NVCA_ERR_HIP; }
        }
        if (early_done) NVCA_HIP_CHECK(ctx, hipEventRecord(early_done, ctx->cs()));
        { TimedLaunch t(ctx, NVCA_K_DEEP); if (launch(2)) return NVCA_ERR_HIP; }
        }
        // the box tables are small (a few KB per frame): the grouping kernel stores them straight into the page-locked host
        // buffer (plain stores, visible to the host once the stream has drained) -- no copy operation behind the last kernel
        if (dev_group) { TimedLaunch t(ctx, NVCA_K_GROUP); launch_group(ctx->cs(), a, rb.gthr.as<int>() + job.r0, grp_zero_copy ? job.h_grp : job.d_grp, kGroupOutCap, batch); }
    }
    NVCA_LAUNCH_CHECK(ctx);
    if (dev_group && grp_zero_copy) {
        // nothing to copy: k_group wrote the host buffer
    } else if (dev_group) {      // the device hands back final boxes; the raw list is only fetched for frames it declined
        NVCA_HIP_CHECK(ctx, hipMemcpyAsync(job.h_grp, job.d_grp, (rec * batch + 2) * sizeof(int), hipMemcpyDeviceToHost, ctx->cs()));
    } else {              // one D2H covers the count and (almost always) every candidate
        job.first = std::min<size_t>(cap, 2048);
        NVCA_HIP_CHECK(ctx, hipMemcpyAsync(job.h_hits, job.d_hits, (job.first + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->cs()));
    }
    return NVCA_OK;
}

// after the stream has been synchronised: raw[b] / grouped[b] for the job's n frames
static int cascade_collect(nvca_ctx *ctx, DetectPlan &dp, const CascadeJob &job, std::vector<std::vector<nvca_rect>> &raw,
                           std::vector<char> *grouped, std::vector<std::vector<int>> *scale_of = nullptr)
{
    const bool hostprof = ctx->sw.host_profile;
    const int batch = job.n;
    raw.assign(batch, {});
    if (scale_of) scale_of->assign(batch, {});
    if (grouped) grouped->assign(batch, 0);
    unsigned long long *hh = job.h_hits;
    if (job.dev_group) {
        const int *tail = job.h_grp + (size_t)(2 + 4 * kGroupOutCap) * batch;
        hh[0] = ((unsigned long long)(unsigned)tail[1] << 32) | (unsigned)tail[0];
    }
    const unsigned long long total = hh[0];
    if (hostprof) {
        unsigned long long dc = 0;
        (void)hipMemcpy(&dc, ctx->ws->ln().deep.p, sizeof(dc), hipMemcpyDeviceToHost);
        fprintf(stderr, "[nvca host] deep windows (last job) %llu, raw candidates %llu (job of %d)\n", dc, total, batch);
    }
    if (total > job.cap) {
        // the count is exact (the kernels count every candidate, they only store the first `cap`): remember the capacity per
        // frame that would have held this launch set.  The detectMultiScale entry points re-run the set once with it
        // (detect_job_advance); the batched face path starts its next batch with it.
        const unsigned long long per = (total + (unsigned long long)batch - 1) / (unsigned long long)batch + 64;
        if (per <= (unsigned long long)kMaxHitCap && (long long)per > ctx->hit_cap_wanted) ctx->hit_cap_wanted = (int)per;
        ctx->set_error("raw candidate capacity exceeded (nvca_ctx_set_hit_capacity)");
        return NVCA_ERR_OVERFLOW;
    }
    size_t have = job.first;
    if (job.dev_group) {
        const size_t rec = 2 + 4 * kGroupOutCap;
        bool need_raw = false;
        grouped->assign(batch, 1);
        for (int b = 0; b < batch; b++) {
            const int *r = job.h_grp + rec * b;
            if (r[0] < 0 || r[0] > kGroupOutCap) { (*grouped)[b] = 0; need_raw = need_raw || r[1] > 0; continue; }
            raw[b].resize(r[0]);
            for (int k = 0; k < r[0]; k++) raw[b][k] = nvca_rect{r[2 + 4 * k], r[3 + 4 * k], r[4 + 4 * k], r[5 + 4 * k]};
        }
        if (!need_raw) return NVCA_OK;
        have = 0;
    }
    if (total > have) {
        NVCA_HIP_CHECK(ctx, hipMemcpyAsync(hh + 1 + have, job.d_hits + 1 + have, (total - have) * sizeof(unsigned long long),
                                           hipMemcpyDeviceToHost, ctx->cs()));
        NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->cs()));
    }
    std::sort(hh + 1, hh + 1 + total);
    for (unsigned long long i = 0; i < total; i++) {
        // a candidate word comes from the device: it indexes host tables only after it has been checked against them (a kernel that
        // did not run, or ran on stale tables, must end as an error code, never as a wild host access)
        const unsigned long long slot_u = hh[1 + i] >> 32;
        if (slot_u >= (unsigned long long)batch || !dp.hit_valid((unsigned)hh[1 + i])) {
            ctx->set_error("internal: candidate list holds an entry outside the scan (device result rejected)");
            return NVCA_ERR_INTERNAL;
        }
        const int slot = (int)slot_u;
        if (!job.dev_group || !(*grouped)[slot]) {
            raw[slot].push_back(dp.hit_rect((unsigned)hh[1 + i]));
            if (scale_of) (*scale_of)[slot].push_back((int)((unsigned)hh[1 + i] >> dp.key_ss));
        }
    }
    return NVCA_OK;
}

static int run_cascade(nvca_ctx *ctx, DetectPlan &dp, size_t sum_slot, int spitch, int batch,
                       std::vector<std::vector<nvca_rect>> &raw, const int *group_thr = nullptr,
                       std::vector<char> *grouped = nullptr, std::vector<std::vector<int>> *scale_of = nullptr)
{
    CascadeJob job; job.n = batch; job.total = batch;
    int rc = cascade_enqueue(ctx, dp, sum_slot, spitch, job, group_thr, grouped != nullptr);
    if (rc) return rc;
    NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->cs()));
    drain_timer(ctx);
    return cascade_collect(ctx, dp, job, raw, grouped, scale_of);
}

static void group_all(std::vector<std::vector<nvca_rect>> &raw, int min_neighbors)
{
    const double GROUP_EPS = 0.2;
    for (auto &r : raw)
        if (min_neighbors != 0) group_rectangles(r, std::max(min_neighbors, 1), GROUP_EPS);
}

// ---- caller host memory -------------------------------------------------------------------------------------------------------
// A caller's host pointer reaches the HIP runtime as it stands ONLY while the memory lies inside a range the caller page-locked
// through nvca_host_register (ctx->host_ranges): a copy of pageable memory makes the runtime pin the caller's pages behind the
// library's back, and under PyTorch's bundled ROCm 7.0 runtime exactly such a copy -- a 97 x 83 numpy image, two tests after frames
// of the same heap had been page-locked and released again -- ended now and then in "Memory access fault by GPU ... on address <a
// page of the host heap>" (DESIGN 6a).  Whatever the runtime remembers about host ranges it has seen, the library does not depend
// on it: everything else is copied by the CPU into / out of page-locked slots of the context's own (ctx->bounce) and crosses
// from there.  A slot carries the event of the last copy that used it and is waited for before it is used again, so the CPU
// copy of piece k + 1 runs beside the DMA of piece k.
static int stream_id(const nvca_ctx *ctx, hipStream_t st)
{
    for (int l = 0; l < kLanes; l++) if (st == ctx->lane_streams[l]) return l;
    if (st == ctx->copy_stream) return kLanes;
    if (st == ctx->pre_streams[0]) return kLanes + 1;
    if (st == ctx->pre_streams[1]) return kLanes + 2;
    return 63;
}
static hipStream_t stream_of_id(const nvca_ctx *ctx, int id)
{
    if (id < kLanes) return ctx->lane_streams[id];
    if (id == kLanes) return ctx->copy_stream;
    if (id == kLanes + 1) return ctx->pre_streams[0];
    if (id == kLanes + 2) return ctx->pre_streams[1];
    return nullptr;
}
static int bounce_take(nvca_ctx *ctx, uint8_t **p, int *slot)
{
    BounceRing &b = ctx->bounce;
    if (!b.buf.p) {
        if (b.buf.ensure(BounceRing::kSlot * BounceRing::kSlots)) { (void)hipGetLastError(); ctx->set_error("allocation failed (page-locked staging)"); return NVCA_ERR_NOMEM; }
        for (hipEvent_t &e : b.ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { ctx->set_error("hipEventCreate failed (page-locked staging)"); return NVCA_ERR_HIP; }
    }
    const int k = b.next;
    b.next = (k + 1) % BounceRing::kSlots;
    if (b.pending[k]) { NVCA_HIP_CHECK(ctx, hipEventSynchronize(b.ev[k])); b.pending[k] = false; }
    *p = b.buf.as<uint8_t>() + (size_t)k * BounceRing::kSlot; *slot = k;
    return NVCA_OK;
}
static int bounce_used(nvca_ctx *ctx, int slot, hipStream_t st)
{
    NVCA_HIP_CHECK(ctx, hipEventRecord(ctx->bounce.ev[slot], st));
    ctx->bounce.pending[slot] = true;
    return NVCA_OK;
}
// large pieces are copied by the context's helper threads too (the PCIe link moves ~50 GB/s; one core's memcpy a fifth of that): as
// many equal parts as there are threads, none below 256 KB
// A large piece on its way INTO a page-locked slot is written once and next read by the DMA engine, never by this core: streaming
// stores (no read-for-ownership of the destination lines, no pollution of the caches with 6 MB a frame).  memcpy picks them only
// above a threshold that a thread's share of a frame does not reach.  NVCA_NT_COPY=0: plain memcpy.
static void copy_streaming(uint8_t *d, const uint8_t *s, size_t n)
{
#if defined(__SSE2__)
    static const bool on = [] { const char *e = getenv("NVCA_NT_COPY"); return !(e && e[0] == '0'); }();
    if (on && n >= (64u << 10)) {
        size_t head = (size_t)(-(intptr_t)d) & 15;
        memcpy(d, s, head); d += head; s += head; n -= head;
        const size_t blocks = n / 64;
        for (size_t i = 0; i < blocks; i++, s += 64, d += 64) {
            const __m128i a = _mm_loadu_si128((const __m128i *)s), b = _mm_loadu_si128((const __m128i *)(s + 16)),
                          c = _mm_loadu_si128((const __m128i *)(s + 32)), e = _mm_loadu_si128((const __m128i *)(s + 48));
            _mm_stream_si128((__m128i *)d, a); _mm_stream_si128((__m128i *)(d + 16), b);
            _mm_stream_si128((__m128i *)(d + 32), c); _mm_stream_si128((__m128i *)(d + 48), e);
        }
        _mm_sfence();
        n -= blocks * 64;
    }
#endif
    memcpy(d, s, n);
}
static void host_copy(nvca_ctx *ctx, void *dst, const void *src, size_t bytes, bool into_slot = false)
{
    static constexpr size_t kMinPart = 256u << 10;
    const int threads = work_pool_threads(ctx->pool) + 1;
    const int parts = (int)std::min<size_t>((size_t)threads, bytes / kMinPart);
    if (parts < 4 || !ctx->pool) { if (into_slot) copy_streaming((uint8_t *)dst, (const uint8_t *)src, bytes); else memcpy(dst, src, bytes); return; }
    struct Arg { uint8_t *d; const uint8_t *s; size_t n, part; bool nt; } arg{(uint8_t *)dst, (const uint8_t *)src, bytes, ((bytes + parts - 1) / parts + 63) & ~(size_t)63, into_slot};
    work_pool_run(ctx->pool, parts, [](void *a, int i) {
        const Arg *g = (const Arg *)a;
        const size_t o = (size_t)i * g->part;
        if (o >= g->n) return;
        if (g->nt) copy_streaming(g->d + o, g->s + o, std::min(g->part, g->n - o)); else memcpy(g->d + o, g->s + o, std::min(g->part, g->n - o));
    }, &arg);
}
static void ensure_pool(nvca_ctx *ctx)
{
    if (ctx->pool || ctx->pool_tried) return;
    ctx->pool_tried = true;
    int t = ctx->sw.host_threads;
    if (t < 0) { const int hc = (int)std::thread::hardware_concurrency(); t = std::min(8, hc / 2) - 1; }
    ctx->pool = work_pool_create(t);
}
int caller_h2d(nvca_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t st)
{
    if (!bytes) return NVCA_OK;
    if (ctx->host_ranges.note_copy(src, bytes, stream_id(ctx, st))) {
        NVCA_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        return NVCA_OK;
    }
    if (bytes >= (1u << 20)) ensure_pool(ctx);
    for (size_t o = 0; o < bytes; o += BounceRing::kSlot) {
        const size_t len = std::min(BounceRing::kSlot, bytes - o);
        uint8_t *h; int slot, rc;
        if ((rc = bounce_take(ctx, &h, &slot))) return rc;
        host_copy(ctx, h, (const uint8_t *)src + o, len, true);
        NVCA_HIP_CHECK(ctx, hipMemcpyAsync((uint8_t *)dst + o, h, len, hipMemcpyHostToDevice, st));
        if ((rc = bounce_used(ctx, slot, st))) return rc;
    }
    return NVCA_OK;
}
// rows of `width` bytes, spitch apart in the caller's memory, to rows dpitch apart on the device
int caller_h2d_rows(nvca_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows, hipStream_t st)
{
    if (!rows || !width) return NVCA_OK;
    if (width > BounceRing::kSlot) { ctx->set_error("row too long for the page-locked staging"); return NVCA_ERR_ARG; }
    if (ctx->host_ranges.note_copy(src, spitch * (rows - 1) + width, stream_id(ctx, st))) {
        NVCA_HIP_CHECK(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyHostToDevice, st));
        return NVCA_OK;
    }
    const size_t per = std::max<size_t>(1, BounceRing::kSlot / width);
    for (size_t r0 = 0; r0 < rows; r0 += per) {
        const size_t nr = std::min(per, rows - r0);
        uint8_t *h; int slot, rc;
        if ((rc = bounce_take(ctx, &h, &slot))) return rc;
        if (spitch == width) host_copy(ctx, h, (const uint8_t *)src + r0 * spitch, nr * width, true);
        else for (size_t y = 0; y < nr; y++) memcpy(h + y * width, (const uint8_t *)src + (r0 + y) * spitch, width);
        if (dpitch == width) NVCA_HIP_CHECK(ctx, hipMemcpyAsync((uint8_t *)dst + r0 * dpitch, h, nr * width, hipMemcpyHostToDevice, st));
        else NVCA_HIP_CHECK(ctx, hipMemcpy2DAsync((uint8_t *)dst + r0 * dpitch, dpitch, h, width, width, nr, hipMemcpyHostToDevice, st));
        if ((rc = bounce_used(ctx, slot, st))) return rc;
    }
    return NVCA_OK;
}
int caller_d2h_rows(nvca_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows, hipStream_t st)
{
    if (!rows || !width) { NVCA_HIP_CHECK(ctx, hipStreamSynchronize(st)); return NVCA_OK; }
    if (width > BounceRing::kSlot) { ctx->set_error("row too long for the page-locked staging"); return NVCA_ERR_ARG; }
    if (ctx->host_ranges.note_copy(dst, dpitch * (rows - 1) + width, stream_id(ctx, st))) {
        NVCA_HIP_CHECK(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyDeviceToHost, st));
        NVCA_HIP_CHECK(ctx, hipStreamSynchronize(st));
        return NVCA_OK;
    }
    const size_t per = std::max<size_t>(1, BounceRing::kSlot / width);
    for (size_t r0 = 0; r0 < rows; r0 += per) {
        const size_t nr = std::min(per, rows - r0);
        uint8_t *h; int slot, rc;
        if ((rc = bounce_take(ctx, &h, &slot))) return rc;
        if (spitch == width) NVCA_HIP_CHECK(ctx, hipMemcpyAsync(h, (const uint8_t *)src + r0 * spitch, nr * width, hipMemcpyDeviceToHost, st));
        else NVCA_HIP_CHECK(ctx, hipMemcpy2DAsync(h, width, (const uint8_t *)src + r0 * spitch, spitch, width, nr, hipMemcpyDeviceToHost, st));
        NVCA_HIP_CHECK(ctx, hipStreamSynchronize(st));
        for (size_t y = 0; y < nr; y++) memcpy((uint8_t *)dst + (r0 + y) * dpitch, h + y * width, width);
    }
    return NVCA_OK;
}
// copy a host/device 2-D byte image into device memory with a pitch
static int stage_2d(nvca_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes,
                    size_t height, int mem)
{
    if (mem == NVCA_MEM_HOST) return caller_h2d_rows(ctx, dst, dpitch, src, spitch, width_bytes, height, ctx->cs());
    NVCA_HIP_CHECK(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, height, hipMemcpyDeviceToDevice, ctx->cs()));
    return NVCA_OK;
}
static int unstage_2d(nvca_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes,
                      size_t height, int mem)
{
    NVCA_LAUNCH_CHECK(ctx);
    if (mem == NVCA_MEM_HOST) {
        const int rc = caller_d2h_rows(ctx, dst, dpitch, src, spitch, width_bytes, height, ctx->cs());
        if (!rc) drain_timer(ctx);
        return rc;
    }
    NVCA_HIP_CHECK(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, height, hipMemcpyDeviceToDevice, ctx->cs()));
    if (ctx->defer_device_sync > 0) return NVCA_OK;   // consumer is queued on the same stream
    NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->cs()));
    drain_timer(ctx);
    return NVCA_OK;
}

// end of a primitive that wrote device memory directly
static int finish_device_op(nvca_ctx *ctx)
{
    NVCA_LAUNCH_CHECK(ctx);
    if (ctx->defer_device_sync > 0) return NVCA_OK;
    NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->cs()));
    drain_timer(ctx);
    return NVCA_OK;
}

static GeomPlan *find_plan(nvca_ctx *ctx, const std::string &key)
{
    auto it = ctx->plans.find(key);
    if (it == ctx->plans.end()) return nullptr;
    it->second->last_use = ++ctx->next_uid;
    return it->second.get();
}

// Plans are cached per (cascade, geometry).  ROI-driven callers (the part detectors) ask for ever new geometries, so the
// cache is bounded: beyond kMaxPlans the least recently used plan goes (its device tables are idle: the stream is drained).
static constexpr size_t kMaxPlans = 1024;
static GeomPlan *store_plan(nvca_ctx *ctx, const std::string &key, std::unique_ptr<GeomPlan> gp)
{
    if (ctx->plans.size() >= kMaxPlans) {
        // kernels that read a victim's tables may still be queued on any lane: drain the device once, then drop the least
        // recently used quarter in one go (ROI-driven callers would otherwise pay the drain for every new geometry)
        (void)hipDeviceSynchronize();
        std::vector<std::pair<uint64_t, std::string>> order;
        for (auto &kv : ctx->plans) if (kv.second->inflight == 0) order.emplace_back(kv.second->last_use, kv.first);
        std::sort(order.begin(), order.end());
        for (size_t i = 0; i < order.size() && i < kMaxPlans / 4; i++) ctx->plans.erase(order[i].second);
    }
    gp->last_use = ++ctx->next_uid;
    GeomPlan *p = gp.get();
    ctx->plans[key] = std::move(gp);
    return p;
}

// plan for "BGR frame -> working image -> scale-cascade scan"
static int get_face_plan(nvca_ctx *ctx, const nvca_cascade *casc, int W, int H, int stride, int cn, int cols, int rows,
                         double sf, int minw, int minh, int maxw, int maxh, GeomPlan **out)
{
    // multi-scale-factor 0 (scaleFactor 1.0): OpenCV's assertion fires in detectMultiScale, the reference logs it and passes the frame on
    // untouched (FACE/kmsfacedetect.cpp:540-542 installs the property with range 0 .. 51); every other value is a ladder that ends
    if (!(sf > 1.0)) { ctx->set_error("scaleFactor must be greater than 1 (multi-scale-factor 0)"); return NVCA_ERR_ARG; }
    char key[256];
    snprintf(key, sizeof(key), "F|%llu|%d|%d|%d|%d|%d|%d|%.17g|%d|%d|%d|%d", (unsigned long long)casc->c.uid, W, H, stride,
             cn, cols, rows, sf, minw, minh, maxw, maxh);
    if (GeomPlan *gp = find_plan(ctx, key)) { *out = gp; return NVCA_OK; }
    std::unique_ptr<GeomPlan> gp(new GeomPlan());
    make_geom(gp->g, W, H, stride, cn, cols, rows);
    build_resize_tab(W, H, cols, rows, gp->tab);
    gp->rowcopy = make_rowcopy(gp->tab);
    int rc = upload_tab(ctx, *gp);
    if (rc) return rc;
    std::string err;
    rc = gp->det.build_scale_cascade(ctx, casc->c, cols, rows, gp->g.spitch, sf, minw, minh, maxw, maxh, err);
    if (rc) { ctx->set_error(err); return rc; }
    rc = gp->det.upload(ctx);
    if (rc) return rc;
    gp->has_det = true;
    *out = store_plan(ctx, key, std::move(gp));
    return NVCA_OK;
}

} // namespace nvca

// =========================================================================
// context
// =========================================================================
nvca_ctx::nvca_ctx() {}
nvca_ctx::~nvca_ctx()
{
    plans.clear();
    nvca::free_scale_tables(this);
    if (ws) ws->release_all();
    trk.release_all();
    if (part_calls_abandon) part_calls_abandon(this);          // submitted, never collected: rolled back (newest first), drained
    for (nvca::PartWorkspace &w : part_sets) w.release_all();
    if (identity_lut) (void)hipFree(identity_lut);
    bounce.release();
    nvca::work_pool_destroy(pool); pool = nullptr;
    overlay_img.release();
    for (auto &kv : roi_stage_recs) { kv.second->release(); delete kv.second; }
    for (RoiBuffers &b : roi_bufs) { b.tables.release(); b.hits.release(); b.h_tables.release(); b.h_hits.release(); }
    for (auto e : timer.pool) (void)hipEventDestroy(e);
    for (auto &e : timer.pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (nvca::FaceTicket *&t : face_tickets) { nvca::free_face_ticket(t); t = nullptr; }
    for (hipEvent_t e : chunk_events) (void)hipEventDestroy(e);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    for (int k = 0; k < 2; k++) { if (pre_streams[k]) (void)hipStreamDestroy(pre_streams[k]); if (pre_done[k]) (void)hipEventDestroy(pre_done[k]); }
    for (int l = 1; l < nvca::kLanes; l++) if (lane_streams[l]) (void)hipStreamDestroy(lane_streams[l]);
    if (stream) (void)hipStreamDestroy(stream);
}

extern "C" {

const char *nvca_version(void) { return "nubovca-hip 0.1 (gfx950)"; }

int nvca_device_count(int *n)
try {
    if (!n) return NVCA_ERR_ARG;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) { *n = 0; return NVCA_ERR_NO_DEVICE; }
    *n = c;
    return NVCA_OK;
}
NVCA_API_CATCH(nullptr)

int nvca_ctx_create(int device_id, nvca_ctx **out)
try {
    if (!out) return NVCA_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return NVCA_ERR_NO_DEVICE;   // no CPU fallback
    if (device_id < 0 || device_id >= n) return NVCA_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return NVCA_ERR_HIP;
    nvca_ctx *ctx = new (std::nothrow) nvca_ctx();
    if (!ctx) return NVCA_ERR_NOMEM;
    ctx->sw = switches();                                   // the environment is read here, once per process
    ctx->device = device_id;
    ctx->ws.reset(new Workspace());
    ctx->ws->cur_lane = &ctx->cur_lane;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return NVCA_ERR_HIP; }
    ctx->lane_streams[0] = ctx->stream;
    for (int l = 1; l < kLanes; l++)
        if (hipStreamCreateWithFlags(&ctx->lane_streams[l], hipStreamNonBlocking) != hipSuccess) { delete ctx; return NVCA_ERR_HIP; }
    *out = ctx;
    return NVCA_OK;
}
NVCA_API_CATCH(nullptr)

void nvca_ctx_destroy(nvca_ctx *ctx)
try {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    delete ctx;
}
NVCA_API_CATCH_VOID

const char *nvca_last_error(const nvca_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int nvca_ctx_set_hit_capacity(nvca_ctx *ctx, int cap)
try {
    if (!ctx || cap < 1 || cap > kMaxHitCap) return NVCA_ERR_ARG;
    ctx->hit_cap = cap;
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)
// One of the A/B switches of DESIGN.md's appendix, for this context (the environment variable of the same name, without
// the NVCA_ prefix and in lower case, sets the process default).  Switches that shape plans drop the context's cached plans.
int nvca_ctx_set_option(nvca_ctx *ctx, const char *name, int value)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (!name) return NVCA_ERR_ARG;
    const std::string n(name);
    Switches &w = ctx->sw;
    bool replan = false;
    if (n == "band") w.band = value;
    else if (n == "band_map") w.band_map = value;
    else if (n == "group_zerocopy") w.group_zero_copy = value != 0;
    else if (n == "host_group") w.host_group = value != 0;
    else if (n == "skip_cascade") w.skip_cascade = value != 0;
    else if (n == "host_profile") w.host_profile = value != 0;
    else if (n == "sparse_ingest") w.sparse_ingest = value != 0;
    else if (n == "ingest_chunk") w.ingest_chunk = value;
    else if (n == "part_stats") w.part_stats = value;
    else if (n == "trk_order") w.trk_order = value;
    else if (n == "trk_fold") w.trk_fold = value != 0;
    else if (n == "quiet") w.quiet = value != 0;
    else if (n == "roi") w.roi = value != 0;
    else if (n == "fb_dense") w.fb_dense = value != 0;
    else if (n == "two_lanes") w.two_lanes = value != 0;
    else if (n == "host_threads") { w.host_threads = value; work_pool_destroy(ctx->pool); ctx->pool = nullptr; ctx->pool_tried = false; }
    else if (n == "pre_cus") w.pre_cus = value > 0 ? value : 0;
    else if (n == "plan_debug") w.plan_debug = value != 0;
    else if (n == "stage_order") w.stage_order = value != 0;
    else if (n == "spec_pairs") w.spec_pairs = value > 0 ? value : 1;
    else if (n == "pair_max") w.pair_max = value;
    else if (n == "pyr_off") { w.pyr_off = value != 0; replan = true; }
    else if (n == "tiles") { w.tiles = value != 0; replan = true; }
    else if (n == "deep_stage") { w.deep_stage = value > 0 ? value : 0; replan = true; }
    else if (n == "deep_lds") { w.deep_lds = value != 0; replan = true; }
    else { ctx->set_error("unknown option: " + n); return NVCA_ERR_ARG; }
    if (replan) {
        for (auto &kv : ctx->plans) if (kv.second->inflight) { ctx->set_error("a batch is in flight: collect it before changing a plan option"); return NVCA_ERR_ARG; }
        (void)hipSetDevice(ctx->device);
        NVCA_HIP_CHECK(ctx, hipDeviceSynchronize());
        ctx->plans.clear();
    }
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)
// the value a switch holds for this context right now (what the environment or an earlier nvca_ctx_set_option left)
int nvca_ctx_get_option(nvca_ctx *ctx, const char *name, int *value)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (!name || !value) return NVCA_ERR_ARG;
    const std::string n(name);
    const Switches &w = ctx->sw;
    if (n == "band") *value = w.band;
    else if (n == "band_map") *value = w.band_map;
    else if (n == "group_zerocopy") *value = w.group_zero_copy;
    else if (n == "host_group") *value = w.host_group;
    else if (n == "skip_cascade") *value = w.skip_cascade;
    else if (n == "host_profile") *value = w.host_profile;
    else if (n == "sparse_ingest") *value = w.sparse_ingest;
    else if (n == "ingest_chunk") *value = w.ingest_chunk;
    else if (n == "part_stats") *value = w.part_stats;
    else if (n == "trk_order") *value = w.trk_order;
    else if (n == "trk_fold") *value = w.trk_fold;
    else if (n == "quiet") *value = w.quiet;
    else if (n == "roi") *value = w.roi;
    else if (n == "fb_dense") *value = w.fb_dense;
    else if (n == "two_lanes") *value = w.two_lanes;
    else if (n == "host_threads") *value = w.host_threads;
    else if (n == "pre_cus") *value = w.pre_cus;
    else if (n == "plan_debug") *value = w.plan_debug;
    else if (n == "stage_order") *value = w.stage_order;
    else if (n == "spec_pairs") *value = w.spec_pairs;
    else if (n == "pair_max") *value = w.pair_max;
    else if (n == "pyr_off") *value = w.pyr_off;
    else if (n == "tiles") *value = w.tiles;
    else if (n == "deep_stage") *value = w.deep_stage;
    else if (n == "deep_lds") *value = w.deep_lds;
    else { ctx->set_error("unknown option: " + n); return NVCA_ERR_ARG; }
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)
int nvca_ctx_set_sum_policy(nvca_ctx *ctx, int policy)
try {
    if (!ctx || (policy != NVCA_SUM_F32PAIR && policy != NVCA_SUM_F64)) return NVCA_ERR_ARG;
    ctx->policy = policy;
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)
int nvca_ctx_synchronize(nvca_ctx *ctx)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (!ctx) return NVCA_ERR_ARG;
    NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->cs()));
    NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->lane_streams[kFaceLane2]));      // a submitted batch may run there
    NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->lane_streams[kTrackerLane]));
#ifdef NVCA_STAMPS
    if (ctx->stamps && switches().stamps_out) {
        std::vector<unsigned long long> h(64 * 16 * 64);
        (void)hipMemcpy(h.data(), ctx->stamps, h.size() * 8, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(switches().stamps_out, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
    if (switches().stamps_out) roi_stamps_dump((std::string(switches().stamps_out) + ".roi.txt").c_str());
#endif
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)
void *nvca_ctx_stream(nvca_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int nvca_host_register(nvca_ctx *ctx, void *ptr, size_t bytes)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (!ptr || !bytes) return NVCA_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    if (ctx->host_ranges.covering(ptr, 1) >= 0 || !ctx->host_ranges.add(ptr, bytes)) { ctx->set_error("nvca_host_register: the range overlaps one that is registered"); return NVCA_ERR_ARG; }
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        bool found; (void)ctx->host_ranges.remove(ptr, &found); ctx->host_ranges.retired.pop_back();       // never was
        ctx->set_error(std::string("hipHostRegister: ") + hipGetErrorString(e));
        return NVCA_ERR_HIP;
    }
    if (alloc_log()) fprintf(stderr, "[nvca alloc] host register   %p .. %p (%zu bytes)%s\n", ptr, (void *)((char *)ptr + bytes), bytes, ctx->host_ranges.was_registered(ptr, bytes) ? " -- overlaps a range released earlier" : "");
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)
// The pages are released only when every stream that carried a copy of the range since it was registered has drained: the
// batched face path copies host frames on the copy stream and on the lanes of the two batches in flight, not only on the
// context's own stream -- an unregister behind a failed or abandoned batch must not pull pages from under a copy.
int nvca_host_unregister(nvca_ctx *ctx, void *ptr)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (!ptr) return NVCA_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    const int i = ctx->host_ranges.find(ptr);
    if (i < 0) { ctx->set_error("nvca_host_unregister: not a pointer nvca_host_register was given"); return NVCA_ERR_ARG; }
    const uint64_t streams = ctx->host_ranges.live[(size_t)i].streams;
    for (int id = 0; id < 64; id++)
        if ((streams >> id) & 1ull) {
            hipStream_t st = stream_of_id(ctx, id);
            if (st) NVCA_HIP_CHECK(ctx, hipStreamSynchronize(st)); else NVCA_HIP_CHECK(ctx, hipDeviceSynchronize());
        }
    bool found; (void)ctx->host_ranges.remove(ptr, &found);
    if (alloc_log()) fprintf(stderr, "[nvca alloc] host unregister %p (streams drained: 0x%llx)\n", ptr, (unsigned long long)streams);
    NVCA_HIP_CHECK(ctx, hipHostUnregister(ptr));
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)

int nvca_ctx_enable_kernel_timing(nvca_ctx *ctx, int on)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (!ctx) return NVCA_ERR_ARG;
    (void)hipStreamSynchronize(ctx->cs());
    drain_timer(ctx);
    ctx->timer.on = on != 0;
    ctx->timer.stride = on > 1 ? on : 1; ctx->timer.seq[0] = ctx->timer.seq[1] = 0; ctx->timer.sample = true;
    for (int k = 0; k < NVCA_K_COUNT; k++) { ctx->timer.total_ms[k] = 0; ctx->timer.launches[k] = 0; }
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)
int nvca_ctx_kernel_timing(nvca_ctx *ctx, double *total_ms, int64_t *launches)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (!ctx) return NVCA_ERR_ARG;
    NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->cs()));
    drain_timer_now(ctx);
    for (int k = 0; k < NVCA_K_COUNT; k++) {
        if (total_ms) total_ms[k] = ctx->timer.total_ms[k];
        if (launches) launches[k] = ctx->timer.launches[k];
        ctx->timer.total_ms[k] = 0; ctx->timer.launches[k] = 0;
    }
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)
const char *nvca_kernel_name(int k)
{
    static const char *names[NVCA_K_COUNT] = {"gray_resize_hist", "equalize_lut", "integral_colsum", "integral_bandscan",
                                              "integral_rows", "cascade_stage0", "cascade_strip", "cascade_deep",
                                              "group_rects", "tracker", "resize_gray", "cascade_tile", "cascade_band", "cascade_roi"};
    return (k >= 0 && k < NVCA_K_COUNT) ? names[k] : "?";
}

// =========================================================================
// cascade
// =========================================================================
int nvca_cascade_load_mem(nvca_ctx *ctx, const char *xml, int64_t len, nvca_cascade **out)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (!ctx || !xml || len <= 0 || !out) return NVCA_ERR_ARG;
    *out = nullptr;
    std::unique_ptr<nvca_cascade> c(new nvca_cascade());
    std::string err;
    int rc = parse_cascade_xml(xml, (size_t)len, c->c, err);
    if (rc) { ctx->set_error("cascade XML: " + err); return rc; }
    c->ctx = ctx;
    c->c.uid = ctx->next_uid++;
    *out = c.release();
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)

// the loader alone, on the host: no device, no context (a deployment can check its cascade files on a box without a GPU; the
// CPU test-suite and the sanitizer build run the parser through the ABI this way)
int nvca_cascade_validate_mem(const char *xml, int64_t len, int *win_w, int *win_h, int *n_stages, int *n_weak, char *err, int err_cap)
try {
    if (err && err_cap > 0) err[0] = 0;
    if (!xml || len <= 0) return NVCA_ERR_ARG;
    Cascade c;
    std::string msg;
    const int rc = parse_cascade_xml(xml, (size_t)len, c, msg);
    if (rc) { if (err && err_cap > 0) snprintf(err, (size_t)err_cap, "%s", msg.c_str()); return rc; }
    if (win_w) *win_w = c.ow;
    if (win_h) *win_h = c.oh;
    if (n_stages) *n_stages = (int)c.stages.size();
    if (n_weak) *n_weak = (int)c.cls.size();
    return NVCA_OK;
}
NVCA_API_CATCH(nullptr)

// Throws on purpose, below the barrier every entry point has: what the caller gets back is the barrier's status code.
// kind 0: std::bad_alloc, 1: std::length_error out of a container, 2: std::runtime_error, 3: a non-standard exception,
// 4: std::out_of_range out of vector::at; anything else: NVCA_OK.  (tests/test_abi_cpu.py)
int nvca_abi_selftest(int kind)
try {
    std::vector<int> v;
    if (kind == 0) throw std::bad_alloc();
    if (kind == 1) v.resize(v.max_size() + 1);
    if (kind == 2) throw std::runtime_error("selftest");
    if (kind == 3) throw 42;
    if (kind == 4) return v.at(7);
    return NVCA_OK;
}
NVCA_API_CATCH(nullptr)

int nvca_cascade_load_xml(nvca_ctx *ctx, const char *path, nvca_cascade **out)
try {
    if (!ctx || !path || !out) return NVCA_ERR_ARG;
    std::ifstream f(path, std::ios::binary);
    if (!f) { ctx->set_error(std::string("cannot open cascade file ") + path); return NVCA_ERR_IO; }
    std::stringstream ss; ss << f.rdbuf();
    const std::string s = ss.str();
    if (s.empty()) { ctx->set_error(std::string("empty cascade file ") + path); return NVCA_ERR_IO; }
    return nvca_cascade_load_mem(ctx, s.data(), (int64_t)s.size(), out);
}
NVCA_API_CATCH(ctx)

void nvca_cascade_free(nvca_cascade *c)
try {
    if (!c) return;
    // drop cached plans that reference this cascade
    if (c->ctx) {
        std::lock_guard<std::recursive_mutex> lk(c->ctx->mu);
        char pre[64];
        for (auto it = c->ctx->plans.begin(); it != c->ctx->plans.end();) {
            const std::string &k = it->first;
            snprintf(pre, sizeof(pre), "|%llu|", (unsigned long long)c->c.uid);
            if (k.find(pre) != std::string::npos && it->second->inflight == 0) { (void)hipDeviceSynchronize(); it = c->ctx->plans.erase(it); }
            else ++it;
        }
        { auto sr = c->ctx->roi_stage_recs.find((uint64_t)c->c.uid);
          if (sr != c->ctx->roi_stage_recs.end()) { (void)hipDeviceSynchronize(); sr->second->release(); delete sr->second; c->ctx->roi_stage_recs.erase(sr); } }
        bool drained = false;
        for (auto it = c->ctx->scale_tables.begin(); it != c->ctx->scale_tables.end();) {      // and its stump tables (one drain of the device for all of them)
            if (it->first.first == (uint64_t)c->c.uid && it->second->refs == 0) {
                if (!drained) { (void)hipDeviceSynchronize(); drained = true; }
                delete it->second; it = c->ctx->scale_tables.erase(it);
            } else ++it;
        }
    }
    delete c;
}
NVCA_API_CATCH_VOID

int nvca_cascade_info(const nvca_cascade *c, int *win_w, int *win_h, int *n_stages, int *n_weak)
try {
    if (!c) return NVCA_ERR_ARG;
    if (win_w) *win_w = c->c.ow;
    if (win_h) *win_h = c->c.oh;
    if (n_stages) *n_stages = (int)c->c.stages.size();
    if (n_weak) *n_weak = (int)c->c.cls.size();
    return NVCA_OK;
}
NVCA_API_CATCH((c ? c->ctx : nullptr))

int nvca_cascade_kind(const nvca_cascade *c, int *has_tilted, int *has_trees)
try {
    if (!c) return NVCA_ERR_ARG;
    if (has_tilted) *has_tilted = c->c.has_tilted ? 1 : 0;
    if (has_trees) *has_trees = c->c.stump_based ? 0 : 1;
    return NVCA_OK;
}
NVCA_API_CATCH((c ? c->ctx : nullptr))

int nvca_cascade_dump(const nvca_cascade *c, int *rects, float *weights, float *thr, float *left_val,
                      float *right_val, int *stage_sizes, float *stage_thr)
try {
    if (!c) return NVCA_ERR_ARG;
    if (!c->c.stump_based) return NVCA_ERR_UNSUPPORTED;
    for (size_t i = 0; i < c->c.cls.size(); i++) {
        const HaarNode &n = c->c.nodes[c->c.cls[i].first_node];
        if (rects) memcpy(rects + i * 12, n.rect, sizeof(int) * 12);
        if (weights) memcpy(weights + i * 3, n.weight, sizeof(float) * 3);
        if (thr) thr[i] = n.threshold;
        if (left_val) left_val[i] = c->c.alpha[c->c.cls[i].first_alpha];
        if (right_val) right_val[i] = c->c.alpha[c->c.cls[i].first_alpha + 1];
    }
    for (size_t s = 0; s < c->c.stages.size(); s++) {
        if (stage_sizes) stage_sizes[s] = c->c.stages[s].ncls;
        if (stage_thr) stage_thr[s] = c->c.stages[s].threshold;
    }
    return NVCA_OK;
}
NVCA_API_CATCH((c ? c->ctx : nullptr))

// =========================================================================
// imgproc primitives
// =========================================================================
static int check_img(nvca_ctx *ctx, const void *p, int w, int h, int stride, int bpp, int mem)
{
    if (!ctx || !p || w <= 0 || h <= 0 || stride < w * bpp || (mem != NVCA_MEM_HOST && mem != NVCA_MEM_DEVICE)) return NVCA_ERR_ARG;
    return NVCA_OK;
}

// stage `n` source frames (host or device) and return device pointers in ws.srcptrs
static size_t staging_need(const nvca_frame *frames, const int *idx, int n)
{
    size_t need = 0;
    for (int i = 0; i < n; i++) {
        const nvca_frame &f = frames[idx ? idx[i] : i];
        if (f.mem == NVCA_MEM_HOST) need += round_up((size_t)f.stride * f.height, 256);
    }
    return need;
}

// frame pointers of n frames -> device pointer array entries [r0, r0 + n); host frames are copied into the staging
// buffer first (from byte offset *off on, advanced).  `st`: the stream the copies are queued on.
static int stage_frames(nvca_ctx *ctx, const nvca_frame *frames, const int *idx, int n, int bpp, int r0 = 0,
                        hipStream_t st = nullptr, size_t *off_io = nullptr, const RowCopy *rows = nullptr)
{
    const bool sparse_off = !ctx->sw.sparse_ingest;
    if (sparse_off || (rows && !rows->on)) rows = nullptr;
    Workspace &ws = *ctx->ws;
    if (!st) st = ctx->cs();
    if (!off_io) {           // stand-alone call: size the buffers here
        if (ws.res[ws.cur_res].srcptrs.ensure((size_t)(r0 + n) * sizeof(void *)) || ws.res[ws.cur_res].h_srcptrs.ensure((size_t)(r0 + n) * sizeof(void *))) {
            ctx->set_error("allocation failed"); return NVCA_ERR_NOMEM;
        }
        const size_t need = staging_need(frames, idx, n);
        if (need && ws.res[ws.cur_res].staging.ensure(need)) { ctx->set_error("allocation failed (frame staging)"); return NVCA_ERR_NOMEM; }
    }
    const void **hp = ws.res[ws.cur_res].h_srcptrs.as<const void *>() + r0;
    size_t off = off_io ? *off_io : 0;
    for (int i = 0; i < n; i++) {
        const nvca_frame &f = frames[idx ? idx[i] : i];
        if (f.mem == NVCA_MEM_HOST) {
            uint8_t *d = ws.res[ws.cur_res].staging.as<uint8_t>() + off;
            int rc;
            if (rows) {
                // only the rows the resize reads; a run that ends on the frame's last row is copied without the row padding
                // (the caller's buffer need not extend past the last pixel)
                const size_t pitch = (size_t)rows->period * f.stride, start = (size_t)rows->first * f.stride;
                const bool tail = rows->first + (rows->count - 1) * rows->period + rows->run == f.height;
                const int full = tail ? rows->count - 1 : rows->count;
                if (full > 0 && (rc = caller_h2d_rows(ctx, d + start, pitch, (const uint8_t *)f.data + start, pitch, (size_t)rows->run * f.stride, (size_t)full, st))) return rc;
                if (tail) {
                    const size_t o = start + (size_t)full * pitch;
                    if ((rc = caller_h2d(ctx, d + o, (const uint8_t *)f.data + o, (size_t)(rows->run - 1) * f.stride + (size_t)f.width * bpp, st))) return rc;
                }
            } else if ((rc = caller_h2d(ctx, d, f.data, (size_t)f.stride * (f.height - 1) + (size_t)f.width * bpp, st))) return rc;
            hp[i] = d;
            off += round_up((size_t)f.stride * f.height, 256);
        } else
            hp[i] = f.data;
    }
    NVCA_HIP_CHECK(ctx, hipMemcpyAsync(ws.res[ws.cur_res].srcptrs.as<const void *>() + r0, hp, (size_t)n * sizeof(void *), hipMemcpyHostToDevice, st));
    if (off_io) *off_io = off;
    return NVCA_OK;
}

static bool frames_aligned4(const nvca_frame *frames, const int *idx, int n)
{
    for (int i = 0; i < n; i++) {
        const nvca_frame &f = frames[idx ? idx[i] : i];
        if ((f.stride & 3) || (f.mem == NVCA_MEM_DEVICE && ((uintptr_t)f.data & 15))) return false;
    }
    return true;
}

int nvca_bgr2gray(nvca_ctx *ctx, const void *src, int w, int h, int stride, int channels, int mem, void *dst, int dst_stride)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (channels != 3 && channels != 4) return NVCA_ERR_ARG;
    int rc = check_img(ctx, src, w, h, stride, channels, mem);
    if (rc || !dst || dst_stride < w) return NVCA_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    PreGeom g; make_geom(g, w, h, stride, channels, w, h);
    if ((rc = ensure_ws(ctx, g, 1))) return rc;
    nvca_frame f{src, w, h, stride, mem, 0};
    // The frame pointer reaches the kernel through a pointer table that is uploaded asynchronously from page-locked host
    // memory.  Callers that chain primitives without draining the stream (the part detectors queue many frames back to back)
    // must not reuse a table entry whose upload may still be pending: every call takes the next entry of a ring.
    static constexpr int kPtrRing = 1024;
    Workspace &ws = *ctx->ws;
    ResultBufs &rb = ws.res[ws.cur_res];
    if (rb.srcptrs.ensure(kPtrRing * sizeof(void *)) || rb.h_srcptrs.ensure(kPtrRing * sizeof(void *))) { ctx->set_error("allocation failed"); return NVCA_ERR_NOMEM; }
    if (ctx->defer_device_sync > 0 && ++ctx->ptr_ring_used >= kPtrRing) {       // a full turn without a drain: drain once
        NVCA_HIP_CHECK(ctx, hipDeviceSynchronize());
        ctx->ptr_ring_used = 0;
    }
    const int slot = ctx->defer_device_sync > 0 ? ctx->ptr_ring_used : 0;
    if ((rc = stage_frames(ctx, &f, nullptr, 1, channels, slot))) return rc;
    { TimedLaunch t(ctx, NVCA_K_GRAY);
      launch_gray(ctx->cs(), rb.srcptrs.as<const uint8_t *>() + slot, g, 0, nullptr, nullptr, nullptr, nullptr, w,
                  ctx->ws->ln().gray.as<uint8_t>(), nullptr, 1, frames_aligned4(&f, nullptr, 1)); }
    return unstage_2d(ctx, dst, dst_stride, ctx->ws->ln().gray.p, g.gpitch, w, h, mem);
}
NVCA_API_CATCH(ctx)

// resize coefficient tables for (source size -> destination size), cached with the other plans
static int get_resize_plan(nvca_ctx *ctx, int sw, int sh, int dw, int dh, GeomPlan **out)
{
    char key[96];
    snprintf(key, sizeof(key), "RZ|%d|%d|%d|%d", sw, sh, dw, dh);
    if (GeomPlan *gp = find_plan(ctx, key)) { *out = gp; return NVCA_OK; }
    std::unique_ptr<GeomPlan> gp(new GeomPlan());
    build_resize_tab(sw, sh, dw, dh, gp->tab);
    int rc = upload_tab(ctx, *gp);
    if (rc) return rc;
    *out = store_plan(ctx, key, std::move(gp));
    return NVCA_OK;
}

int nvca_resize_linear(nvca_ctx *ctx, const void *src, int sw, int sh, int sstride, int channels, int mem, void *dst,
                       int dw, int dh, int dstride)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (channels == 3) {
        int rc3 = check_img(ctx, src, sw, sh, sstride, 3, mem);
        if (rc3 || !dst || dw <= 0 || dh <= 0 || dstride < dw * 3) return NVCA_ERR_ARG;
        (void)hipSetDevice(ctx->device);
        Workspace &w3 = *ctx->ws;
        const size_t sp = round_up((size_t)sw * 3, 64), dp3 = round_up((size_t)dw * 3, 64);
        if (w3.ln().staging.ensure(sp * sh + 64) || w3.ln().aux.ensure(dp3 * dh + 64)) { ctx->set_error("allocation failed"); return NVCA_ERR_NOMEM; }
        if ((rc3 = stage_2d(ctx, w3.ln().staging.p, sp, src, sstride, (size_t)sw * 3, sh, mem))) return rc3;
        GeomPlan *gp3 = nullptr;
        if ((rc3 = get_resize_plan(ctx, sw, sh, dw, dh, &gp3))) return rc3;
        { TimedLaunch t(ctx, NVCA_K_RESIZE1);
          launch_resize3(ctx->cs(), w3.ln().staging.as<uint8_t>(), sw, sh, (int)sp, gp3->tab.mode, gp3->d_xofs.as<int>(), gp3->d_ialpha.as<short>(),
                         gp3->d_yofs.as<int>(), gp3->d_ibeta.as<short>(), gp3->tab.xmax, w3.ln().aux.as<uint8_t>(), dw, dh, (int)dp3); }
        return unstage_2d(ctx, dst, dstride, w3.ln().aux.p, dp3, (size_t)dw * 3, dh, mem);
    }
    if (channels != 1) return NVCA_ERR_ARG;
    int rc = check_img(ctx, src, sw, sh, sstride, 1, mem);
    if (rc || !dst || dw <= 0 || dh <= 0 || dstride < dw) return NVCA_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    Workspace &ws = *ctx->ws;
    PreGeom gs; make_geom(gs, sw, sh, sstride, 1, sw, sh);
    PreGeom gd; make_geom(gd, sw, sh, sstride, 1, dw, dh);
    GeomPlan *gp = nullptr;
    if ((rc = get_resize_plan(ctx, sw, sh, dw, dh, &gp))) return rc;
    if (mem == NVCA_MEM_DEVICE) {          // device images are read and written in place (ordered on the context's stream)
        { TimedLaunch t(ctx, NVCA_K_RESIZE1);
          launch_resize1(ctx->cs(), (const uint8_t *)src, sw, sh, sstride, gp->tab.mode, gp->d_xofs.as<int>(),
                         gp->d_ialpha.as<short>(), gp->d_yofs.as<int>(), gp->d_ibeta.as<short>(), gp->tab.xmax,
                         (uint8_t *)dst, dw, dh, dstride, nullptr); }
        return finish_device_op(ctx);
    }
    if ((rc = ensure_ws(ctx, gs, 1)) || (rc = ensure_ws(ctx, gd, 1))) return rc;
    if (ws.ln().aux.ensure(gd.gray_slot + 64)) { ctx->set_error("allocation failed"); return NVCA_ERR_NOMEM; }
    if ((rc = stage_2d(ctx, ws.ln().gray.p, gs.gpitch, src, sstride, sw, sh, mem))) return rc;
    { TimedLaunch t(ctx, NVCA_K_RESIZE1);
      launch_resize1(ctx->cs(), ws.ln().gray.as<uint8_t>(), sw, sh, gs.gpitch, gp->tab.mode, gp->d_xofs.as<int>(),
                     gp->d_ialpha.as<short>(), gp->d_yofs.as<int>(), gp->d_ibeta.as<short>(), gp->tab.xmax,
                     ws.ln().aux.as<uint8_t>(), dw, dh, gd.gpitch, nullptr); }
    return unstage_2d(ctx, dst, dstride, ws.ln().aux.p, gd.gpitch, dw, dh, mem);
}
NVCA_API_CATCH(ctx)

int nvca_equalize_hist(nvca_ctx *ctx, const void *src, int w, int h, int stride, int mem, void *dst, int dst_stride)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    int rc = check_img(ctx, src, w, h, stride, 1, mem);
    if (rc || !dst || dst_stride < w) return NVCA_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    Workspace &ws = *ctx->ws;
    PreGeom g; make_geom(g, w, h, stride, 1, w, h);
    if ((rc = ensure_ws(ctx, g, 1))) return rc;
    if (mem == NVCA_MEM_DEVICE) {          // histogram of the caller's image, LUT applied straight into the destination (in place allowed)
        NVCA_HIP_CHECK(ctx, hipMemsetAsync(ws.ln().hist.p, 0, 256 * sizeof(unsigned), ctx->cs()));
        { TimedLaunch t(ctx, NVCA_K_GRAY); launch_hist(ctx->cs(), (const uint8_t *)src, w, h, stride, ws.ln().hist.as<unsigned>()); }
        { TimedLaunch t(ctx, NVCA_K_LUT); launch_lut(ctx->cs(), ws.ln().hist.as<unsigned>(), w * h, ws.ln().lut.as<uint8_t>(), 1, 1); }
        launch_apply_lut(ctx->cs(), (const uint8_t *)src, w, h, stride, ws.ln().lut.as<uint8_t>(), (uint8_t *)dst, dst_stride);
        return finish_device_op(ctx);
    }
    if (ws.ln().aux.ensure(g.gray_slot + 64)) { ctx->set_error("allocation failed"); return NVCA_ERR_NOMEM; }
    if ((rc = stage_2d(ctx, ws.ln().gray.p, g.gpitch, src, stride, w, h, mem))) return rc;
    NVCA_HIP_CHECK(ctx, hipMemsetAsync(ws.ln().hist.p, 0, 256 * sizeof(unsigned), ctx->cs()));
    { TimedLaunch t(ctx, NVCA_K_GRAY); launch_hist(ctx->cs(), ws.ln().gray.as<uint8_t>(), w, h, g.gpitch, ws.ln().hist.as<unsigned>()); }
    { TimedLaunch t(ctx, NVCA_K_LUT); launch_lut(ctx->cs(), ws.ln().hist.as<unsigned>(), w * h, ws.ln().lut.as<uint8_t>(), 1, 1); }   // slot 0 left zeroed again
    launch_apply_lut(ctx->cs(), ws.ln().gray.as<uint8_t>(), w, h, g.gpitch, ws.ln().lut.as<uint8_t>(), ws.ln().aux.as<uint8_t>(), g.gpitch);
    return unstage_2d(ctx, dst, dst_stride, ws.ln().aux.p, g.gpitch, w, h, mem);
}
NVCA_API_CATCH(ctx)

int nvca_draw_shapes(nvca_ctx *ctx, const nvca_frame *frame, int channels, const nvca_shape *shapes, int n)
try {
    // host frames need no device (and no context): plain loops over the mapped buffer
    const bool host = frame && frame->mem == NVCA_MEM_HOST;
    if (!frame || (!ctx && !host) || (channels != 3 && channels != 4) || n < 0 || (n > 0 && !shapes) || n > 1024) return NVCA_ERR_ARG;
    if (!frame->data || frame->width <= 0 || frame->height <= 0 || frame->stride < frame->width * channels || (frame->mem != NVCA_MEM_HOST && frame->mem != NVCA_MEM_DEVICE)) return NVCA_ERR_ARG;
    for (int i = 0; i < n; i++)
        if ((shapes[i].kind != NVCA_SHAPE_RECT3 && shapes[i].kind != NVCA_SHAPE_RING4) || std::abs((long long)shapes[i].x) > (1 << 24) || std::abs((long long)shapes[i].y) > (1 << 24) ||
            std::abs((long long)shapes[i].w) > (1 << 24) || std::abs((long long)shapes[i].h) > (1 << 24)) return NVCA_ERR_ARG;
    if (!n) return NVCA_OK;
    if (host) { draw_shapes_host((uint8_t *)frame->data, frame->width, frame->height, frame->stride, channels, shapes, n); return NVCA_OK; }
    NVCA_LOCK_OR_FAIL(ctx);
    int rc;
    (void)hipSetDevice(ctx->device);
    int bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN;          // common bounding box, clipped to the frame
    for (int i = 0; i < n; i++) {
        const nvca_shape &sh = shapes[i];
        int x0, y0, x1, y1;
        if (sh.kind == NVCA_SHAPE_RING4) { const int r = (sh.w > 0 ? sh.w : 0) + 2; x0 = sh.x - r; x1 = sh.x + r; y0 = sh.y - r; y1 = sh.y + r; }
        else { x0 = std::min(sh.x, sh.x + sh.w) - 1; x1 = std::max(sh.x, sh.x + sh.w) + 1; y0 = std::min(sh.y, sh.y + sh.h) - 1; y1 = std::max(sh.y, sh.y + sh.h) + 1; }
        bx0 = std::min(bx0, x0); by0 = std::min(by0, y0); bx1 = std::max(bx1, x1); by1 = std::max(by1, y1);
    }
    bx0 = std::max(bx0, 0); by0 = std::max(by0, 0); bx1 = std::min(bx1, frame->width - 1); by1 = std::min(by1, frame->height - 1);
    if (bx0 > bx1 || by0 > by1) return NVCA_OK;
    void *d_shapes = nullptr;
    if ((rc = part_table(ctx, shapes, (size_t)n * sizeof(nvca_shape), &d_shapes))) return rc;
    launch_draw_shapes(ctx->cs(), (uint8_t *)frame->data, frame->width, frame->height, frame->stride, channels, (const nvca_shape *)d_shapes, n, bx0, by0, bx1, by1);
    return finish_device_op(ctx);
}
NVCA_API_CATCH(ctx)

int nvca_overlay_blend(nvca_ctx *ctx, const nvca_frame *frame, const nvca_rect *boxes, int n, const nvca_overlay *ov)
try {
    const bool host = frame && frame->mem == NVCA_MEM_HOST;
    if (!frame || !ov || (!ctx && !host) || n < 0 || (n > 0 && !boxes) || n > 1024) return NVCA_ERR_ARG;
    if (!frame->data || frame->width <= 0 || frame->height <= 0 || frame->stride < frame->width * 3 || (frame->mem != NVCA_MEM_HOST && frame->mem != NVCA_MEM_DEVICE)) return NVCA_ERR_ARG;
    if (!ov->data || ov->width <= 0 || ov->height <= 0 || (ov->channels != 1 && ov->channels != 3 && ov->channels != 4) || ov->stride < ov->width * ov->channels ||
        ov->width > 8192 || ov->height > 8192) return NVCA_ERR_ARG;
    if (!(std::fabs(ov->offset_x_percent) <= 64 && std::fabs(ov->offset_y_percent) <= 64 && ov->width_percent >= 0 && ov->width_percent <= 64 && ov->height_percent >= 0 && ov->height_percent <= 64)) return NVCA_ERR_ARG;
    for (int i = 0; i < n; i++)
        if (std::abs((long long)boxes[i].x) > (1 << 20) || std::abs((long long)boxes[i].y) > (1 << 20) || boxes[i].w < 0 || boxes[i].h < 0 || boxes[i].w > (1 << 14) || boxes[i].h > (1 << 14)) return NVCA_ERR_ARG;
    if (!n || ov->height_percent == 0 || ov->width_percent == 0) return NVCA_OK;           // FACE/kmsfacedetect.cpp:436-439
    if (host) { overlay_blend_host((uint8_t *)frame->data, frame->width, frame->height, frame->stride, boxes, n, *ov); return NVCA_OK; }
    NVCA_LOCK_OR_FAIL(ctx);
    (void)hipSetDevice(ctx->device);
    int rc;
    const size_t bytes = (size_t)ov->stride * (ov->height - 1) + (size_t)ov->width * ov->channels;
    if (ctx->overlay_img.ensure(bytes + 64)) { ctx->set_error("allocation failed (overlay image)"); return NVCA_ERR_NOMEM; }
    if ((rc = caller_h2d(ctx, ctx->overlay_img.p, ov->data, bytes, ctx->cs()))) return rc;
    for (int b = 0; b < n; b++) {            // in order: a later box overwrites an earlier one where they overlap
        const OverlayPlace p = overlay_place(boxes[b], *ov);
        if (p.w <= 0 || p.h <= 0) continue;
        GeomPlan *gp = nullptr;
        if ((rc = get_resize_plan(ctx, ov->width, ov->height, p.w, p.h, &gp))) return rc;
        launch_overlay(ctx->cs(), (uint8_t *)frame->data, frame->width, frame->height, frame->stride, p, ctx->overlay_img.as<uint8_t>(), ov->height, ov->stride, ov->channels,
                       gp->tab.mode, gp->d_xofs.as<int>(), gp->d_ialpha.as<short>(), gp->d_yofs.as<int>(), gp->d_ibeta.as<short>(), gp->tab.xmax);
    }
    // the image is the caller's: the upload must have left it before the call returns
    NVCA_LAUNCH_CHECK(ctx);
    NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->cs()));
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)

int nvca_flip_horizontal(nvca_ctx *ctx, const void *src, int w, int h, int stride, int mem, void *dst, int dst_stride)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    int rc = check_img(ctx, src, w, h, stride, 1, mem);
    if (rc || !dst || dst_stride < w) return NVCA_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    Workspace &ws = *ctx->ws;
    PreGeom g; make_geom(g, w, h, stride, 1, w, h);
    if (mem == NVCA_MEM_DEVICE && src != dst) {
        launch_flip_h(ctx->cs(), (const uint8_t *)src, w, h, stride, (uint8_t *)dst, dst_stride);
        return finish_device_op(ctx);
    }
    if ((rc = ensure_ws(ctx, g, 1))) return rc;
    if (ws.ln().aux.ensure(g.gray_slot + 64)) { ctx->set_error("allocation failed"); return NVCA_ERR_NOMEM; }
    if ((rc = stage_2d(ctx, ws.ln().gray.p, g.gpitch, src, stride, w, h, mem))) return rc;
    launch_flip_h(ctx->cs(), ws.ln().gray.as<uint8_t>(), w, h, g.gpitch, ws.ln().aux.as<uint8_t>(), g.gpitch);
    return unstage_2d(ctx, dst, dst_stride, ws.ln().aux.p, g.gpitch, w, h, mem);
}
NVCA_API_CATCH(ctx)

int nvca_integral(nvca_ctx *ctx, const void *src, int w, int h, int stride, int mem, int32_t *sum, double *sqsum)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    int rc = check_img(ctx, src, w, h, stride, 1, mem);
    if (rc || !sum) return NVCA_ERR_ARG;
    if (mem != NVCA_MEM_HOST) { ctx->set_error("nvca_integral: host output only"); return NVCA_ERR_ARG; }
    (void)hipSetDevice(ctx->device);
    Workspace &ws = *ctx->ws;
    PreGeom g; make_geom(g, w, h, stride, 1, w, h);
    if ((rc = ensure_ws(ctx, g, 1))) return rc;
    if ((rc = stage_2d(ctx, ws.ln().gray.p, g.gpitch, src, stride, w, h, mem))) return rc;
    run_integral(ctx, g, nullptr, 1);
    rc = unstage_2d(ctx, sum, (size_t)(w + 1) * 4, ws.ln().sum.p, (size_t)g.spitch * 4, (size_t)(w + 1) * 4, h + 1, NVCA_MEM_HOST);
    if (rc) return rc;
    if (sqsum) {                                       // device layout: u32 low-word plane, then u8 high-byte plane
        const size_t n = (size_t)(w + 1) * (h + 1);
        std::vector<unsigned> lo(n); std::vector<uint8_t> hi(n);
        rc = unstage_2d(ctx, lo.data(), (size_t)(w + 1) * 4, ws.ln().sqsum.p, (size_t)g.spitch * 4, (size_t)(w + 1) * 4, h + 1, NVCA_MEM_HOST);
        if (rc) return rc;
        rc = unstage_2d(ctx, hi.data(), (size_t)(w + 1), ws.ln().sqsum.as<unsigned>() + g.sum_slot, (size_t)g.spitch, (size_t)(w + 1), h + 1, NVCA_MEM_HOST);
        if (rc) return rc;
        for (size_t i = 0; i < n; i++) sqsum[i] = (double)(((unsigned long long)hi[i] << 32) | lo[i]);
    }
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)

int nvca_integral_tilted(nvca_ctx *ctx, const void *src, int w, int h, int stride, int mem, int32_t *tilted)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    int rc = check_img(ctx, src, w, h, stride, 1, mem);
    if (rc || !tilted) return NVCA_ERR_ARG;
    if (mem != NVCA_MEM_HOST) { ctx->set_error("nvca_integral_tilted: host output only"); return NVCA_ERR_ARG; }
    (void)hipSetDevice(ctx->device);
    Workspace &ws = *ctx->ws;
    PreGeom g; make_geom(g, w, h, stride, 1, w, h);
    if ((rc = ensure_ws(ctx, g, 1))) return rc;
    if ((rc = stage_2d(ctx, ws.ln().gray.p, g.gpitch, src, stride, w, h, mem))) return rc;
    if ((rc = run_tilted(ctx, g, nullptr, 1))) return rc;
    return unstage_2d(ctx, tilted, (size_t)(w + 1) * 4, ws.ln().tilted.p, (size_t)g.spitch * 4, (size_t)(w + 1) * 4, h + 1, NVCA_MEM_HOST);
}
NVCA_API_CATCH(ctx)

// =========================================================================
// detectMultiScale
// =========================================================================
} // extern "C"

// A detectMultiScale call in halves: enqueue() queues the next launch set of the call on the context's stream and returns;
// advance(), after the stream has drained, consumes what the set produced and either finishes the call or asks for another
// set (FIND_BIGGEST narrows its scan once).  Many calls can therefore share ONE wait per round: the part detectors queue the
// face passes of every stream of a tick, then every ROI pass, with three synchronisations per tick instead of several per
// stream (parts.cpp).  The image planes are shared working memory: jobs use them one after the other in stream order; what a
// job leaves behind for the host (its candidate list) lives in its own result region (CascadeJob::r0).
namespace nvca {

struct FbStep { double factor, ystep; int winw, winh; };

struct DetectJob {
    // ---- request
    int kind = 0;                                    // 0: scale-cascade scan, 1: CV_HAAR_SCALE_IMAGE, 2: CV_HAAR_FIND_BIGGEST_OBJECT
    const nvca_cascade *casc = nullptr;
    const void *img[kJobImages] = {nullptr}; int nimg = 1;       // plain scan / SCALE_IMAGE: images of one geometry share the launches
    int cols = 0, rows = 0, stride = 0, mem = 0;
    double sf = 1.1; int min_neighbors = 0, flags = 0, minw = 0, minh = 0, maxw = 0, maxh = 0;
    bool raw_only = false;
    // ---- result
    std::vector<nvca_rect> out[kJobImages];
    // ---- progress
    int phase = 0;                                   // 0: new, 1: first launch set queued, 2: narrowed set queued, 3: done
    int slots() const { return nimg; }
    GeomPlan *gp = nullptr;                          // cached plan of the queued set (kept from eviction while queued)
    std::unique_ptr<DetectPlan> own;                 // FIND_BIGGEST: this call's narrowed plan
    DetectPlan *dp = nullptr;                        // plan of the queued set (null: nothing was queued)
    CascadeJob cj; int gthr = 0;
    // FIND_BIGGEST: the serial loop's state between the two sets
    std::vector<FbStep> ladder; std::vector<std::vector<nvca_rect>> hits; std::vector<char> have; std::vector<int> ladder_of;
    std::vector<nvca_rect> all; nvca_rect scanROI{0, 0, 0, 0}; bool narrowed_done = false; size_t fb_i = 0; int cur_minw = 0, cur_minh = 0;
    int regrown = 0;                                 // launch sets re-run with a larger candidate list (at most one per set)
    // small-image path (kernels_roi.hip): the job's steps of the queued launch and the candidates that came back
    struct RoiStepInfo { double ystep, out_factor; int winw, winh, ladder; };
    bool small = false;                              // the job runs on the small-image path (decided at its first round)
    int roi_prev_phase = 0;                          // its phase before the queued set (a set that overflowed the list is queued again)
    bool fused = false;                              // the queued set went into the round's k_roi launch
    std::vector<RoiStepInfo> rinfo;
    std::vector<unsigned> rkeys[kJobImages];         // per image: step << 26 | iy << 13 | ix, ascending (= OpenCV's serial order)
    // FIND_BIGGEST on the small-image path, dense first launch (Switches::fb_dense): per step of the queued launch where its stage-0 reject bits
    // lie in the launch's bitmap (word offset, words per grid row, grid size); per LADDER step what came back -- every window that passes the whole
    // cascade, visited by the serial walk or not (iy << 13 | ix, ascending), and the reject bits of the step's full grid
    struct RejInfo { int off, wpr, nx, ny; };
    bool dense = false;                              // the queued launch was a dense one
    std::vector<RejInfo> rej_info;                   // [step of the launch]
    std::vector<std::vector<unsigned>> dense_hits;   // [ladder step]
    std::vector<const unsigned long long *> rej_bits; std::vector<int> rej_wpr, rej_rows;      // [ladder step]: into the launch's page-locked bitmap (valid until the next launch of its buffer set: the job is advanced before)
};
// was window ix of a grid row visited by the serial walk that started at column `start`?  (visited iff the run of stage-0 rejects
// immediately left of it, not reaching below `start`, has even length: the walk steps by 2 behind a stage-0 reject, by 1 otherwise)
static inline bool fb_visited(const unsigned long long *row, int start, int ix)
{
    int run = 0;
    for (int x = ix - 1; x >= start && ((row[x >> 6] >> (x & 63)) & 1ull); x--) run++;
    return !(run & 1);
}

static bool fb_make_spec(const DetectJob &j, int spitch, const FbStep &st, int startX, int endX, int startY, int endY, ScaleSpec &sp)
{   // scan grid of one ladder step; false: nothing to scan there
    if (!(endX > startX && endY > startY)) return false;
    sp = ScaleSpec();
    sp.table_factor = st.factor; sp.plane_off = 0; sp.pitch = spitch; sp.plane_rows = j.rows + 1; sp.adaptive = 1;
    sp.out_factor = 0; sp.out_w = st.winw; sp.out_h = st.winh;
    for (int ix = startX; ix < endX; ix++) sp.xs.push_back(cv_round(ix * st.ystep));
    for (int iy = startY; iy < endY; iy++) sp.ys.push_back(cv_round(iy * st.ystep));
    // cvRunHaarClassifierCascadeSum returns -1 (no hit, step 1) outside the image: drop such grid points
    while (!sp.xs.empty() && (sp.xs.back() + st.winw >= j.cols + 1)) sp.xs.pop_back();
    while (!sp.ys.empty() && (sp.ys.back() + st.winh >= j.rows + 1)) sp.ys.pop_back();
    const bool neg = (!sp.xs.empty() && sp.xs.front() < 0) || (!sp.ys.empty() && sp.ys.front() < 0);
    return !sp.xs.empty() && !sp.ys.empty() && !neg;
}

// cvHaarDetectObjectsForROC, CV_HAAR_SCALE_IMAGE branch (EYE/kmseyedetect.cpp:991-993, NOSE/kmsnosedetect.cpp:843-846,
// MOUTH/kmsmouthdetect.cpp:845-848, EAR/kmseardetect.cpp:656-659): per factor the image is resized, integrated and
// scanned with the unscaled window on a fixed grid.  All pyramid levels (of both images: the ear detector scans an image
// and its mirror, EAR/kmseardetect.cpp:796-803) are evaluated by one launch set.
static int si_plan(nvca_ctx *ctx, const DetectJob &j, GeomPlan **out)
{
    const Cascade &c = j.casc->c;
    const int cols = j.cols, rows = j.rows;
    int rc;
    // pyramid layout, resize tables and scan tables depend only on (cascade, image size, parameters): built once
    char key[256];
    snprintf(key, sizeof(key), "SI|%llu|%d|%d|%.17g|%d|%d|%d|%d", (unsigned long long)c.uid, cols, rows, j.sf, j.minw, j.minh, j.maxw, j.maxh);
    GeomPlan *pp = find_plan(ctx, key);
    if (!pp) {
        std::unique_ptr<GeomPlan> np(new GeomPlan());
        np->P = (int)round_up(cols + 1, 8);
        for (double factor = 1;; factor *= j.sf) {
            const int winw = cv_round(c.ow * factor), winh = cv_round(c.oh * factor);
            const int szw = cv_round(cols / factor), szh = cv_round(rows / factor);
            if (szw - c.ow + 1 <= 0 || szh - c.oh + 1 <= 0) break;
            if (winw > j.maxw || winh > j.maxh) break;
            if (winw < j.minw || winh < j.minh) continue;
            if (szw + 1 <= 1 + c.ow) continue;                   // HaarDetectObjects_ScaleImage_Invoker's early return
            PyrLevel L; L.f = factor; L.szw = szw; L.szh = szh; L.winw = winw; L.winh = winh;
            L.gpitch = (int)round_up(szw, 64); L.gray_off = np->gray_total; L.plane_off = (int)np->plane_total;
            np->gray_total += round_up((size_t)L.gpitch * szh, 256);
            np->plane_total += round_up((size_t)np->P * (szh + 1), 64);
            np->lv.push_back(L);
            if (np->lv.size() > 62) break;
        }
        std::vector<ScaleSpec> specs;
        for (const PyrLevel &L : np->lv) {
            std::unique_ptr<GeomPlan> gp(new GeomPlan());
            build_resize_tab(cols, rows, L.szw, L.szh, gp->tab);
            np->level_tabs.push_back(std::move(gp));
            ScaleSpec sp;
            sp.table_factor = 1.; sp.plane_off = L.plane_off; sp.pitch = np->P; sp.plane_rows = L.szh + 1; sp.adaptive = 0;
            sp.out_factor = L.f; sp.out_w = L.winw; sp.out_h = L.winh;
            const int ystep = L.f > 2 ? 1 : 2;
            for (int x = 0; x < L.szw - c.ow; x += ystep) sp.xs.push_back(x);
            for (int y = 0; y < L.szh - c.oh; y += ystep) sp.ys.push_back(y);
            specs.push_back(std::move(sp));
        }
        if ((rc = upload_tabs(ctx, np->level_tabs, np->d_level_tabs))) return rc;
        if (!np->lv.empty()) {
            std::string err;
            if ((rc = np->det.build_custom(ctx, c, std::move(specs), false, err))) { ctx->set_error(err); return rc; }
            if ((rc = np->det.upload(ctx))) return rc;
            std::vector<PyrLevelDev> dl(np->lv.size());
            np->pyr_ok = !ctx->sw.pyr_off;
            for (size_t li = 0; li < np->lv.size(); li++) {
                const PyrLevel &L = np->lv[li]; GeomPlan *t = np->level_tabs[li].get();
                PyrLevelDev &d = dl[li]; memset(&d, 0, sizeof(d));
                d.szw = L.szw; d.szh = L.szh; d.gpitch = L.gpitch; d.mode = t->tab.mode; d.xmax = t->tab.xmax; d.plane_off = L.plane_off;
                d.gray_off = (long long)L.gray_off;
                d.xofs = t->d_xofs.as<int>(); d.ialpha = t->d_ialpha.as<short>(); d.yofs = t->d_yofs.as<int>(); d.ibeta = t->d_ibeta.as<short>();
                np->pyr_maxw = std::max(np->pyr_maxw, L.szw); np->pyr_maxh = std::max(np->pyr_maxh, L.szh);
                if (L.szw > 1023) np->pyr_ok = false;            // one column per thread, plus the zero column
            }
            if (np->d_pyr.ensure(dl.size() * sizeof(PyrLevelDev))) { ctx->set_error("allocation failed (pyramid table)"); return NVCA_ERR_NOMEM; }
            NVCA_HIP_CHECK(ctx, hipMemcpy(np->d_pyr.p, dl.data(), dl.size() * sizeof(PyrLevelDev), hipMemcpyHostToDevice));
        }
        pp = store_plan(ctx, key, std::move(np));
    }
    *out = pp;
    return NVCA_OK;
}

// Device images of a job that sit at equal distances (the working images of a batched part call are carved that way) are read
// where they are; anything else is copied into the lane's gray slots first.
static bool job_images_in_place(const DetectJob &j, size_t *slot)
{
    if (j.mem != NVCA_MEM_DEVICE) return false;
    *slot = 0;
    if (j.nimg == 1) return true;
    const uint8_t *a = (const uint8_t *)j.img[0], *b = (const uint8_t *)j.img[1];
    if (b <= a) return false;
    const size_t d = (size_t)(b - a);
    if (d < (size_t)j.stride * (j.rows - 1) + j.cols) return false;
    for (int k = 2; k < j.nimg; k++) if ((const uint8_t *)j.img[k] != a + d * k) return false;
    *slot = d;
    return true;
}

static int si_enqueue(nvca_ctx *ctx, DetectJob &j, int r0, int total)
{
    Workspace &ws = *ctx->ws;
    const Cascade &c = j.casc->c;
    const int cols = j.cols, rows = j.rows, nimg = j.nimg;
    GeomPlan *pp = nullptr;
    int rc;
    if ((rc = si_plan(ctx, j, &pp))) return rc;
    j.phase = 1; j.dp = nullptr;
    if (pp->lv.empty()) return NVCA_OK;
    const int P = pp->P;
    const size_t gray_total = pp->gray_total, plane_total = pp->plane_total;
    PreGeom g0; make_geom(g0, cols, rows, j.stride, 1, cols, rows);
    if ((rc = ensure_ws(ctx, g0, nimg))) return rc;
    if (ws.ln().aux.ensure(gray_total * nimg + 64) || ws.ln().sum.ensure((plane_total * nimg + 4 * (size_t)P) * sizeof(int)) || ws.ln().sqsum.ensure(plane_total * nimg * sizeof(unsigned long long)) ||
        (c.has_tilted && ws.ln().tilted.ensure((plane_total * nimg + 4 * (size_t)P) * sizeof(int)))) {
        ctx->set_error("allocation failed (pyramid)"); return NVCA_ERR_NOMEM;
    }
    if (c.has_tilted && (size_t)2 * (pp->pyr_maxw + pp->pyr_maxh + 2) * sizeof(int) > 64 * 1024) { ctx->set_error("image too large for the tilted integral"); return NVCA_ERR_ARG; }
    const uint8_t *src0 = ws.ln().gray.as<uint8_t>(); int spitch0 = g0.gpitch; size_t sslot0 = g0.gray_slot;
    size_t in_place_slot = 0;
    if (job_images_in_place(j, &in_place_slot)) { src0 = (const uint8_t *)j.img[0]; spitch0 = j.stride; sslot0 = in_place_slot; }
    else
        for (int k = 0; k < nimg; k++)
            if ((rc = stage_2d(ctx, ws.ln().gray.as<uint8_t>() + g0.gray_slot * k, g0.gpitch, j.img[k], j.stride, cols, rows, j.mem))) return rc;
    if (pp->pyr_ok) {            // all levels of all images: one resize launch, one integral launch
        { TimedLaunch t(ctx, NVCA_K_RESIZE1);
          launch_pyr_resize(ctx->cs(), src0, cols, rows, spitch0, sslot0, pp->d_pyr.as<PyrLevelDev>(),
                            (int)pp->lv.size(), nimg, pp->pyr_maxw, pp->pyr_maxh, ws.ln().aux.as<uint8_t>(), gray_total); }
        { TimedLaunch t(ctx, NVCA_K_INTEGRAL);
          launch_pyr_integral(ctx->cs(), ws.ln().aux.as<uint8_t>(), gray_total, pp->d_pyr.as<PyrLevelDev>(), (int)pp->lv.size(), nimg,
                              ws.ln().sum.as<int>(), ws.ln().sqsum.as<unsigned>(), plane_total, P); }
        if (c.has_tilted) {          // cvIntegral(&img1, &sum1, &sqsum1, _tilted) per level
            TimedLaunch t(ctx, NVCA_K_INTEGRAL);
            launch_pyr_tilted(ctx->cs(), ws.ln().aux.as<uint8_t>(), gray_total, pp->d_pyr.as<PyrLevelDev>(), (int)pp->lv.size(), nimg,
                              ws.ln().tilted.as<int>(), plane_total, P, pp->pyr_maxw, pp->pyr_maxh);
        }
    } else
    for (size_t li = 0; li < pp->lv.size(); li++) {
        const PyrLevel &L = pp->lv[li];
        GeomPlan *gp = pp->level_tabs[li].get();
        uint8_t *lg = ws.ln().aux.as<uint8_t>() + L.gray_off;
        { TimedLaunch t(ctx, NVCA_K_RESIZE1);               // cvResize(img, &img1, CV_INTER_LINEAR)
          launch_resize1(ctx->cs(), src0, cols, rows, spitch0, gp->tab.mode, gp->d_xofs.as<int>(),
                         gp->d_ialpha.as<short>(), gp->d_yofs.as<int>(), gp->d_ibeta.as<short>(), gp->tab.xmax, lg, L.szw,
                         L.szh, L.gpitch, nullptr, nimg, sslot0, gray_total); }
        PreGeom g; make_geom(g, L.szw, L.szh, L.gpitch, 1, L.szw, L.szh);
        g.gpitch = L.gpitch; g.spitch = P; g.sum_slot = plane_total; g.gray_slot = gray_total;
        run_integral(ctx, g, nullptr, nimg, lg, ws.ln().sum.as<int>() + L.plane_off,
                     (unsigned long long *)(ws.ln().sqsum.as<unsigned>() + L.plane_off));     // lo plane of the level; hi plane at + plane_total
        if (c.has_tilted && (rc = run_tilted(ctx, g, nullptr, nimg, lg, ws.ln().tilted.as<int>() + L.plane_off))) return rc;
    }
    j.cj = CascadeJob(); j.cj.r0 = r0; j.cj.n = nimg; j.cj.total = total;
    if ((rc = cascade_enqueue(ctx, pp->det, plane_total, P, j.cj, nullptr, false))) return rc;
    j.gp = pp; pp->inflight++; j.dp = &pp->det;
    return NVCA_OK;
}

// plain scale-cascade scan (flags without SCALE_IMAGE / FIND_BIGGEST): FACE/kmsfacedetect.cpp:809-811, EYE/kmseyedetect.cpp:958-960
static int plain_enqueue(nvca_ctx *ctx, DetectJob &j, int r0, int total)
{
    GeomPlan *gp = nullptr;
    int rc;
    if ((rc = get_face_plan(ctx, j.casc, j.cols, j.rows, j.stride, 1, j.cols, j.rows, j.sf, j.minw, j.minh, j.maxw, j.maxh, &gp))) return rc;
    const int nimg = j.nimg;
    if ((rc = ensure_ws(ctx, gp->g, nimg))) return rc;
    PreGeom g = gp->g;
    const uint8_t *src = nullptr;
    size_t in_place_slot = 0;
    if (job_images_in_place(j, &in_place_slot) && j.stride % 4 == 0 && ((uintptr_t)j.img[0] & 3) == 0 && in_place_slot % 4 == 0) {
        src = (const uint8_t *)j.img[0]; g.gpitch = j.stride; g.gray_slot = in_place_slot;      // the integral kernels read rows in 4-byte words
    } else
        for (int k = 0; k < nimg; k++)
            if ((rc = stage_2d(ctx, ctx->ws->ln().gray.as<uint8_t>() + gp->g.gray_slot * k, gp->g.gpitch, j.img[k], j.stride, j.cols, j.rows, j.mem))) return rc;
    run_integral(ctx, g, nullptr, nimg, src);
    if (j.casc->c.has_tilted && (rc = run_tilted(ctx, g, nullptr, nimg, src))) return rc;
    j.gthr = (!j.raw_only && j.min_neighbors != 0) ? std::max(j.min_neighbors, 1) : 0;
    j.cj = CascadeJob(); j.cj.r0 = r0; j.cj.n = nimg; j.cj.total = total;
    const std::vector<int> gthrv(nimg, j.gthr);
    if ((rc = cascade_enqueue(ctx, gp->det, gp->g.sum_slot, gp->g.spitch, j.cj, j.gthr ? gthrv.data() : nullptr, true))) return rc;
    j.gp = gp; gp->inflight++; j.dp = &gp->det; j.phase = 1;
    return NVCA_OK;
}

// cvHaarDetectObjectsForROC with CV_HAAR_FIND_BIGGEST_OBJECT (NOSE/kmsnosedetect.cpp:870-873, MOUTH/kmsmouthdetect.cpp:870-873,
// EAR/kmseardetect.cpp:712-715): scale-cascade scan from the largest factor down; after the first grouped detection
// the scan narrows to a region of interest and a minimum size.  The serial loop changes its scan only once, so two launch
// sets do: (1) every step on its full grid (a cached plan per geometry), (2) once the region is known, the remaining steps
// on their narrowed grids.  fb_replay() replays the serial logic on those results, step by step.
static int fb_stage_image(nvca_ctx *ctx, const DetectJob &j, PreGeom &g)
{
    make_geom(g, j.cols, j.rows, j.stride, 1, j.cols, j.rows);
    int rc;
    if ((rc = ensure_ws(ctx, g, 1))) return rc;
    if ((rc = stage_2d(ctx, ctx->ws->ln().gray.p, g.gpitch, j.img[0], j.stride, j.cols, j.rows, j.mem))) return rc;
    run_integral(ctx, g, nullptr, 1);
    if (j.casc->c.has_tilted && (rc = run_tilted(ctx, g, nullptr, 1))) return rc;
    return NVCA_OK;
}

static int fb_enqueue_first(nvca_ctx *ctx, DetectJob &j, int r0, int total)
{
    const Cascade &c = j.casc->c;
    const int cols = j.cols, rows = j.rows;
    PreGeom g; int rc;
    if ((rc = fb_stage_image(ctx, j, g))) return rc;
    // the ladder of factors, largest first, exactly as the serial loop walks it
    j.ladder.clear();
    {
        int n_factors = 0; double factor;
        for (n_factors = 0, factor = 1; factor * c.ow < cols - 10 && factor * c.oh < rows - 10; n_factors++, factor *= j.sf)
            ;
        const double inv = 1. / j.sf; factor *= inv;
        for (; n_factors-- > 0; factor *= inv) j.ladder.push_back(FbStep{factor, std::max(2., factor), cv_round(c.ow * factor), cv_round(c.oh * factor)});
    }
    j.hits.assign(j.ladder.size(), {}); j.have.assign(j.ladder.size(), 0);
    j.all.clear(); j.scanROI = nvca_rect{0, 0, 0, 0}; j.narrowed_done = false; j.fb_i = 0; j.cur_minw = j.minw; j.cur_minh = j.minh;
    char key[256];
    snprintf(key, sizeof(key), "FB|%llu|%d|%d|%.17g|%d|%d|%d|%d", (unsigned long long)c.uid, cols, rows, j.sf, j.minw, j.minh, j.maxw, j.maxh);
    GeomPlan *p1 = find_plan(ctx, key);
    if (!p1) {
        std::unique_ptr<GeomPlan> np(new GeomPlan());
        std::vector<ScaleSpec> specs;
        for (size_t i = 0; i < j.ladder.size(); i++) {
            const FbStep &st = j.ladder[i];
            if (st.winw < j.minw || st.winh < j.minh) break;
            if (st.winw > j.maxw || st.winh > j.maxh) continue;
            ScaleSpec sp;
            if (fb_make_spec(j, g.spitch, st, 0, cv_round((cols - st.winw) / st.ystep), 0, cv_round((rows - st.winh) / st.ystep), sp)) {
                specs.push_back(std::move(sp)); np->fb_ladder.push_back((int)i);
            }
        }
        if (!specs.empty()) {
            std::string err;
            if ((rc = np->det.build_custom(ctx, c, std::move(specs), false, err))) { ctx->set_error(err); return rc; }
            if ((rc = np->det.upload(ctx))) return rc;
        }
        p1 = store_plan(ctx, key, std::move(np));
    }
    j.phase = 1; j.dp = nullptr;
    // steps the full-grid plan does not hold have nothing to scan
    for (size_t i = 0; i < j.ladder.size(); i++) j.have[i] = 1;
    if (!p1->fb_ladder.empty()) {
        j.ladder_of = p1->fb_ladder;
        j.cj = CascadeJob(); j.cj.r0 = r0; j.cj.n = 1; j.cj.total = total;
        if ((rc = cascade_enqueue(ctx, p1->det, g.sum_slot, g.spitch, j.cj, nullptr, false))) return rc;
        j.gp = p1; p1->inflight++; j.dp = &p1->det;
    }
    return NVCA_OK;
}

// the narrowed launch set: this step and all later ones on their narrowed grids (nothing changes the scan any more)
static int fb_enqueue_narrowed(nvca_ctx *ctx, DetectJob &j, int r0, int total)
{
    PreGeom g; int rc;
    if ((rc = fb_stage_image(ctx, j, g))) return rc;         // the planes have served other jobs in between
    j.cj = CascadeJob(); j.cj.r0 = r0; j.cj.n = 1; j.cj.total = total;
    if ((rc = cascade_enqueue(ctx, *j.own, g.sum_slot, g.spitch, j.cj, nullptr, false))) return rc;
    j.dp = j.own.get();
    return NVCA_OK;
}

static bool roi_grid(int cols, int rows, double ystep, int winw, int winh, int startX, int endX, int startY, int endY, RoiStep &st);
// the serial loop of cvHaarDetectObjectsForROC on the scan results at hand; returns 1 when it needs the narrowed set first
static int fb_replay(nvca_ctx *ctx, DetectJob &j)
{
    const Cascade &c = j.casc->c;
    const bool rough = (j.flags & NVCA_HAAR_DO_ROUGH_SEARCH) != 0;
    const int cols = j.cols, rows = j.rows;
    const int spitch = (int)round_up(cols + 1, 8);
    for (size_t i = j.fb_i; i < j.ladder.size(); i++) {
        const FbStep &st = j.ladder[i];
        if (st.winw < j.cur_minw || st.winh < j.cur_minh) break;
        if (st.winw > j.maxw || st.winh > j.maxh) continue;
        const bool narrowed = j.scanROI.w * j.scanROI.h > 0;
        if (narrowed && !j.narrowed_done) {
            j.narrowed_done = true;
            std::vector<ScaleSpec> specs; j.ladder_of.clear();
            for (size_t k = i; k < j.ladder.size(); k++) {
                const FbStep &sk = j.ladder[k];
                j.hits[k].clear(); j.have[k] = 1;
                if (sk.winw < j.cur_minw || sk.winh < j.cur_minh) break;
                if (sk.winw > j.maxw || sk.winh > j.maxh) continue;
                ScaleSpec sp;
                const int sx0 = cv_round(j.scanROI.x / sk.ystep), sx1 = cv_round((j.scanROI.x + j.scanROI.w - sk.winw) / sk.ystep);
                const int sy0 = cv_round(j.scanROI.y / sk.ystep), sy1 = cv_round((j.scanROI.y + j.scanROI.h - sk.winh) / sk.ystep);
                if (j.small && j.dense && k < j.dense_hits.size() && j.rej_wpr[k] > 0) {
                    // dense first launch: the narrowed walk of this step is replayed here -- its windows are grid points of the full grid, the
                    // launch reported every one of them that passes the cascade, and which of them the walk from column sx0 visits follows
                    // from the stage-0 reject bits (no second launch, no second wait).  A step the first launch did not hold (below the call's
                    // minSize: the narrowed search lowers it to 0.4 / 0.6 of the object found) still takes the second launch, below.
                    RoiStep tmp;
                    if (roi_grid(cols, rows, sk.ystep, sk.winw, sk.winh, sx0, sx1, sy0, sy1, tmp)) {
                        const int wpr = j.rej_wpr[k];
                        for (unsigned key : j.dense_hits[k]) {            // ascending (iy, ix): the serial order
                            const int iy = (int)(key >> 13), ix = (int)(key & 8191);
                            if (iy < tmp.startY || iy >= tmp.endY || ix < tmp.startX || ix >= tmp.endX) continue;
                            if (iy >= j.rej_rows[k] || !fb_visited(j.rej_bits[k] + (size_t)iy * wpr, tmp.startX, ix)) continue;
                            j.hits[k].push_back(nvca_rect{cv_round(ix * sk.ystep), cv_round(iy * sk.ystep), sk.winw, sk.winh});
                        }
                    }
                } else if (j.small) {             // small-image path: no plan, the narrowed grids go into the next round's launch as they are
                    RoiStep tmp;
                    if (roi_grid(cols, rows, sk.ystep, sk.winw, sk.winh, sx0, sx1, sy0, sy1, tmp)) { j.ladder_of.push_back((int)k); j.have[k] = 0; }
                } else if (fb_make_spec(j, spitch, sk, sx0, sx1, sy0, sy1, sp)) {
                    specs.push_back(std::move(sp)); j.ladder_of.push_back((int)k); j.have[k] = 0;
                }
            }
            if (j.small && !j.ladder_of.empty()) { j.fb_i = i; return 1; }
            if (!specs.empty()) {
                j.own.reset(new DetectPlan()); std::string err;
                int rc;
                if ((rc = j.own->build_custom(ctx, c, std::move(specs), false, err))) { ctx->set_error(err); return rc < 0 ? rc : NVCA_ERR_ARG; }
                if ((rc = j.own->upload(ctx))) return rc;
                j.fb_i = i;
                return 1;                                    // come back with the narrowed scans
            }
        }
        j.all.insert(j.all.end(), j.hits[i].begin(), j.hits[i].end());
        if (!j.all.empty() && j.scanROI.w * j.scanROI.h == 0) {
            std::vector<nvca_rect> tmp(j.all);
            group_rectangles(tmp, std::max(j.min_neighbors, 1), 0.2);
            if (!tmp.empty()) {
                nvca_rect maxRect{0, 0, 0, 0};
                for (const nvca_rect &r : tmp) if (r.w * r.h > maxRect.w * maxRect.h) maxRect = r;
                j.all.push_back(maxRect);
                j.scanROI = maxRect;
                const int dx = cv_round(maxRect.w * 0.2), dy = cv_round(maxRect.h * 0.2);
                j.scanROI.x = std::max(j.scanROI.x - dx, 0); j.scanROI.y = std::max(j.scanROI.y - dy, 0);
                j.scanROI.w = std::min(j.scanROI.w + dx * 2, cols - 1 - j.scanROI.x);
                j.scanROI.h = std::min(j.scanROI.h + dy * 2, rows - 1 - j.scanROI.y);
                const double minScale = rough ? 0.6 : 0.4;
                j.cur_minw = cv_round(maxRect.w * minScale); j.cur_minh = cv_round(maxRect.h * minScale);
            }
        }
    }
    group_rectangles(j.all, std::max(j.min_neighbors, 1), 0.2);
    j.out[0].clear();
    if (!j.all.empty()) {
        nvca_rect best{0, 0, 0, 0};
        for (const nvca_rect &r : j.all) if (r.w * r.h > best.w * best.h) best = r;
        j.out[0].push_back(best);
    }
    return 0;
}

// ---- small images: one launch for every such job of a round (kernels_roi.hip) -----------------------------------------------
// A job qualifies when its image's integral pair fits the workgroup's LDS, the cascade is a stump cascade with upright
// features, and the ladder fits the candidate key.  No plan is built: the launch gets, per job, a handful of step records
// (the cached stump table of the step's factor, the variance rectangle, the grid limits) -- so a face region of a size never
// seen before costs no table work, and all regions of all streams of a round share ONE launch.
static constexpr int kRoiMaxWords = 14848;          // (cols + 1) * (rows + 2) words per plane: the part detectors' 160 x 90 face-pass image still fits (two planes + queues + a level image = 157 KB of the 160 KB of LDS)
struct RoiBatch {
    std::vector<RoiJobDev> jobs; std::vector<RoiStep> steps; std::vector<unsigned char> tabs; std::vector<DetectJob *> owners; std::vector<int> owner_img;
    std::vector<ScaleTable *> held;                 // stump tables of the launch: kept from eviction until it has been collected
    int plane_words = 0, lev_bytes = 0, lane = 0; unsigned cap = 0; size_t first = 0;
    size_t rej_words = 0;                           // stage-0 reject bitmaps of the launch's dense steps (u64 words)
    void release() { for (ScaleTable *t : held) if (t->refs > 0) t->refs--; held.clear(); }
    ~RoiBatch() { release(); }
    RoiBatch() = default;
    RoiBatch(const RoiBatch &) = delete; RoiBatch &operator=(const RoiBatch &) = delete;
    // (a round object is reused from round to round: what the previous round held is released first)
    void reset() { release(); jobs.clear(); steps.clear(); tabs.clear(); owners.clear(); owner_img.clear(); plane_words = 0; lev_bytes = 0; lane = 0; cap = 0; first = 0; rej_words = 0; }
};
static bool roi_eligible(const nvca_ctx *ctx, const DetectJob &j, int njobs_in_round)
{
    const Cascade &c = j.casc->c;
    if (!ctx->sw.roi) return false;
    if (!c.stump_based || c.has_tilted || (j.nimg != 1 && j.mem != NVCA_MEM_DEVICE)) return false;
    if ((long long)(j.cols + 1) * (j.rows + 2) > kRoiMaxWords || j.cols < 1 || j.rows < 1) return false;
    if (j.mem != NVCA_MEM_DEVICE && njobs_in_round != 1) return false;       // a host image is staged in the lane's one gray buffer
    return true;
}
static const StageRec *roi_stage_recs(nvca_ctx *ctx, const Cascade &c)
{
    auto it = ctx->roi_stage_recs.find(c.uid);
    if (it != ctx->roi_stage_recs.end()) return it->second->as<StageRec>();
    std::vector<StageRec> st; build_stage_recs(c, st);
    std::unique_ptr<DevBuf> d(new DevBuf());
    if (d->ensure(st.size() * sizeof(StageRec) + 8) || hipMemcpy(d->p, st.data(), st.size() * sizeof(StageRec), hipMemcpyHostToDevice) != hipSuccess) {
        d->release(); ctx->set_error("allocation failed (stage records)"); return nullptr;
    }
    const StageRec *p = d->as<StageRec>();
    ctx->roi_stage_recs[c.uid] = d.release();
    return p;
}
static ScaleTable *roi_table(nvca_ctx *ctx, RoiBatch &rb, const Cascade &c, double factor)
{
    ScaleTable *t = get_scale_table(ctx, c, factor);
    if (t) { t->refs++; rb.held.push_back(t); }
    return t;
}
static void roi_step_common(RoiStep &st, const ScaleTable &t)
{
    memset(&st, 0, sizeof(st));
    st.trecs = t.dev.as<TStumpRec>(); st.ex = t.ex; st.ey = t.ey; st.ew = t.ew; st.eh = t.eh; st.inv_area = t.inv_area; st.step = 1;
}
// scale-cascade grid of one ladder step, limits as indices: false = nothing to scan (fb_make_spec's rules: grid points whose
// window would leave the image -- cvRunHaarClassifierCascadeSum returns -1 there -- are dropped from the end, a negative origin voids the step)
static bool roi_grid(int cols, int rows, double ystep, int winw, int winh, int startX, int endX, int startY, int endY, RoiStep &st)
{
    if (!(endX > startX && endY > startY)) return false;
    while (endX > startX && cv_round((endX - 1) * ystep) + winw >= cols + 1) endX--;
    while (endY > startY && cv_round((endY - 1) * ystep) + winh >= rows + 1) endY--;
    if (!(endX > startX && endY > startY)) return false;
    if (cv_round(startX * ystep) < 0 || cv_round(startY * ystep) < 0 || endX > 8191 || endY > 8191) return false;
    st.startX = startX; st.endX = endX; st.startY = startY; st.endY = endY; st.ystep = ystep; st.adaptive = 1;
    return true;
}
// returns NVCA_OK with j.fused set when the job's next set went into the batch, NVCA_OK with j.fused clear when it has to take
// the large-image path after all (too many steps), or an error
static int roi_add_job(nvca_ctx *ctx, RoiBatch &rb, DetectJob &j)
{
    const Cascade &c = j.casc->c;
    const int cols = j.cols, rows = j.rows;
    j.fused = false;
    const StageRec *d_stages = roi_stage_recs(ctx, c);
    if (!d_stages) return NVCA_ERR_NOMEM;
    std::vector<RoiStep> steps; std::vector<DetectJob::RoiStepInfo> info; std::vector<unsigned char> tabs;
    const size_t tab0 = rb.tabs.size();
    int lev_bytes = 0;
    const bool dense = j.kind == 2 && j.phase == 0 && ctx->sw.fb_dense && j.nimg == 1;
    std::vector<DetectJob::RejInfo> rej; size_t rej_local = 0;
    if (j.phase == 0) for (int k = 0; k < kJobImages; k++) j.out[k].clear();
    if (j.kind == 1) {
        // the pyramid levels of si_plan, each with its cv::resize tables
        ScaleTable *t1 = nullptr;
        for (double factor = 1;; factor *= j.sf) {
            const int winw = cv_round(c.ow * factor), winh = cv_round(c.oh * factor);
            const int szw = cv_round(cols / factor), szh = cv_round(rows / factor);
            if (szw - c.ow + 1 <= 0 || szh - c.oh + 1 <= 0) break;
            if (winw > j.maxw || winh > j.maxh) break;
            if (winw < j.minw || winh < j.minh) continue;
            if (szw + 1 <= 1 + c.ow) continue;
            if (!t1 && !(t1 = roi_table(ctx, rb, c, 1.))) return NVCA_ERR_NOMEM;
            RoiStep st; roi_step_common(st, *t1);
            st.szw = szw; st.szh = szh; st.step = factor > 2 ? 1 : 2; st.startX = 0; st.endX = szw - c.ow; st.startY = 0; st.endY = szh - c.oh;
            if (st.endX <= 0 || st.endY <= 0) continue;
            ResizeTab tab; build_resize_tab(cols, rows, szw, szh, tab);
            st.mode = tab.mode; st.xmax = tab.xmax;
            auto put = [&](const void *p, size_t n) { const size_t at = (tab0 + tabs.size() + 15) & ~(size_t)15; tabs.resize(at - tab0 + n); if (n) memcpy(tabs.data() + at - tab0, p, n); return (int)at; };
            st.xofs_off = put(tab.xofs.data(), tab.xofs.size() * 4); st.yofs_off = put(tab.yofs.data(), tab.yofs.size() * 4);
            st.ialpha_off = put(tab.ialpha.data(), tab.ialpha.size() * 2); st.ibeta_off = put(tab.ibeta.data(), tab.ibeta.size() * 2);
            steps.push_back(st); info.push_back(DetectJob::RoiStepInfo{0., factor, winw, winh, -1});
            lev_bytes = std::max(lev_bytes, szw * szh);
        }
        j.phase = 1;
    } else if (j.kind == 0) {
        std::vector<double> factors;
        scale_grid(c.ow, c.oh, cols, rows, j.sf, j.minw, j.minh, j.maxw, j.maxh, false, factors);
        for (double factor : factors) {
            const double ystep = std::max(2., factor);
            ScaleTable *t = roi_table(ctx, rb, c, factor);
            if (!t) return NVCA_ERR_NOMEM;
            RoiStep st; roi_step_common(st, *t);
            if (!roi_grid(cols, rows, ystep, t->winw, t->winh, 0, cv_round((cols - t->winw) / ystep), 0, cv_round((rows - t->winh) / ystep), st)) continue;
            steps.push_back(st); info.push_back(DetectJob::RoiStepInfo{ystep, 0., t->winw, t->winh, -1});
        }
        j.gthr = (!j.raw_only && j.min_neighbors != 0) ? std::max(j.min_neighbors, 1) : 0;
        j.phase = 1;
    } else {
        if (j.phase == 0) {
            // the ladder of factors, largest first, exactly as the serial loop walks it (fb_enqueue_first)
            j.ladder.clear();
            int n_factors = 0; double factor;
            for (n_factors = 0, factor = 1; factor * c.ow < cols - 10 && factor * c.oh < rows - 10; n_factors++, factor *= j.sf)
                ;
            const double inv = 1. / j.sf; factor *= inv;
            for (; n_factors-- > 0; factor *= inv) j.ladder.push_back(FbStep{factor, std::max(2., factor), cv_round(c.ow * factor), cv_round(c.oh * factor)});
            j.hits.assign(j.ladder.size(), {}); j.have.assign(j.ladder.size(), 1);
            j.all.clear(); j.scanROI = nvca_rect{0, 0, 0, 0}; j.narrowed_done = false; j.fb_i = 0; j.cur_minw = j.minw; j.cur_minh = j.minh;
            j.ladder_of.clear();
            for (size_t i = 0; i < j.ladder.size(); i++) {
                const FbStep &fs = j.ladder[i];
                if (fs.winw < j.minw || fs.winh < j.minh) break;
                if (fs.winw > j.maxw || fs.winh > j.maxh) continue;
                ScaleTable *t = roi_table(ctx, rb, c, fs.factor);
                if (!t) return NVCA_ERR_NOMEM;
                RoiStep st; roi_step_common(st, *t);
                if (!roi_grid(cols, rows, fs.ystep, fs.winw, fs.winh, 0, cv_round((cols - fs.winw) / fs.ystep), 0, cv_round((rows - fs.winh) / fs.ystep), st)) continue;
                if (dense) {                     // every stage-0 passer of the full grid + the grid's reject bits: a narrowed re-scan is replayed on the host
                    st.adaptive = 2; st.rej_wpr = (st.endX + 63) / 64; st.rej_off = (int)(rb.rej_words + rej_local);
                    rej.push_back(DetectJob::RejInfo{st.rej_off, st.rej_wpr, st.endX, st.endY});
                    rej_local += (size_t)st.rej_wpr * st.endY;
                }
                steps.push_back(st); info.push_back(DetectJob::RoiStepInfo{fs.ystep, 0., fs.winw, fs.winh, (int)i}); j.ladder_of.push_back((int)i);
            }
            j.phase = 1;
        } else {
            // the narrowed set fb_replay asked for: steps fb_i .. on their narrowed grids (j.ladder_of / j.have were set by the replay)
            for (int li : j.ladder_of) {
                const FbStep &fs = j.ladder[li];
                ScaleTable *t = roi_table(ctx, rb, c, fs.factor);
                if (!t) return NVCA_ERR_NOMEM;
                RoiStep st; roi_step_common(st, *t);
                if (!roi_grid(cols, rows, fs.ystep, fs.winw, fs.winh, cv_round(j.scanROI.x / fs.ystep), cv_round((j.scanROI.x + j.scanROI.w - fs.winw) / fs.ystep),
                              cv_round(j.scanROI.y / fs.ystep), cv_round((j.scanROI.y + j.scanROI.h - fs.winh) / fs.ystep), st)) continue;
                steps.push_back(st); info.push_back(DetectJob::RoiStepInfo{fs.ystep, 0., fs.winw, fs.winh, li});
            }
        }
    }
    bool fits = steps.size() <= 63;                                  // the key holds 6 bits of step
    {
        // A step's rows are independent of one another (the adaptive x step works row by row, a pyramid level's grid is fixed):
        // a step with more windows than the queues hold goes out as several records, a band of whole rows each -- one workgroup
        // per band instead of one per step walking its bands one after the other (the largest pyramid level of a 160 x 90 face
        // pass is 10 k windows: alone it set the length of the whole launch).  Every band builds the integral pair for itself.
        std::vector<RoiStep> bands;
        for (size_t li = 0; li < steps.size(); li++) {
            RoiStep st = steps[li];
            const int nx = (st.endX - st.startX + st.step - 1) / st.step, ny = (st.endY - st.startY + st.step - 1) / st.step;
            if (nx > kRoiMaxWin) fits = false;                       // (a grid row longer than the queues: not with images this small)
            st.key_step = (int)li;
            st.key_x0 = st.startX; st.key_dx = st.step; st.key_dy = st.step;
            const int rows_per = nx > 0 && nx < kRoiMaxWin ? kRoiMaxWin / nx : 1;
            for (int gy0 = 0; gy0 < std::max(ny, 1); gy0 += rows_per) {
                RoiStep b = st;
                b.startY = st.startY + gy0 * st.step;
                b.endY = std::min(st.endY, st.startY + (gy0 + rows_per) * st.step);
                b.key_y0 = b.startY;
                bands.push_back(b);
            }
        }
        steps.swap(bands);
    }
    if (!fits && j.roi_prev_phase == 2) { ctx->set_error("internal: a narrowed search outgrew the small-image path"); return NVCA_ERR_INTERNAL; }   // (its full grids fitted)
    if (!fits) { j.phase = j.roi_prev_phase; return NVCA_OK; }       // this one takes the large-image path
    j.fused = true; j.rinfo.swap(info); j.dp = nullptr;
    j.dense = dense && !rej.empty(); j.rej_info.swap(rej);
    if (j.dense) rb.rej_words += rej_local;
    for (int k = 0; k < kJobImages; k++) j.rkeys[k].clear();
    if (steps.empty()) return NVCA_OK;                               // nothing to scan: the job completes with what it has
    rb.tabs.insert(rb.tabs.end(), tabs.begin(), tabs.end());
    for (int k = 0; k < j.nimg; k++) {            // every image of the job: its own records (the steps name their image), the same tables
        RoiJobDev d; memset(&d, 0, sizeof(d));
        d.w = cols; d.h = rows; d.stride = j.stride; d.img = (const uint8_t *)j.img[k];
        if (j.mem != NVCA_MEM_DEVICE) {
            PreGeom g; make_geom(g, cols, rows, j.stride, 1, cols, rows);
            int rc;
            if ((rc = ensure_ws(ctx, g, 1))) return rc;
            if ((rc = stage_2d(ctx, ctx->ws->ln().gray.p, g.gpitch, j.img[0], j.stride, cols, rows, j.mem))) return rc;
            d.img = ctx->ws->ln().gray.as<uint8_t>(); d.stride = g.gpitch;
        }
        d.first_step = (int)rb.steps.size(); d.nsteps = (int)steps.size(); d.scale_image = j.kind == 1;
        d.stages = d_stages; d.nstages = (int)c.stages.size(); d.pair_policy = ctx->policy == NVCA_SUM_F32PAIR; d.slot = (int)rb.jobs.size();
        for (RoiStep &st : steps) st.job = d.slot;
        rb.steps.insert(rb.steps.end(), steps.begin(), steps.end());
        rb.jobs.push_back(d); rb.owners.push_back(&j); rb.owner_img.push_back(k);
    }
    rb.plane_words = std::max(rb.plane_words, (cols + 1) * (rows + 2));
    rb.lev_bytes = std::max(rb.lev_bytes, lev_bytes);
    return NVCA_OK;
}
// upload the round's tables and launch k_roi on the current lane
static int roi_launch(nvca_ctx *ctx, RoiBatch &rb, bool full_cap)
{
    const int nj = (int)rb.jobs.size();
    const size_t jb = (size_t)nj * sizeof(RoiJobDev), sb = rb.steps.size() * sizeof(RoiStep);
    const size_t o_steps = (jb + 255) & ~(size_t)255, o_tabs = (o_steps + sb + 255) & ~(size_t)255, total = o_tabs + rb.tabs.size() + 64;
    // the list starts at a quarter of a million candidates for the whole launch however many jobs share it; a launch that
    // overflows it is queued again with the exact size (run_detect_jobs raises hit_cap for the rest of the call)
    const long long want = (long long)ctx->hit_cap * nj;
    rb.cap = (unsigned)std::min<long long>(want, full_cap ? (1ll << 26) : (1ll << 18));
    const size_t first = std::min<size_t>(rb.cap, std::max<size_t>(8192, ctx->roi_first_hint));
    rb.first = first;
    if (ctx->rbuf().tables.ensure(total) || ctx->rbuf().h_tables.ensure(total) || ctx->rbuf().hits.ensure(((size_t)rb.cap + 1) * 8) || ctx->rbuf().h_hits.ensure(((size_t)rb.cap + 1) * 8) ||
        ctx->rbuf().rej.ensure((rb.rej_words + 1) * 8) || ctx->rbuf().h_rej.ensure((rb.rej_words + 1) * 8)) {
        ctx->set_error("allocation failed (small-image detector)"); return NVCA_ERR_NOMEM;
    }
    unsigned char *h = ctx->rbuf().h_tables.as<unsigned char>();
    memcpy(h, rb.jobs.data(), jb); memcpy(h + o_steps, rb.steps.data(), sb);
    if (!rb.tabs.empty()) memcpy(h + o_tabs, rb.tabs.data(), rb.tabs.size());
    NVCA_HIP_CHECK(ctx, hipMemcpyAsync(ctx->rbuf().tables.p, h, total - 64, hipMemcpyHostToDevice, ctx->cs()));
    NVCA_HIP_CHECK(ctx, hipMemsetAsync(ctx->rbuf().hits.p, 0, sizeof(unsigned long long), ctx->cs()));
    const int lds = rb.plane_words * 8 + kRoiMaxWin * (8 + 2 + 2) + 16 + ((rb.lev_bytes + 15) & ~15) + 64;      // k_roi's carve-up
    if (const int e = roi_grant_lds(lds)) { ctx->set_error(std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString((hipError_t)e)); return NVCA_ERR_HIP; }
    const unsigned char *d = ctx->rbuf().tables.as<unsigned char>();
    { TimedLaunch t(ctx, NVCA_K_ROI);
      launch_roi(ctx->cs(), (const RoiJobDev *)d, (int)rb.steps.size(), (const RoiStep *)(d + o_steps), d + o_tabs, ctx->rbuf().hits.as<unsigned long long>(), rb.cap, rb.plane_words, lds,
                 ctx->rbuf().rej.as<unsigned long long>()); }
    if (rb.rej_words) NVCA_HIP_CHECK(ctx, hipMemcpyAsync(ctx->rbuf().h_rej.p, ctx->rbuf().rej.p, rb.rej_words * 8, hipMemcpyDeviceToHost, ctx->cs()));
    NVCA_LAUNCH_CHECK(ctx);
    NVCA_HIP_CHECK(ctx, hipMemcpyAsync(ctx->rbuf().h_hits.p, ctx->rbuf().hits.p, (first + 1) * 8, hipMemcpyDeviceToHost, ctx->cs()));
    return NVCA_OK;
}
// after the lane has drained: hand every job its candidates; NVCA_ERR_OVERFLOW (with hit_cap_wanted set) when the list was too short
static int roi_collect(nvca_ctx *ctx, RoiBatch &rb)
{
    unsigned long long *hh = ctx->rbuf().h_hits.as<unsigned long long>();
    const unsigned long long total = hh[0];
    const int nj = (int)rb.jobs.size();
    if (total > rb.cap) {
        const unsigned long long per = (total + (unsigned long long)nj - 1) / (unsigned long long)nj + 64;
        if (per <= (unsigned long long)kMaxHitCap && (long long)per > ctx->hit_cap_wanted) ctx->hit_cap_wanted = (int)per;
        ctx->set_error("raw candidate capacity exceeded (nvca_ctx_set_hit_capacity)");
        return NVCA_ERR_OVERFLOW;
    }
    // the list's head came back with the launch; how much of it to fetch that way next time follows the recent rounds (a second
    // copy is a second wait)
    ctx->roi_first_hint = std::max<size_t>((size_t)(total + total / 4), ctx->roi_first_hint - ctx->roi_first_hint / 16);
    const size_t first = rb.first;
    if (total > first) {
        NVCA_HIP_CHECK(ctx, hipMemcpyAsync(hh + 1 + first, ctx->rbuf().hits.as<unsigned long long>() + 1 + first, (total - first) * 8, hipMemcpyDeviceToHost, ctx->cs()));
        NVCA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->cs()));
    }
    // the dense jobs' reject bits: per ladder step of the job (api.cpp, fb_replay)
    for (DetectJob *o : rb.owners) {
        if (!o->dense) continue;
        const unsigned long long *hr = ctx->rbuf().h_rej.as<unsigned long long>();
        o->rej_bits.assign(o->ladder.size(), nullptr); o->rej_wpr.assign(o->ladder.size(), 0); o->rej_rows.assign(o->ladder.size(), 0); o->dense_hits.assign(o->ladder.size(), {});
        for (size_t k = 0; k < o->rej_info.size() && k < o->rinfo.size(); k++) {
            const DetectJob::RejInfo &ri = o->rej_info[k];
            const int li = o->rinfo[k].ladder;
            if (li < 0 || (size_t)li >= o->ladder.size() || (size_t)ri.off + (size_t)ri.wpr * ri.ny > rb.rej_words) { ctx->set_error("internal: reject bitmap of an unknown ladder step"); return NVCA_ERR_INTERNAL; }
            o->rej_bits[li] = hr + ri.off; o->rej_wpr[li] = ri.wpr; o->rej_rows[li] = ri.ny;
        }
    }
    // (the list is in the order the workgroups appended: every job sorts its own keys into the serial order when it advances)
    for (unsigned long long i = 0; i < total; i++) {
        const unsigned long long slot = hh[1 + i] >> 32;
        const unsigned key = (unsigned)hh[1 + i];
        if (slot >= (unsigned long long)nj || (key >> 26) >= rb.owners[slot]->rinfo.size()) {
            ctx->set_error("internal: candidate of an unknown job / step (device result rejected)"); return NVCA_ERR_INTERNAL;
        }
        rb.owners[slot]->rkeys[rb.owner_img[slot]].push_back(key);
    }
    return NVCA_OK;
}

// queue the job's next launch set; its candidates go to result slots [r0, r0 + slots()) of `total`
static int detect_job_enqueue(nvca_ctx *ctx, DetectJob &j, int r0, int total)
{
    (void)hipSetDevice(ctx->device);
    if (j.phase == 0) {
        for (int k = 0; k < kJobImages; k++) j.out[k].clear();
        if (j.kind == 2) return fb_enqueue_first(ctx, j, r0, total);
        if (j.kind == 1) return si_enqueue(ctx, j, r0, total);
        return plain_enqueue(ctx, j, r0, total);
    }
    if (j.phase == 2) return fb_enqueue_narrowed(ctx, j, r0, total);
    return NVCA_OK;
}

// after the stream has drained: consume the queued set's results.  phase 3: the call is complete (out[] holds the objects);
// phase 2: it needs another set (enqueue again)
static int detect_job_advance(nvca_ctx *ctx, DetectJob &j)
{
    int rc = NVCA_OK;
    std::vector<std::vector<nvca_rect>> raw;
    std::vector<char> grouped;
    std::vector<std::vector<int>> sc;
    bool have = j.dp != nullptr;
    const bool was_fused = j.fused;
    if (j.fused) {
        // candidates of the round's k_roi launch, sorted into the serial order (step, row, column) -> rectangle
        raw.assign(j.nimg, {}); sc.assign(j.nimg, {}); grouped.assign(j.nimg, 0);
        for (int k = 0; k < j.nimg; k++) {
            std::sort(j.rkeys[k].begin(), j.rkeys[k].end());
            raw[k].reserve(j.rkeys[k].size()); sc[k].reserve(j.rkeys[k].size());
            for (unsigned key : j.rkeys[k]) {
                const DetectJob::RoiStepInfo &ri = j.rinfo[key >> 26];
                const int iy = (key >> 13) & 8191, ix = key & 8191;
                if (j.dense && k == 0) {
                    // a dense launch reports every window that passes the cascade; the serial walk of the FULL grid (start column 0) visits only some
                    const size_t li = (size_t)ri.ladder;
                    if (li >= j.rej_bits.size() || j.rej_wpr[li] <= 0 || iy >= j.rej_rows[li] || ix >= j.rej_wpr[li] * 64) {
                        ctx->set_error("internal: dense candidate outside its reject bitmap (device result rejected)"); j.phase = 3; return NVCA_ERR_INTERNAL;
                    }
                    j.dense_hits[li].push_back((unsigned)(iy << 13 | ix));
                    if (!fb_visited(j.rej_bits[li] + (size_t)iy * j.rej_wpr[li], 0, ix)) continue;
                }
                if (ri.out_factor != 0) raw[k].push_back(nvca_rect{cv_round(ix * ri.out_factor), cv_round(iy * ri.out_factor), ri.winw, ri.winh});
                else raw[k].push_back(nvca_rect{cv_round(ix * ri.ystep), cv_round(iy * ri.ystep), ri.winw, ri.winh});
                sc[k].push_back(ri.ladder);
            }
            j.rkeys[k].clear();
        }
        j.fused = false;
        have = true;
    } else
    if (j.dp) rc = cascade_collect(ctx, *j.dp, j.cj, raw, j.kind == 0 ? &grouped : nullptr, j.kind == 2 ? &sc : nullptr);
    if (j.gp) { j.gp->inflight--; j.gp = nullptr; }
    if (rc == NVCA_ERR_OVERFLOW && j.regrown < 2 && ctx->hit_cap_wanted > ctx->hit_cap) {
        // More raw candidates than the lists hold.  OpenCV has no such limit (and a FIND_BIGGEST search would have stopped at
        // its first object long before: NOSE/kmsnosedetect.cpp:870-873, MOUTH/kmsmouthdetect.cpp:870-873, EAR/kmseardetect.cpp:712-715),
        // so the call must answer, not fail: the same launch set runs once more with lists of exactly the size the exact count
        // asks for (run_detect_jobs applies hit_cap_wanted before the next round), and the serial logic is replayed on the
        // complete candidate lists -- the result is what the reference returns.
        j.regrown++; j.dp = nullptr;
        if (j.phase == 1) j.phase = 0;              // the first (or only) set again; a narrowed FIND_BIGGEST set stays in phase 2
        return NVCA_OK;
    }
    if (rc) { j.phase = 3; return rc; }
    if (j.kind == 0) {
        if (have)
            for (int k = 0; k < j.nimg; k++) {
                if (j.gthr && !grouped[k]) group_rectangles(raw[k], j.gthr, 0.2);
                j.out[k].swap(raw[k]);
            }
        j.phase = 3;
    } else if (j.kind == 1) {
        if (have) { if (!j.raw_only) group_all(raw, j.min_neighbors); for (int k = 0; k < j.nimg; k++) j.out[k].swap(raw[k]); }
        j.phase = 3;
    } else {
        if (have) {
            for (size_t k = 0; k < raw[0].size(); k++) {
                // the large-image path numbers a candidate by its scale inside the plan (ladder_of maps it back), the small-image
                // path by its ladder step directly
                size_t li = (size_t)sc[0][k];
                if (!was_fused) { if (li >= j.ladder_of.size()) li = (size_t)-1; else li = (size_t)j.ladder_of[li]; }
                if (li >= j.hits.size()) { ctx->set_error("internal: candidate of an unknown ladder step"); j.phase = 3; return NVCA_ERR_INTERNAL; }
                j.hits[li].push_back(raw[0][k]);
            }
            for (int li : j.ladder_of) j.have[li] = 1;
        }
        j.dp = nullptr;
        const int r = fb_replay(ctx, j);
        if (r < 0) { j.phase = 3; return r; }
        j.phase = r == 1 ? 2 : 3;
    }
    j.dp = nullptr;
    return NVCA_OK;
}

// ---- working images of a batched part call ---------------------------------------------------------------------------
int part_arena(nvca_ctx *ctx, size_t bytes, uint8_t **base)
{
    if (ctx->pw().arena.ensure(bytes + 256)) { ctx->set_error("allocation failed (part detectors' images)"); return NVCA_ERR_NOMEM; }
    *base = ctx->pw().arena.as<uint8_t>();
    return NVCA_OK;
}
int part_luts(nvca_ctx *ctx, int n_keep, int n_scratch, uint8_t **keep)
{
    PartWorkspace &pw = ctx->pw();
    const size_t need_l = (size_t)(n_keep + n_scratch + 1) * 256, need_h = (size_t)(std::max(n_keep, n_scratch) + 1) * 256 * sizeof(unsigned);
    if (pw.luts.ensure(need_l)) { ctx->set_error("allocation failed (part detectors' LUTs)"); return NVCA_ERR_NOMEM; }
    const void *old = pw.hist.p;
    if (pw.hist.ensure(need_h)) { ctx->set_error("allocation failed (part detectors' histograms)"); return NVCA_ERR_NOMEM; }
    if (pw.hist.p != old) NVCA_HIP_CHECK(ctx, hipMemset(pw.hist.p, 0, pw.hist.bytes));       // k_lut leaves what it read zeroed again
    *keep = pw.luts.as<uint8_t>();
    return NVCA_OK;
}
// a small table for the next launch: page-locked staging ring -> device ring, copied on the current lane
int part_table(nvca_ctx *ctx, const void *host, size_t bytes, void **dev)
{
    PartWorkspace &pw = ctx->pw();
    static constexpr size_t kRing = 256 * 1024;
    if (pw.tables.ensure(kRing) || pw.h_tables.ensure(kRing)) { ctx->set_error("allocation failed (part detectors' tables)"); return NVCA_ERR_NOMEM; }
    const size_t room = round_up(bytes, 64);
    if (room > kRing) { ctx->set_error("part detectors: table too large"); return NVCA_ERR_ARG; }
    if (pw.tab_used + room > kRing) { NVCA_HIP_CHECK(ctx, hipDeviceSynchronize()); pw.tab_used = 0; }      // a full turn: earlier uploads must have been consumed
    uint8_t *h = pw.h_tables.as<uint8_t>() + pw.tab_used, *d = pw.tables.as<uint8_t>() + pw.tab_used;
    memcpy(h, host, bytes);
    NVCA_HIP_CHECK(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->cs()));
    pw.tab_used += room;
    *dev = d;
    return NVCA_OK;
}
int part_gray_eq(nvca_ctx *ctx, const void *const *bgr, int n, int w, int h, int stride, uint8_t *gray, size_t slot, uint8_t *luts)
{
    int rc;
    void *d_ptrs = nullptr;
    if ((rc = part_table(ctx, bgr, (size_t)n * sizeof(void *), &d_ptrs))) return rc;
    PreGeom g; make_geom(g, w, h, stride, 3, w, h);
    g.gpitch = w; g.gray_slot = slot;
    bool aligned = stride % 4 == 0 && w % 4 == 0 && slot % 4 == 0 && ((uintptr_t)gray & 3) == 0;
    for (int k = 0; k < n; k++) aligned = aligned && ((uintptr_t)bgr[k] & 3) == 0;
    unsigned *hist = ctx->pw().hist.as<unsigned>();
    { TimedLaunch t(ctx, NVCA_K_GRAY);
      launch_gray(ctx->cs(), (const uint8_t *const *)d_ptrs, g, 0, nullptr, nullptr, nullptr, nullptr, w, gray, hist, n, aligned); }
    { TimedLaunch t(ctx, NVCA_K_LUT); launch_lut(ctx->cs(), hist, w * h, luts, n, 1); }
    NVCA_LAUNCH_CHECK(ctx);
    return NVCA_OK;
}
int part_image_batch(nvca_ctx *ctx, const PartImageBatch &b, const uint8_t *luts)
{
    int rc;
    const int n = (int)b.src.size();
    if (!n) return NVCA_OK;
    GeomPlan *gp = nullptr;
    if ((rc = get_resize_plan(ctx, b.sw, b.sh, b.dw, b.dh, &gp))) return rc;
    // one table: n source pointers, then (gray sources with a LUT) n LUT indices
    std::vector<unsigned char> tab((size_t)n * sizeof(void *) + (size_t)n * sizeof(int));
    memcpy(tab.data(), b.src.data(), (size_t)n * sizeof(void *));
    const bool with_lut = !b.bgr && (int)b.lut_idx.size() == n;
    if (with_lut) memcpy(tab.data() + (size_t)n * sizeof(void *), b.lut_idx.data(), (size_t)n * sizeof(int));
    void *d_tab = nullptr;
    if ((rc = part_table(ctx, tab.data(), tab.size(), &d_tab))) return rc;
    unsigned *hist = b.post_eq ? ctx->pw().hist.as<unsigned>() : nullptr;
    uint8_t *scratch = ctx->pw().luts.as<uint8_t>() + ctx->pw().luts.bytes - (size_t)(n + 1) * 256;       // the scratch LUTs sit at the end
    if (b.post_eq && (size_t)(n + 1) * 256 > ctx->pw().luts.bytes) { ctx->set_error("internal: LUT storage"); return NVCA_ERR_ARG; }
    { TimedLaunch t(ctx, NVCA_K_RESIZE1);
      launch_work_resize(ctx->cs(), b.bgr, (const uint8_t *const *)d_tab, with_lut ? (const int *)((uint8_t *)d_tab + (size_t)n * sizeof(void *)) : nullptr, luts,
                         b.sh, b.sstride, gp->tab.mode, gp->d_xofs.as<int>(), gp->d_ialpha.as<short>(), gp->d_yofs.as<int>(), gp->d_ibeta.as<short>(),
                         gp->tab.xmax, b.dst, b.dw, b.dh, b.dw, b.slot, hist, n); }
    if (b.post_eq) {
        { TimedLaunch t(ctx, NVCA_K_LUT); launch_lut(ctx->cs(), hist, b.dw * b.dh, scratch, n, 1); }
        launch_apply_lut(ctx->cs(), b.dst, b.dw, b.dh, b.dw, scratch, b.dst, b.dw, n, b.slot, b.slot);
    }
    NVCA_LAUNCH_CHECK(ctx);
    return NVCA_OK;
}
int part_flip_batch(nvca_ctx *ctx, const uint8_t *src, uint8_t *dst, int w, int h, int n, size_t slot)
{
    launch_flip_h(ctx->cs(), src, w, h, w, dst, w, n, slot, slot);
    NVCA_LAUNCH_CHECK(ctx);
    return NVCA_OK;
}
int part_images_done(nvca_ctx *ctx, const int *lanes, int n)
{
    PartWorkspace &pw = ctx->pw();
    if (!pw.images_done) NVCA_HIP_CHECK(ctx, hipEventCreateWithFlags(&pw.images_done, hipEventDisableTiming));
    NVCA_HIP_CHECK(ctx, hipEventRecord(pw.images_done, ctx->cs()));
    for (int i = 0; i < n; i++)
        if (lanes[i] != ctx->cur_lane) NVCA_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->lane_streams[lanes[i]], pw.images_done, 0));
    return NVCA_OK;
}

// run a set of detectMultiScale calls to completion: one wait per round for all of them.  lanes (optional, [n]): the lane
// each job runs on -- jobs of one lane execute in order, lanes side by side
double g_jobs_fine_s[6] = {0, 0, 0, 0, 0, 0};        // NVCA_PART_STATS: roi_add_job, roi_launch, roi_collect, helper-thread advance, serial advance, small-path jobs (count)
double g_jobs_enqueue_s = 0, g_jobs_wait_s = 0, g_jobs_advance_s = 0;      // NVCA_PART_STATS (diagnostic, one context at a time): where run_detect_jobs spends the host's time
static inline double mono_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// One round of a job set in two halves, so that a caller may leave a round queued and come back for it (parts.cpp: a submitted part-detector
// batch keeps its face passes in flight while the batch before it is collected).  begin: every unfinished job queues its next launch
// set (small images: all in ONE k_roi launch); end: the lanes are waited for, the candidates handed out, every job advanced.
struct JobRound { RoiBatch rb; bool used[kLanes] = {false}; int rc = NVCA_OK; double t1 = 0; int roi_regrown = 0; };
static int jobs_round_begin(nvca_ctx *ctx, DetectJob *const *jobs, int n, const int *lanes, int lane0, JobRound &R, bool *pending_out)
{
    const bool g_job_stats = ctx->sw.part_stats > 0;
    const int roi_regrown = R.roi_regrown;
    {
        if (ctx->hit_cap_wanted > ctx->hit_cap) ctx->hit_cap = ctx->hit_cap_wanted;      // a set overflowed in the last round: it runs again with room (this call only)
        int pending = 0;
        for (int i = 0; i < n; i++) if (jobs[i]->phase != 3) pending++;
        if (!pending) { *pending_out = false; return NVCA_OK; }
        *pending_out = true;
        int &rc = R.rc; rc = NVCA_OK;
        bool (&used)[kLanes] = R.used;
        for (bool &u : used) u = false;
        const double t0 = g_job_stats ? mono_s() : 0;
        // small images first: every such job of the round goes into ONE k_roi launch (no plan, no per-job launches)
        R.rb.reset();
        RoiBatch &rb = R.rb;
        for (int i = 0; i < n && !rc; i++) {
            DetectJob &j = *jobs[i];
            if (j.phase == 3) continue;
            if (j.phase == 0 && j.regrown == 0) j.small = roi_eligible(ctx, j, n);
            if (!j.small) continue;
            ctx->cur_lane = lanes ? lanes[i] : lane0;
            if (rb.jobs.empty()) rb.lane = ctx->cur_lane;
            j.roi_prev_phase = j.phase;
            const double ta = g_job_stats ? mono_s() : 0;
            rc = roi_add_job(ctx, rb, j);
            if (g_job_stats) { g_jobs_fine_s[0] += mono_s() - ta; g_jobs_fine_s[5] += 1; }
            if (!rc && !j.fused) j.small = false;          // more ladder steps than the key holds: the large-image path takes it
            else used[ctx->cur_lane] = true;
        }
        int total = 0, r0 = 0;
        for (int i = 0; i < n; i++) if (jobs[i]->phase != 3 && !jobs[i]->small) total += jobs[i]->slots();
        for (int i = 0; i < n && !rc; i++) {
            if (jobs[i]->phase == 3 || jobs[i]->small) continue;
            ctx->cur_lane = lanes ? lanes[i] : lane0;
            used[ctx->cur_lane] = true;
            rc = detect_job_enqueue(ctx, *jobs[i], r0, total);
            r0 += jobs[i]->slots();
        }
        const double tl = g_job_stats ? mono_s() : 0;
        if (!rc && !rb.jobs.empty()) { ctx->cur_lane = rb.lane; used[rb.lane] = true; rc = roi_launch(ctx, rb, roi_regrown > 0); }
        const double t1 = g_job_stats ? mono_s() : 0;
        R.t1 = t1;
        if (g_job_stats) g_jobs_fine_s[1] += t1 - tl;
        if (g_job_stats) g_jobs_enqueue_s += t1 - t0;
    }
    ctx->cur_lane = lane0;
    return NVCA_OK;          // (a failed enqueue is carried in R.rc: the round is still waited for and closed by jobs_round_end)
}
static int jobs_round_end(nvca_ctx *ctx, DetectJob *const *jobs, int n, const int *lanes, int lane0, JobRound &R)
{
    const bool g_job_stats = ctx->sw.part_stats > 0;
    int rc = R.rc;
    bool (&used)[kLanes] = R.used;
    RoiBatch &rb = R.rb;
    int &roi_regrown = R.roi_regrown;
    const double t1 = R.t1;
    {
        for (int l = 0; l < kLanes; l++) {
            if (!used[l]) continue;
            const hipError_t he = hipStreamSynchronize(ctx->lane_streams[l]);
            if (he != hipSuccess && !rc) { ctx->set_error(std::string("hipStreamSynchronize: ") + hipGetErrorString(he)); rc = NVCA_ERR_HIP; }
        }
        ctx->cur_lane = lane0;
        const double t2 = g_job_stats ? mono_s() : 0;
        if (g_job_stats) g_jobs_wait_s += t2 - t1;
        if (g_job_stats && !rb.jobs.empty() && n >= ctx->sw.part_stats) {
            static int lines = 0;
            if (++lines > 200 && lines <= 212) {          // (a dozen rounds of the steady state: what a round holds and how long its launch took)
                int kinds[3] = {0, 0, 0}, narrowed = 0;
                for (DetectJob *o : rb.owners) { kinds[o->kind]++; if (o->roi_prev_phase == 2) narrowed++; }
                fprintf(stderr, "[nvca jobs] small-image round: %zu images (plain %d, scale-image %d, biggest-object %d of which narrowed %d), %zu workgroups, waited %.0f us\n",
                        rb.jobs.size(), kinds[0], kinds[1], kinds[2], narrowed, rb.steps.size(), (t2 - t1) * 1e6);
            }
        }
        struct Adv { double t; bool on; ~Adv() { if (on) g_jobs_advance_s += mono_s() - t; } } adv{t2, g_job_stats};
        drain_timer(ctx);
        bool roi_again = false;
        if (!rc && !rb.jobs.empty()) {
            ctx->cur_lane = rb.lane;
            const int r = roi_collect(ctx, rb);
            if (g_job_stats) g_jobs_fine_s[2] += mono_s() - t2;
            ctx->cur_lane = lane0;
            if (r == NVCA_ERR_OVERFLOW && roi_regrown < 2 && ctx->hit_cap_wanted > ctx->hit_cap) {
                // the round's candidate list was too short: its jobs are queued again, with room (see detect_job_advance)
                roi_regrown++; roi_again = true;
                if (g_job_stats) fprintf(stderr, "[nvca jobs] a small-image round overflowed its candidate list (cap %u for %zu jobs): queued again with %d per job\n", rb.cap, rb.jobs.size(), ctx->hit_cap_wanted);
                for (DetectJob *o : rb.owners) { o->phase = o->roi_prev_phase; o->fused = false; for (int k = 0; k < kJobImages; k++) o->rkeys[k].clear(); }
            } else if (r) rc = r;
        }
        // the small-path jobs' candidates are turned into rectangles, replayed (FIND_BIGGEST) and grouped job by job: independent
        // host work, shared with the context's helper threads (a job touches nothing but itself; set_error is locked)
        std::vector<DetectJob *> par;
        const double tp0 = g_job_stats ? mono_s() : 0;
        if (!rc && !roi_again)
            for (int i = 0; i < n; i++) if (jobs[i]->phase != 3 && jobs[i]->fused) par.push_back(jobs[i]);
        if (par.size() >= 4) {
            // the jobs with the most candidates first: the helpers take indices in order, the long ones must not come last
            auto weight = [](const DetectJob *j) { size_t w = 0; for (int k = 0; k < j->nimg; k++) w += j->rkeys[k].size(); return w; };
            std::stable_sort(par.begin(), par.end(), [&](const DetectJob *x, const DetectJob *y) { return weight(x) > weight(y); });
            ensure_pool(ctx);
            struct Arg { nvca_ctx *ctx; DetectJob **jobs; std::atomic<int> rc; } arg{ctx, par.data(), {0}};
            work_pool_run(ctx->pool, (int)par.size(), [](void *a, int i) {
                Arg *g = (Arg *)a;
                int r;
                try { r = detect_job_advance(g->ctx, *g->jobs[i]); }
                catch (const std::bad_alloc &) { r = NVCA_ERR_NOMEM; }
                catch (...) { r = NVCA_ERR_INTERNAL; }
                if (r) { g->jobs[i]->phase = 3; int z = 0; g->rc.compare_exchange_strong(z, r); }
            }, &arg);
            if (arg.rc.load()) rc = arg.rc.load();
            for (DetectJob *j : par) j->roi_prev_phase = -1;          // handled
        }
        const double tp1 = g_job_stats ? mono_s() : 0;
        if (g_job_stats) g_jobs_fine_s[3] += tp1 - tp0;
        for (int i = 0; i < n; i++) {
            if (jobs[i]->phase == 3) continue;
            if (rc) { if (jobs[i]->gp) { jobs[i]->gp->inflight--; jobs[i]->gp = nullptr; } jobs[i]->phase = 3; continue; }
            if (roi_again && std::find(rb.owners.begin(), rb.owners.end(), jobs[i]) != rb.owners.end()) continue;
            if (par.size() >= 4 && jobs[i]->roi_prev_phase == -1) { jobs[i]->roi_prev_phase = 0; continue; }
            ctx->cur_lane = lanes ? lanes[i] : lane0;
            const int r = detect_job_advance(ctx, *jobs[i]);
            if (r) rc = r;
        }
        ctx->cur_lane = lane0;
        if (g_job_stats) g_jobs_fine_s[4] += mono_s() - tp1;
        if (rc) {
            for (int i = 0; i < n; i++) { if (jobs[i]->gp) { jobs[i]->gp->inflight--; jobs[i]->gp = nullptr; } jobs[i]->phase = 3; }
            return rc;
        }
        return NVCA_OK;
    }
}
JobRound *job_round_new() { return new (std::nothrow) JobRound(); }
void job_round_free(JobRound *r) { delete r; }
// the first round of a job set, left queued (R from job_round_new).  Jobs that cannot take the small-image path make the caller
// wait for the round as before: *queued = false and nothing is launched.
int detect_jobs_begin(nvca_ctx *ctx, DetectJob *const *jobs, int n, const int *lanes, JobRound *R, bool *queued)
{
    *queued = false;
    for (int i = 0; i < n; i++) if (jobs[i]->phase != 0 || !roi_eligible(ctx, *jobs[i], n)) return NVCA_OK;
    const int lane0 = ctx->cur_lane;
    bool pending = false;
    const int rc = jobs_round_begin(ctx, jobs, n, lanes, lane0, *R, &pending);
    ctx->cur_lane = lane0;
    *queued = pending;
    return rc;
}
// ... and the rest of the set: the queued round is closed (queued == true), then round after round until every job is done
int detect_jobs_finish(nvca_ctx *ctx, DetectJob *const *jobs, int n, const int *lanes, JobRound *R, bool queued)
{
    const int lane0 = ctx->cur_lane;
    struct Restore { nvca_ctx *c; int l, cap, wanted; ~Restore() { c->cur_lane = l; c->hit_cap = cap; c->hit_cap_wanted = wanted; } } restore{ctx, lane0, ctx->hit_cap, ctx->hit_cap_wanted};
    ctx->hit_cap_wanted = 0;
    if (queued) { const int rc = jobs_round_end(ctx, jobs, n, lanes, lane0, *R); if (rc) return rc; }
    for (;;) {
        bool pending = false;
        int rc = jobs_round_begin(ctx, jobs, n, lanes, lane0, *R, &pending);
        if (!pending) return rc;
        if ((rc = jobs_round_end(ctx, jobs, n, lanes, lane0, *R))) return rc;
    }
}
int run_detect_jobs(nvca_ctx *ctx, DetectJob *const *jobs, int n, const int *lanes)
{
    JobRound R;
    return detect_jobs_finish(ctx, jobs, n, lanes, &R, false);
}

} // namespace nvca

int nvca::detect_scale_image_pair(nvca_ctx *ctx, const nvca_cascade *casc, const void *img_a, const void *img_b, int w, int h, int stride,
                                  int mem, double sf, int min_neighbors, int minw, int minh, std::vector<nvca_rect> *outs)
{
    NVCA_LOCK_OR_FAIL(ctx);
    if (!casc || !(sf > 1.0) || check_img(ctx, img_a, w, h, stride, 1, mem) || check_img(ctx, img_b, w, h, stride, 1, mem)) return NVCA_ERR_ARG;
    DetectJob j; j.kind = 1; j.casc = casc; j.img[0] = img_a; j.img[1] = img_b; j.nimg = 2; j.cols = w; j.rows = h; j.stride = stride; j.mem = mem;
    j.sf = sf; j.min_neighbors = min_neighbors; j.minw = minw; j.minh = minh; j.maxw = w; j.maxh = h;
    DetectJob *jp = &j;
    const int rc = run_detect_jobs(ctx, &jp, 1, nullptr);
    if (rc) return rc;
    outs[0].swap(j.out[0]); outs[1].swap(j.out[1]);
    return NVCA_OK;
}

nvca::DetectJob *nvca::detect_job_new() { return new (std::nothrow) DetectJob(); }
void nvca::detect_job_free(DetectJob *j) { delete j; }
const std::vector<nvca_rect> &nvca::detect_job_out(const DetectJob *j, int k) { return j->out[k]; }
int nvca::detect_job_add_image(DetectJob *j, const void *image)
{
    if (j->kind == 2 || j->phase != 0 || j->nimg >= kJobImages) return -1;
    j->img[j->nimg] = image;
    return j->nimg++;
}

// fill in a job from detectMultiScale's arguments (flags decide the kind); NVCA_ERR_ARG for bad arguments
int nvca::make_detect_job(nvca_ctx *ctx, DetectJob &j, const nvca_cascade *casc, const void *gray, int w, int h, int stride, int mem,
                          double sf, int min_neighbors, int flags, int minw, int minh, int maxw, int maxh, bool raw_only)
{
    if (check_img(ctx, gray, w, h, stride, 1, mem) || !casc || !(sf > 1.0)) return NVCA_ERR_ARG;
    if (maxw == 0 || maxh == 0) { maxw = w; maxh = h; }
    j = DetectJob();
    j.casc = casc; j.img[0] = gray; j.nimg = 1; j.cols = w; j.rows = h; j.stride = stride; j.mem = mem;
    j.sf = sf; j.min_neighbors = min_neighbors; j.minw = minw; j.minh = minh; j.maxw = maxw; j.maxh = maxh; j.raw_only = raw_only;
    if (flags & NVCA_HAAR_FIND_BIGGEST_OBJECT) {
        flags &= ~(NVCA_HAAR_SCALE_IMAGE | NVCA_HAAR_DO_CANNY_PRUNING);
        if (raw_only) return NVCA_ERR_ARG;
        j.kind = 2;
    } else if (flags & NVCA_HAAR_SCALE_IMAGE) j.kind = 1;
    else j.kind = 0;
    j.flags = flags;
    return NVCA_OK;
}

extern "C" {

static int detect_gray(nvca_ctx *ctx, const nvca_cascade *casc, const void *gray, int w, int h, int stride, int mem,
                       double sf, int min_neighbors, int flags, int minw, int minh, int maxw, int maxh, bool raw_only,
                       std::vector<nvca_rect> &out)
{
    NVCA_LOCK_OR_FAIL(ctx);
    DetectJob j;
    int rc = make_detect_job(ctx, j, casc, gray, w, h, stride, mem, sf, min_neighbors, flags, minw, minh, maxw, maxh, raw_only);
    if (rc) return rc;
    DetectJob *jp = &j;
    if ((rc = run_detect_jobs(ctx, &jp, 1, nullptr))) return rc;
    out.swap(j.out[0]);
    return NVCA_OK;
}

int nvca_detect_multiscale(nvca_ctx *ctx, const nvca_cascade *cascade, const void *gray, int w, int h, int stride,
                           int mem, double scale_factor, int min_neighbors, int flags, int min_w, int min_h,
                           int max_w, int max_h, nvca_rect *out, int cap, int *n_out)
try {
    if (!n_out || cap < 0 || (cap > 0 && !out)) return NVCA_ERR_ARG;
    std::vector<nvca_rect> r;
    int rc = detect_gray(ctx, cascade, gray, w, h, stride, mem, scale_factor, min_neighbors, flags, min_w, min_h, max_w,
                         max_h, false, r);
    if (rc) return rc;
    *n_out = (int)r.size();
    for (int i = 0; i < std::min<int>(cap, (int)r.size()); i++) out[i] = r[i];
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)

int nvca_detect_raw(nvca_ctx *ctx, const nvca_cascade *cascade, const void *gray, int w, int h, int stride, int mem,
                    double scale_factor, int flags, int min_w, int min_h, int max_w, int max_h, nvca_rect *out,
                    int cap, int *n_out)
try {
    if (!n_out || cap < 0 || (cap > 0 && !out)) return NVCA_ERR_ARG;
    if (flags & NVCA_HAAR_FIND_BIGGEST_OBJECT) return NVCA_ERR_ARG;
    std::vector<nvca_rect> r;
    int rc = detect_gray(ctx, cascade, gray, w, h, stride, mem, scale_factor, 0, flags, min_w, min_h, max_w, max_h, true, r);
    if (rc) return rc;
    *n_out = (int)r.size();
    for (int i = 0; i < std::min<int>(cap, (int)r.size()); i++) out[i] = r[i];
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)

int nvca_group_rectangles(nvca_ctx *ctx, nvca_rect *rects, int n, int group_threshold, double eps, int *n_out)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (!ctx || n < 0 || (n > 0 && !rects) || !n_out) return NVCA_ERR_ARG;
    std::vector<nvca_rect> v(rects, rects + n);
    group_rectangles(v, group_threshold, eps);
    for (size_t i = 0; i < v.size(); i++) rects[i] = v[i];
    *n_out = (int)v.size();
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)

} // extern "C"

// =========================================================================
// NuboFaceDetector stream
// =========================================================================
struct nvca_face_stream {
    nvca_ctx *ctx;
    const nvca_cascade *cascade;
    nvca_face_params p;
    Faces faces;
    int num_frame = 0, num_iter = 0, frames_with_no_detection = 0, num_frames_to_process = 0;
    int pending_events = 0;
};

namespace {
constexpr int kGOP = 4;                               // FACE/kmsfacedetect.cpp:28
constexpr int kMaxNoDetection = 1;                    // :30
constexpr int kNumFramesToProcess = 10;               // :23

struct FrameWork {
    bool analysed = false;
    int cols = 0, rows = 0, norm_scale = 0;
    std::vector<nvca_rect> det;
};

// the frame gating of kms_face_detect_process_frame (:794-803, :829-830); independent of detection results
bool face_gate(nvca_face_stream *s)
{
    bool received = true;
    if (s->p.detect_event) {                          // __receive_event :722-755
        received = false;
        if (s->pending_events > 0) { s->pending_events--; received = true; s->num_frames_to_process = kNumFramesToProcess; }
    }
    if (!received && s->num_frames_to_process <= 0) return false;     // early return: counters untouched
    s->num_frame++; s->num_iter++;
    bool run = false;
    const int px = s->p.process_x_every_4;
    if ((2 == px && (1 == s->num_frame % 2)) || ((2 != px) && (s->num_frame <= px))) {
        s->num_frames_to_process--;
        run = true;
    }
    if (kGOP == s->num_frame) s->num_frame = 0;
    return run;
}
} // namespace

extern "C" {

void nvca_face_params_default(nvca_face_params *p)
try {
    if (!p) return;
    p->width_to_process = 160; p->process_x_every_4 = 4; p->scale_factor_pct = 25; p->track_threshold = 40;
    p->euclidean_threshold = 8; p->area_threshold = 500; p->min_neighbors = 3; p->detect_event = 0;
}
NVCA_API_CATCH_VOID

int nvca_face_stream_create(nvca_ctx *ctx, const nvca_cascade *cascade, const nvca_face_params *params, nvca_face_stream **out)
try {
    if (!ctx || !cascade || !out) return NVCA_ERR_ARG;
    nvca_face_stream *s = new (std::nothrow) nvca_face_stream();
    if (!s) return NVCA_ERR_NOMEM;
    s->ctx = ctx; s->cascade = cascade;
    if (params) s->p = *params; else nvca_face_params_default(&s->p);
    *out = s;
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)
void nvca_face_stream_destroy(nvca_face_stream *s) { delete s; }
int nvca_face_stream_set_params(nvca_face_stream *s, const nvca_face_params *params)
try {
    if (!s || !params) return NVCA_ERR_ARG;
    s->p = *params;
    return NVCA_OK;
}
NVCA_API_CATCH((s ? s->ctx : nullptr))
int nvca_face_stream_motion_event(nvca_face_stream *s)
try {
    if (!s) return NVCA_ERR_ARG;
    s->pending_events++;
    return NVCA_OK;
}
NVCA_API_CATCH((s ? s->ctx : nullptr))

} // extern "C"

// A batch between its two halves: everything the second half (results -> temporal logic -> boxes) needs.
namespace nvca {
struct FaceTicket {
    bool pending = false;
    uint64_t serial = 0;
    int n = 0;
    std::vector<nvca_face_stream *> streams;
    std::vector<FrameWork> work;
    struct Group { GeomPlan *gp; std::vector<int> idx, gthr; std::vector<CascadeJob> jobs; };
    std::vector<Group> groups;
    hipEvent_t done = nullptr;
    // Two batches in flight run on two lanes (stream + planes each).  The second one's pre-processing (gray, LUT, integral:
    // bandwidth-bound) waits for the first one's band kernel and then runs beside its late-stage and grouping kernels (a few
    // hundred small workgroups that leave most of the GPU idle) -- not beside the band kernel itself, which wants every wave slot.
    int lane = 0;
    hipEvent_t band_done = nullptr;
};
}

// first half: gating, then every launch of the batch queued on the context's stream (result set `res`); no waiting
static int face_submit(nvca_ctx *ctx, int n, nvca_face_stream *const *streams, const nvca_frame *frames, int res, FaceTicket &tk)
{
    if (n < 0 || (n > 0 && (!streams || !frames))) return NVCA_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    ctx->timer.tick(0);
    // a batch that overflowed its candidate lists was reported as such (its frames' gates had advanced: there is no re-run on
    // this path); the streams go on with lists sized for what that batch produced, so the following frames are answered
    if (ctx->hit_cap_wanted > ctx->hit_cap && !(ctx->face_tickets[1] && ctx->face_tickets[1]->pending) && !(ctx->face_tickets[2] && ctx->face_tickets[2]->pending)) {
        ctx->hit_cap = ctx->hit_cap_wanted; ctx->hit_cap_wanted = 0;
    }
    Workspace &ws = *ctx->ws;
    struct UseRes { Workspace &w; UseRes(Workspace &x, int r) : w(x) { w.cur_res = r; } ~UseRes() { w.cur_res = 0; } } use_res(ws, res);   // every other entry point works on set 0
    // the lane of this batch: the synchronous call and the first submitted batch on lane 0, the second submitted batch on its own
    tk.lane = (res == 2 && ctx->sw.two_lanes) ? kFaceLane2 : 0;
    struct UseLane { nvca_ctx *c; int old; UseLane(nvca_ctx *x, int l) : c(x), old(x->cur_lane) { c->cur_lane = l; } ~UseLane() { c->cur_lane = old; } } use_lane(ctx, tk.lane);
    if (!tk.band_done) NVCA_HIP_CHECK(ctx, hipEventCreateWithFlags(&tk.band_done, hipEventDisableTiming));
    // "pre_cus" = n: the pre-processing of a submitted batch runs on a stream of its own that is confined to n compute units
    // (4 per XCD at 32: the mask's bits go round the XCDs first), beside the other batch's band kernel instead of behind it: the
    // band kernel loses those CUs' share of its workgroups for as long as the bandwidth-bound kernels run there and keeps every
    // wave slot of the others (spread over all CUs the same kernels keep band workgroups from starting everywhere: DESIGN 6)
    hipStream_t pre_stream = nullptr; int pre_k = 0;
    if (res > 0 && ctx->sw.two_lanes && ctx->sw.pre_cus > 0) {
        pre_k = res == 2 ? 1 : 0;
        if (ctx->pre_streams_cus != ctx->sw.pre_cus) {
            for (int k = 0; k < 2; k++) if (ctx->pre_streams[k]) { (void)hipStreamSynchronize(ctx->pre_streams[k]); (void)hipStreamDestroy(ctx->pre_streams[k]); ctx->pre_streams[k] = nullptr; }
            ctx->pre_streams_cus = ctx->sw.pre_cus;
        }
        if (!ctx->pre_streams[pre_k]) {
            uint32_t mask[16] = {0};
            for (int b = 0; b < std::min(ctx->sw.pre_cus, 512); b++) mask[b >> 5] |= 1u << (b & 31);
            if (hipExtStreamCreateWithCUMask(&ctx->pre_streams[pre_k], 16, mask) != hipSuccess) { (void)hipGetLastError(); ctx->pre_streams[pre_k] = nullptr; }
        }
        if (!ctx->pre_done[pre_k] && hipEventCreateWithFlags(&ctx->pre_done[pre_k], hipEventDisableTiming) != hipSuccess) ctx->pre_done[pre_k] = nullptr;
        if (ctx->pre_streams[pre_k] && ctx->pre_done[pre_k]) pre_stream = ctx->pre_streams[pre_k];
    }
    if (!pre_stream)
    for (int o = 1; o < 3; o++) {
        // a batch in flight on the other lane: this one's kernels start behind its band kernel (see FaceTicket)
        FaceTicket *ot = ctx->face_tickets[o];
        if (o != res && ot && ot->pending && ot->lane != tk.lane && ot->band_done) NVCA_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->cs(), ot->band_done, 0));
    }
    // while the pre-processing is being queued the lane's stream IS the confined one; the lane's own stream picks up behind it
    struct PreSwap {
        nvca_ctx *c; int lane; hipStream_t own, pre; hipEvent_t ev; bool on = false;
        int begin() { if (!pre || on) return 0; hipError_t e = hipEventRecord(ev, own); if (e == hipSuccess) e = hipStreamWaitEvent(pre, ev, 0); if (e != hipSuccess) return 1; c->lane_streams[lane] = pre; on = true; return 0; }
        int end() { if (!on) return 0; c->lane_streams[lane] = own; on = false; hipError_t e = hipEventRecord(ev, pre); if (e == hipSuccess) e = hipStreamWaitEvent(own, ev, 0); return e != hipSuccess; }
        ~PreSwap() { if (on) { c->lane_streams[lane] = own; (void)hipStreamSynchronize(pre); } }
    } pre_swap{ctx, tk.lane, ctx->lane_streams[tk.lane], pre_stream, pre_stream ? ctx->pre_done[pre_k] : nullptr};
    bool band_recorded = false;
    tk.n = n; tk.streams.assign(streams, streams + n); tk.work.assign(n, FrameWork()); tk.groups.clear();
    std::vector<FrameWork> &work = tk.work;
    // ---- pass 1: geometry + gating, in frame order
    for (int i = 0; i < n; i++) {
        nvca_face_stream *s = streams[i];
        const nvca_frame &f = frames[i];
        if (!s || s->ctx != ctx || check_img(ctx, f.data, f.width, f.height, f.stride, 3, f.mem)) return NVCA_ERR_ARG;
        if (s->p.width_to_process <= 0) { ctx->set_error("width-to-process must be > 0"); return NVCA_ERR_ARG; }
    }
    // The gates advance per-stream counters; plans and buffers are resolved after them and may still fail (too many scales,
    // allocation).  A failed submit must leave every stream as it found it -- callers (the GStreamer shim) re-submit the
    // frames one by one -- so the counters are restored on any error return.
    struct GateSnap { nvca_face_stream *s; int num_frame, num_iter, to_process, pending; };
    struct GateRollback {
        std::vector<GateSnap> v; bool armed = true;
        ~GateRollback() { if (armed) for (const GateSnap &g : v) { g.s->num_frame = g.num_frame; g.s->num_iter = g.num_iter; g.s->num_frames_to_process = g.to_process; g.s->pending_events = g.pending; } }
    } gates;
    for (int i = 0; i < n; i++) {
        nvca_face_stream *s = streams[i];
        bool seen = false;
        for (const GateSnap &g : gates.v) if (g.s == s) { seen = true; break; }
        if (!seen) gates.v.push_back(GateSnap{s, s->num_frame, s->num_iter, s->num_frames_to_process, s->pending_events});
    }
    for (int i = 0; i < n; i++) {
        nvca_face_stream *s = streams[i];
        const nvca_frame &f = frames[i];
        FrameWork &w = work[i];
        // kms_face_detect_conf_images :304 -- INTEGER ratio kept in a float; kms_face_send_event :190
        const float fscale = (float)(f.width / s->p.width_to_process);
        w.norm_scale = f.width / s->p.width_to_process;
        double scale = fscale;
        w.rows = f.height; w.cols = f.width;                           // process_frame :770-783
        if (cv_round(f.height / scale) > 0) w.rows = cv_round(f.height / scale); else scale = 1;
        if (cv_round(f.width / scale) > 0) w.cols = cv_round(f.width / scale); else scale = 1;
        w.analysed = face_gate(s);
    }
    // ---- pass 2: one launch set per distinct geometry; result slots are numbered over the whole batch
    std::vector<char> done(n, 0);
    int gbase = 0;
    size_t stage_off = 0;                      // host frames of all groups share this batch's staging buffer
    {
        ResultBufs &rb = ws.res[ws.cur_res];
        size_t need = 0; int na = 0;
        for (int i = 0; i < n; i++) if (work[i].analysed) { na++; need += staging_need(frames + i, nullptr, 1); }
        if (rb.srcptrs.ensure((size_t)std::max(na, 1) * sizeof(void *)) || rb.h_srcptrs.ensure((size_t)std::max(na, 1) * sizeof(void *)) ||
            (need && rb.staging.ensure(need))) { ctx->set_error("allocation failed (frame staging)"); return NVCA_ERR_NOMEM; }
    }
    for (int i = 0; i < n; i++) {
        if (!work[i].analysed || done[i]) continue;
        const nvca_face_stream *s0 = streams[i];
        const nvca_frame &f0 = frames[i];
        tk.groups.emplace_back();
        FaceTicket::Group &grp = tk.groups.back();
        std::vector<int> &idx = grp.idx;
        for (int j = i; j < n; j++) {
            const nvca_face_stream *sj = streams[j];
            const nvca_frame &fj = frames[j];
            if (work[j].analysed && !done[j] && sj->cascade == s0->cascade && fj.width == f0.width && fj.height == f0.height &&
                fj.stride == f0.stride && work[j].cols == work[i].cols && work[j].rows == work[i].rows &&
                sj->p.scale_factor_pct == s0->p.scale_factor_pct) { idx.push_back(j); done[j] = 1; }
        }
        const int batch = (int)idx.size();
        const int cols = work[i].cols, rows = work[i].rows;
        GeomPlan *gp = nullptr;
        const double sf = 1 + s0->p.scale_factor_pct * 1.0 / 100;      // MULTI_SCALE_FACTOR :142
        int rc = get_face_plan(ctx, s0->cascade, f0.width, f0.height, f0.stride, 3, cols, rows, sf, cols / 20, rows / 20, 0, 0, &gp);
        if (rc) return rc;
        grp.gp = gp; gp->inflight++;
        // Host frames: the batch goes through in chunks -- chunk c+1's H2D copies run on the copy stream while the
        // kernels of chunk c execute (with pageable memory the host blocks in the copy, the queued kernels do not).
        // Every chunk reuses planes [0, chunk); only its candidate list / box table are its own (CascadeJob).
        bool any_host = false;
        for (int b = 0; b < batch; b++) any_host = any_host || frames[idx[b]].mem == NVCA_MEM_HOST;
        const int chunk_env = ctx->sw.ingest_chunk;
        const int chunk = (any_host && chunk_env > 0 && batch >= 2 * chunk_env) ? chunk_env : batch;
        const bool piped = chunk < batch;
        if ((rc = ensure_ws(ctx, gp->g, chunk))) return rc;

        std::vector<int> &gthr = grp.gthr;
        gthr.resize(batch);
        for (int b = 0; b < batch; b++) { const int mn = streams[idx[b]]->p.min_neighbors; gthr[b] = mn != 0 ? std::max(mn, 1) : 0; }
        std::vector<CascadeJob> &jobs = grp.jobs;
        for (int s0 = 0; s0 < batch; s0 += chunk) {
            const int nc = std::min(chunk, batch - s0);
            if (pre_swap.begin()) { ctx->set_error("event hand-over to the confined stream failed"); return NVCA_ERR_HIP; }
            if ((rc = stage_frames(ctx, frames, idx.data() + s0, nc, 3, gbase + s0, piped ? ctx->copy_stream : ctx->cs(), &stage_off, &gp->rowcopy))) return rc;
            if (piped) {
                while (ctx->chunk_events.size() <= jobs.size()) {
                    hipEvent_t ev; NVCA_HIP_CHECK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                    ctx->chunk_events.push_back(ev);
                }
                NVCA_HIP_CHECK(ctx, hipEventRecord(ctx->chunk_events[jobs.size()], ctx->copy_stream));
                NVCA_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->cs(), ctx->chunk_events[jobs.size()], 0));
            }
            int hist_clean = ws.ln().hist_clean;                                // k_lut leaves the histograms it read zeroed again
            if (hist_clean < nc) {
                NVCA_HIP_CHECK(ctx, hipMemsetAsync(ws.ln().hist.p, 0, (size_t)nc * 256 * sizeof(unsigned), ctx->cs()));
                hist_clean = nc;
            }
            ws.ln().hist_clean = 0;                                             // dirty until the LUT kernel is queued
            CascadeJob job; job.r0 = gbase + s0; job.n = nc; job.total = n;
            unsigned long long *z_hits = nullptr, *z_deep = nullptr;
            if ((rc = cascade_counters(ctx, gp->det, job, &z_hits, &z_deep))) return rc;
            { TimedLaunch t(ctx, NVCA_K_GRAY);                             // cv::resize + cvtColor :805-806 (+ histogram)
              launch_gray(ctx->cs(), ws.res[ws.cur_res].srcptrs.as<const uint8_t *>() + gbase + s0, gp->g, gp->tab.mode, gp->d_xofs.as<int>(),
                          gp->d_ialpha.as<short>(), gp->d_yofs.as<int>(), gp->d_ibeta.as<short>(), gp->tab.xmax,
                          ws.ln().gray.as<uint8_t>(), ws.ln().hist.as<unsigned>(), nc, frames_aligned4(frames, idx.data() + s0, nc)); }
            { TimedLaunch t(ctx, NVCA_K_LUT);                              // equalizeHist :807 (applied inside the integral pass)
              launch_lut(ctx->cs(), ws.ln().hist.as<unsigned>(), cols * rows, ws.ln().lut.as<uint8_t>(), nc, 1, z_hits, z_deep); }
            job.counters_zeroed = true;
            ws.ln().hist_clean = hist_clean;
            run_integral(ctx, gp->g, ws.ln().lut.as<uint8_t>(), nc);
            if (streams[idx[0]]->cascade->c.has_tilted && (rc = run_tilted(ctx, gp->g, ws.ln().lut.as<uint8_t>(), nc))) return rc;
            if (pre_swap.end()) { ctx->set_error("event hand-over from the confined stream failed"); return NVCA_ERR_HIP; }
            if ((rc = cascade_enqueue(ctx, gp->det, gp->g.sum_slot, gp->g.spitch, job, gthr.data() + s0, true, tk.band_done))) return rc;   // detectMultiScale :809-811
            band_recorded = true;
            jobs.push_back(job);
        }
        gbase += batch;
    }
    if (!band_recorded) NVCA_HIP_CHECK(ctx, hipEventRecord(tk.band_done, ctx->cs()));       // nothing analysed: nothing to wait for
    if (!tk.done) NVCA_HIP_CHECK(ctx, hipEventCreateWithFlags(&tk.done, hipEventDisableTiming));
    NVCA_HIP_CHECK(ctx, hipEventRecord(tk.done, ctx->cs()));
    tk.pending = true;
    gates.armed = false;
    return NVCA_OK;
}

static void face_release(FaceTicket &tk)
{
    for (FaceTicket::Group &g : tk.groups) if (g.gp) g.gp->inflight--;
    tk.groups.clear(); tk.pending = false;
}

// second half: wait for the batch, turn candidates into tracked faces and boxes (frame order)
static int face_collect(nvca_ctx *ctx, int res, FaceTicket &tk, nvca_rect *out, int *ids, int cap, int *n_out)
{
    (void)hipSetDevice(ctx->device);
    Workspace &ws = *ctx->ws;
    struct UseRes { Workspace &w; UseRes(Workspace &x, int r) : w(x) { w.cur_res = r; } ~UseRes() { w.cur_res = 0; } } use_res(ws, res);
    struct UseLane { nvca_ctx *c; int old; UseLane(nvca_ctx *x, int l) : c(x), old(x->cur_lane) { c->cur_lane = l; } ~UseLane() { c->cur_lane = old; } } use_lane(ctx, tk.lane);
    const int n = tk.n;
    hipError_t he = hipEventSynchronize(tk.done);
    if (he != hipSuccess) { ctx->set_error(std::string("hipEventSynchronize: ") + hipGetErrorString(he)); face_release(tk); return NVCA_ERR_HIP; }
    drain_timer(ctx);
    int rc = NVCA_OK;
    for (FaceTicket::Group &grp : tk.groups) {
        int gi0 = grp.jobs.empty() ? 0 : grp.jobs.front().r0;
        for (const CascadeJob &job : grp.jobs) {
            std::vector<std::vector<nvca_rect>> raw;
            std::vector<char> grouped;
            if ((rc = cascade_collect(ctx, grp.gp->det, job, raw, &grouped))) { face_release(tk); return rc; }
            for (int b = 0; b < job.n; b++) {
                const int gi = job.r0 - gi0 + b;                         // position inside the group
                if (grp.gthr[gi] != 0 && !grouped[b]) group_rectangles(raw[b], grp.gthr[gi], 0.2);
                tk.work[grp.idx[gi]].det.swap(raw[b]);
            }
        }
    }
    // ---- pass 3: temporal logic + emission, in frame order
    for (int i = 0; i < n; i++) {
        nvca_face_stream *s = tk.streams[i];
        FrameWork &w = tk.work[i];
        if (w.analysed) {
            if (!w.det.empty()) s->faces.track(w.det, s->p.track_threshold);           // :813-816
            else if (s->frames_with_no_detection < kMaxNoDetection) s->frames_with_no_detection += 1;   // :817-826
            else { s->frames_with_no_detection = 0; s->faces.clear(); }
        }
        const int nf = (int)s->faces.faces.size();
        n_out[i] = nf;
        for (int k = 0; k < std::min(nf, cap); k++) {                  // kms_face_send_event :208-211
            const nvca_rect &r = s->faces.faces[k].box;
            nvca_rect &o = out[(size_t)i * cap + k];
            o.x = (int)((unsigned)r.x * (unsigned)w.norm_scale); o.y = (int)((unsigned)r.y * (unsigned)w.norm_scale);
            o.w = (int)((unsigned)r.w * (unsigned)w.norm_scale); o.h = (int)((unsigned)r.h * (unsigned)w.norm_scale);
            if (ids) ids[(size_t)i * cap + k] = s->faces.faces[k].id;
        }
    }
    face_release(tk);
    return NVCA_OK;
}

void nvca::free_face_ticket(FaceTicket *t)
{
    if (!t) return;
    if (t->done) (void)hipEventDestroy(t->done);
    if (t->band_done) (void)hipEventDestroy(t->band_done);
    delete t;
}

static FaceTicket &ticket_slot(nvca_ctx *ctx, int k)
{
    if (!ctx->face_tickets[k]) ctx->face_tickets[k] = new FaceTicket();
    return *ctx->face_tickets[k];
}

extern "C" {

int nvca_face_batch_process(nvca_ctx *ctx, int n, nvca_face_stream *const *streams, const nvca_frame *frames,
                            nvca_rect *out, int *ids, int cap, int *n_out)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (n < 0 || (n > 0 && (!streams || !frames || !n_out)) || cap < 0 || (cap > 0 && !out)) return NVCA_ERR_ARG;
    // a stream's frames are consumed in order: the synchronous call may not overtake a submitted batch of the same stream
    for (int k = 1; k < 3; k++)
        if (ctx->face_tickets[k] && ctx->face_tickets[k]->pending)
            for (int i = 0; i < n; i++)
                for (nvca_face_stream *s : ctx->face_tickets[k]->streams)
                    if (s == streams[i]) { ctx->set_error("a stream of this batch has a submitted batch in flight: collect it first"); return NVCA_ERR_ARG; }
    FaceTicket &tk = ticket_slot(ctx, 0);
    int rc = face_submit(ctx, n, streams, frames, 0, tk);
    if (rc) { (void)hipStreamSynchronize(ctx->cs()); face_release(tk); return rc; }
    return face_collect(ctx, 0, tk, out, ids, cap, n_out);
}
NVCA_API_CATCH(ctx)

// Pipelined form of nvca_face_batch_process for a serving loop: submit() queues a batch and returns, collect() waits for
// the oldest submitted batch and delivers its boxes.  Up to two batches may be in flight, so the host-side work between
// batches (result unpacking, the caller's own bookkeeping) overlaps the GPU.  Batches are collected in submission order.
int nvca_face_batch_submit(nvca_ctx *ctx, int n, nvca_face_stream *const *streams, const nvca_frame *frames, int *ticket)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (!ticket) return NVCA_ERR_ARG;
    int k = 0;
    for (int c = 1; c < 3; c++) if (!(ctx->face_tickets[c] && ctx->face_tickets[c]->pending)) { k = c; break; }
    if (!k) { ctx->set_error("two batches are in flight: collect one first"); return NVCA_ERR_ARG; }
    FaceTicket &tk = ticket_slot(ctx, k);
    tk.serial = ++ctx->face_serial;
    int rc = face_submit(ctx, n, streams, frames, k, tk);
    if (rc) { (void)hipStreamSynchronize(ctx->lane_streams[tk.lane]); face_release(tk); return rc; }
    *ticket = k;
    return NVCA_OK;
}
NVCA_API_CATCH(ctx)
int nvca_face_batch_collect(nvca_ctx *ctx, int ticket, nvca_rect *out, int *ids, int cap, int *n_out)
try {
    NVCA_LOCK_OR_FAIL(ctx);
    if (ticket < 1 || ticket > 2 || !ctx->face_tickets[ticket] || !ctx->face_tickets[ticket]->pending) { ctx->set_error("no such batch in flight"); return NVCA_ERR_ARG; }
    FaceTicket &tk = *ctx->face_tickets[ticket];
    const int other = 3 - ticket;
    if (ctx->face_tickets[other] && ctx->face_tickets[other]->pending && ctx->face_tickets[other]->serial < tk.serial) {
        ctx->set_error("batches are collected in submission order"); return NVCA_ERR_ARG;
    }
    if (cap < 0 || (cap > 0 && !out) || (tk.n > 0 && !n_out)) return NVCA_ERR_ARG;
    return face_collect(ctx, ticket, tk, out, ids, cap, n_out);
}
NVCA_API_CATCH(ctx)

int nvca_face_stream_process(nvca_face_stream *s, const nvca_frame *frame, nvca_rect *out, int *ids, int cap, int *n_out)
try {
    if (!s || !frame) return NVCA_ERR_ARG;
    nvca_face_stream *arr[1] = {s};
    return nvca_face_batch_process(s->ctx, 1, arr, frame, out, ids, cap, n_out);
}
NVCA_API_CATCH((s ? s->ctx : nullptr))

// =========================================================================
// NuboTracker stream (device path lands with the tracker kernels)
// =========================================================================
void nvca_tracker_params_default(nvca_tracker_params *p)
try {
    if (!p) return;
    p->threshold = 20; p->min_area = 50; p->max_area = 30000; p->distance = 35; p->mhi_duration = 0.2; p->seg_thresh = 32;
}
NVCA_API_CATCH_VOID

} // extern "C"
