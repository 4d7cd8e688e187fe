// nvca_internal.h -- shared declarations of libnubovca_hip (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <memory>
#include <mutex>
#include "../../include/nubovca.h"
#include "host_ranges.h"

namespace nvca {

// --------------------------------------------------------------------------
// Cascade as loaded from old-format XML (OpenCV CvHaarClassifierCascade).
// --------------------------------------------------------------------------
struct HaarNode {
    int   rect[3][4];   // x,y,w,h ; zero when absent
    float weight[3];
    int   nrect;        // 2 or 3 (icvCreateHidHaarClassifierCascade's rect[2] test)
    float threshold;
    int   left, right;  // >0 child node index, <=0 -> alpha[-idx]
    int   tilted;
};
struct HaarClassifier { int first_node, nnodes, first_alpha; };
struct HaarStage { int first_cls, ncls; float threshold; /* as in the XML */ };

struct Cascade {
    int ow = 0, oh = 0;
    std::vector<HaarStage> stages;
    std::vector<HaarClassifier> cls;
    std::vector<HaarNode> nodes;
    std::vector<float> alpha;
    bool stump_based = true;
    bool has_tilted = false; // some feature reads the tilted integral
    bool generic() const { return !stump_based || has_tilted; }   // evaluated by the general kernels (kernels_cascade.hip)
    uint64_t uid = 0;       // identity for plan caching
    mutable std::vector<unsigned char> stage_rec_cache;   // StageRec[] (plan.cpp, built on first use: the summation-order proof is per cascade)
};

// returns NVCA_OK or NVCA_ERR_PARSE / NVCA_ERR_UNSUPPORTED; err gets a message
int parse_cascade_xml(const char *text, size_t len, Cascade &out, std::string &err);

// --------------------------------------------------------------------------
// Device-side records (plain structs shared by host table builder and kernels)
// --------------------------------------------------------------------------
struct StageRec {
    int first, count; float thr;
    int flags;              // bit 0: two_rects (every stump has 2 rects); bit 1: votes may be summed in any order;
                            // bit 2: every vote is an integer multiple of 2^vote_exp and the stage's sums stay below 2^31 of them:
                            // the tile kernels add the integer votes (TStumpRec::a0i / a1i) and compare with thr_i -- bit for bit
                            // the f64 sum OpenCV forms, whatever the order
    int thr_i;              // pass  <=>  integer sum >= thr_i   (== !(sum * 2^vote_exp < (double)thr))
    int vote_exp;
    int spec_run;           // stages from this one on, this one included, whose votes are exact in any order (flag bit 1): how far a
                            // round of the tile kernels may look ahead (tile_stages, several stages per round)
    int pad1;
};

struct ScaleRec {           // one evaluated scale
    int    winw, winh;
    int    plane_off;       // element offset of this scale's sum/sqsum planes inside a slot (0: the full-image planes)
    int    pitch;           // row pitch of those planes (elements)
    int    endX, endY;      // scan grid: ix in [0,endX), iy in [0,endY)
    int    eq[4];           // equRect corner offsets
    int    xpos_off, ypos_off;           // into the position tables (indexed by ix / iy)
    int    sq32;            // 1: the squared-pixel sum of the variance window is below 2^32 at this scale (ew * eh * 255^2):
                            //    the low-word plane alone gives it exactly (modulo-2^32 corner arithmetic)
    int    task_off;        // first stage-0 wave task (64 windows) of this scale
    int    wpr;             // wave tasks (64-bit reject words) per scan row
    int    adaptive;        // 1: OpenCV's adaptive x step applies (scale-cascade scan); 0: every grid point is visited
    double inv_area;
    double factor;
    const struct TStumpRec *trecs;   // the cascade's stumps at this scale's factor (device; shared by every plan that uses the factor)
    const struct GNodeRec *grecs;    // general cascades (tree weak classifiers / tilted features): every node at this scale's factor
    const struct LStumpRec *lrecs;   // the same stumps as trecs in the compact per-lane form (tile kernels: a stump per lane)
};

// A node of a weak classifier in its general form: up to three rectangles, upright (corners of the integral image) or
// tilted (corners of the tilted integral), a threshold and two children.  Corner offsets are window-relative pixels in the
// order + - - + (cvSetImagesForHaarClassifierCascade: p0 - p1 - p2 + p3), per (cascade, factor) like TStumpRec.
struct GNodeRec {
    short dx[3][4], dy[3][4];
    float w[3];
    float thr;
    int left, right;        // > 0: node index inside the weak classifier; <= 0: leaf, alpha index = -value (absolute)
    int flags;              // low byte: rectangles (2 or 3); bit 8: tilted
    int pad;
};
static_assert(sizeof(GNodeRec) == 80, "GNodeRec layout");

struct StripRec { int scale, iy0, nrows, ix0, ncols, pad0, pad1, pad2; };   // a block's share of the scan: nrows x ncols windows
// A tile is a block of nx x ny windows (<= 32 x 32) of one scale.  The windows of a scale only ever touch the integral
// image on a near-lattice of positions (window origin + scaled rectangle corner), a small fraction of the pixels they
// span; the tile kernel stages exactly those sample rows x columns, compacted, in LDS and looks corners up through a
// column map and a row map.  Tile size in windows is therefore the same at every scale.
struct TileRec {
    int scale, ix0, iy0, nx, ny;
    int x0, y0;             // plane coordinates of the tile's first window: origin of the maps
    int ncol, nrow;         // distinct sample columns / rows staged
    int span_x, span_y;     // map extents: the largest column / row offset from (x0, y0) ever looked up, + 1
    int col_off, row_off;   // first entry of the column / row coordinate lists (plane coordinates, u16) in `tcoords`
    int pad_t;
    int pad0, pad1;
};
struct TStumpRec {          // a stump with separate corner columns / rows (window-relative pixels)
    int x0[3], x1[3], y0[3], y1[3];
    float w[3];
    int nrect;              // low byte: rectangles (2 or 3); bits 8..11 "share": bit 8 / 9: rectangle 1 has rectangle 0's rows /
                            // columns; bit 10 / 11: rectangle 2 likewise
    double thr, a0, a1;
    int a0i, a1i;           // the votes as integers (a / 2^vote_exp of the stump's stage) where StageRec flag bit 2 is set
                            // 96 bytes: the tile kernels fetch a record with two wide scalar loads (16 + 8 dwords)
};
static_assert(sizeof(TStumpRec) == 96, "TStumpRec layout is read dword by dword in kernels_cascade.hip");
// The same stump in 48 bytes, for lanes that each evaluate a DIFFERENT stump (the tile kernels once few windows of a wave are
// left: lane = (window, stump) pair): three 16-byte vector loads per lane instead of a record held in scalar registers.
// Corner coordinates are window-relative pixels TIMES TWO (byte offsets into the tile's u16 maps), low half = first corner;
// rectangle 2 is absent iff both its words are zero.  Threshold and votes are the file's floats (the kernels widen them
// exactly); a stage's stumps are in the order of the TStumpRec table.
struct alignas(16) LStumpRec {
    unsigned xx0, yy0, xx1, yy1, xx2, yy2;      // per rectangle: xx = 2 x0 | 2 x1 << 16, yy = 2 y0 | 2 y1 << 16
    float w0, w1, w2, thr, a0, a1;
};
static_assert(sizeof(LStumpRec) == 48, "LStumpRec is read as three int4 in kernels_cascade.hip");
// A band is one row of tiles (<= 32 window rows of one scale, the full scan width): k_band walks it left to right in one
// workgroup, so stage 0 and OpenCV's adaptive x step (which depends on the stage-0 results to the left) need no pre-pass.
// Per scale: the distinct corner columns / rows (window-relative pixels) of the late stages' stumps.  k_deep stages that
// ncol x nrow patch of a surviving window in LDS after the first late stage (ncol == 0: scale not eligible, global gathers).
struct DeepRec { int col_off, ncol, row_off, nrow, span_x, span_y, pad0, pad1; };
static constexpr int kDeepMaxSide = 64;            // patch side (distinct columns / rows)
static constexpr int kDeepMaxSpan = 1280;          // largest corner offset + 1 the patch maps cover
struct BandRec { int scale, iy0, ny, first_tile, ntiles, pad0, pad1, pad2; };
static constexpr int kTileWin = 32;                 // windows per tile row (window id = ry * 32 + rx)
static constexpr int kTileRows = 24;                // window rows per tile
static constexpr int kTileSlots = kTileWin * kTileRows;   // windows per tile = queue capacity = threads of the tile kernels
// 768 threads: two tile workgroups take 24 of a CU's 32 wave slots (and 85 registers each) -- the bandwidth-bound
// pre-processing kernels of the next batch fit beside them without ever keeping a tile workgroup from starting
static constexpr int kTileThreads = kTileSlots;
static constexpr int kTileLdsBudget = 76 * 1024;   // two tiles resident per CU (160 KiB LDS) and 8 KiB left for small workgroups beside them
static constexpr int kTileMaxCols = 256;            // staged columns per tile (4 per lane)
// LDS bytes the tile kernel needs for a tile (host sizing and kernel carve-up agree through these)
__host__ __device__ inline int tile_pitch(int ncol) { return ncol | 1; }
// fixed part (carve_tile in kernels_cascade.hip): stage accumulators (8 B a queue slot) | two window queues | window origins |
// counters and stage statistics (32 words) | per-window variance normaliser
__host__ __device__ inline int tile_lds_fixed()
{
    return kTileSlots * 8 + 2 * kTileSlots * 2 + 4 * kTileWin + 128 + kTileSlots * 8;
}
__host__ __device__ inline int tile_lds_bytes(int ncol, int nrow, int span_x, int span_y)
{
    return 4 * nrow * tile_pitch(ncol) + 2 * ((span_x + 3) & ~3) + 2 * ((span_y + 3) & ~3) + tile_lds_fixed();
}

static constexpr int kStripMaxWin = 512;   // windows per strip (LDS budget of the evaluator)
static constexpr int kIntegralBand = 16;   // rows per integral band

// resize tables (cv::resize INTER_LINEAR 8U fixed point)
struct ResizeTab {
    int sw = 0, sh = 0, dw = 0, dh = 0;
    int mode = 0;           // 0 identity, 1 bilinear, 2 area-fast 2x2
    int xmax = 0;
    std::vector<int> xofs, yofs;
    std::vector<short> ialpha, ibeta;
};
void build_resize_tab(int sw, int sh, int dw, int dh, ResizeTab &t);

// --------------------------------------------------------------------------
// Environment switches: A/B and diagnostic knobs, all off-path by default and none of them changes a
// result.  The environment is read ONCE per process -- when the first context is created (nvca_ctx_create) -- never on a hot
// entry point.  A context starts from those process defaults; nvca_ctx_set_option changes one of them for that context.
// --------------------------------------------------------------------------
struct Switches {
    bool group_zero_copy = true;     // NVCA_GROUP_ZEROCOPY=0: box tables through a copy instead of direct stores to the host buffer
    bool skip_cascade = false;       // NVCA_SKIP_CASCADE: timing experiments on the pre-processing kernels only
    bool host_group = false;         // NVCA_HOST_GROUP: cv::groupRectangles on the host
    int  band_map = 0;               // NVCA_BAND_MAP=1/2: frame-major band walk
    int  band = -1;                  // NVCA_BAND=0/1: force pre-pass + tile kernels / band kernel (-1: by batch size)
    bool host_profile = false;       // NVCA_HOST_PROFILE: host-side timing prints
    bool sparse_ingest = true;       // NVCA_SPARSE_INGEST=0: whole host frames in shrink-first mode
    bool pyr_off = false;            // NVCA_PYR_OFF: per-level launches for SCALE_IMAGE
    int  part_stats = 0;             // NVCA_PART_STATS[=n]: phase timers of part batches with n (default 8) or more streams; 0: off
    int  ingest_chunk = 8;           // NVCA_INGEST_CHUNK=n: chunk size of host-frame batches, 0 = no chunking
    int  deep_stage = 0;             // NVCA_DEEP_STAGE=s: first stage of k_deep (0: the plan's default)
    bool tiles = true;               // NVCA_TILES=0: row-strip kernel
    bool plan_debug = false;         // NVCA_PLAN_DEBUG: per-scale tile sizes on stderr
    bool deep_lds = true;            // NVCA_DEEP_LDS_OFF: k_deep without LDS patches
    bool trk_fold = true;            // NVCA_TRK_FOLD=0: NuboTracker's components through the per-pixel kernels (k_ccl_flatten / _reduce / _collect) instead of the per-tile reduction + fold of tile roots
    int  trk_order = -1;             // NVCA_TRK_ORDER: visiting order of k_ccl_reduce (-1: decided per frame on the device)
    int  host_threads = -1;          // NVCA_HOST_THREADS=n: helper threads for per-job host work (-1: min(8, cores / 2) - 1; 0: none)
    bool two_lanes = true;           // NVCA_TWO_LANES=0: both submitted face batches on the context's stream (one after the other)
    bool fb_dense = true;            // NVCA_FB_DENSE=0: a FIND_BIGGEST search on the small-image path re-scans its narrowed grids in a second launch instead of replaying them on the host from the first launch's dense candidates + stage-0 reject bits
    bool roi = true;                 // NVCA_ROI=0: small images take the large-image path too (plan + four launches per job)
    bool stage_order = false;        // NVCA_STAGE_ORDER=1 (0, the default: the cascade's own order on every tile -- the cheapest one for a cascade whose stages each reject about half, as trained ones do; a round then may take two stages at once): the tile kernels walk the early stages 1 .. 5 in the cascade's order on every tile (1: in the order the previous tile of the band found cheapest -- cost per window killed; the set of survivors is the same.  A cascade whose stages each reject about half, as trained ones do, keeps its own order either way)
    int  pair_max = 32;              // NVCA_PAIR_MAX=n (<= 32): windows up to which a round of the tile kernels runs lane = (window, stump) instead of a window per lane
    int  spec_pairs = 1536;          // NVCA_SPEC_PAIRS=n: with at most 32 windows left a round takes as many stages as stay within n (window, stump) pairs (768 = one step of the workgroup)
    int  pre_cus = 0;                // NVCA_PRE_CUS=n: a submitted face batch's pre-processing runs on a stream confined to n CUs (hipExtStreamCreateWithCUMask), beside the other batch's band kernel (0: behind it, on the lane's own stream)
    bool quiet = false;              // NVCA_QUIET: no one-time notes on stderr (a plan that falls back to the row-strip kernel)
    const char *stamps_out = nullptr;   // NVCA_STAMPS_OUT (diagnostic build only)
};
const Switches &switches();

// --------------------------------------------------------------------------
// Context
// --------------------------------------------------------------------------
#define NVCA_HIP_CHECK(ctx, expr)                                                   \
    do {                                                                            \
        hipError_t e__ = (expr);                                                    \
        if (e__ != hipSuccess) {                                                    \
            (ctx)->set_error(std::string(#expr) + ": " + hipGetErrorString(e__));   \
            return NVCA_ERR_HIP;                                                    \
        }                                                                           \
    } while (0)

struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    int ensure(size_t n);          // grows (never shrinks); returns hipError as int
    void release();
    template <class T> T *as() const { return (T *)p; }
};
struct PinnedBuf {
    void *p = nullptr; size_t bytes = 0;
    int ensure(size_t n);
    void release();
    template <class T> T *as() const { return (T *)p; }
};

// Page-locked memory of the context's own that caller host memory crosses through when it is not page-locked by the caller
// (nvca_host_register): a ring of fixed slots, each with the event of the last copy that read or wrote it (api.cpp, caller_h2d ...).
struct BounceRing {
    static constexpr size_t kSlot = 8u << 20;      // a 1080p BGR frame (6.2 MB) is one slot: one CPU copy (shared by the helper threads) + one DMA
    static constexpr int kSlots = 12;
    PinnedBuf buf; hipEvent_t ev[kSlots] = {}; bool pending[kSlots] = {}; int next = 0;
    void release() { for (hipEvent_t &e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; } buf.release(); }
};

// tracker workspace (tracker.cpp): slot table, labels, per-root accumulators, component list, frame staging
struct TrkWorkspace {
    DevBuf slots, labels, acc, out, staging, flags, roots, tiles; PinnedBuf h_slots, h_out;
    // the live-tile list's marks are stamped with the launch's tick and never cleared (kernels_tracker.hip, TileList): they are
    // zeroed when the buffer's layout changes (frame size, batch) and when the tick would come round
    int tick = 0, tiles_w = 0, tiles_h = 0, tiles_batch = 0;
    void release_all() { slots.release(); labels.release(); acc.release(); out.release(); staging.release(); flags.release(); roots.release(); tiles.release(); h_slots.release(); h_out.release(); tiles_w = tiles_h = tiles_batch = 0; }
};

// batched part detectors (parts.cpp): working images of a call carved from one arena, the small tables its launches read
struct PartWorkspace {
    DevBuf arena, tables, hist, luts; PinnedBuf h_tables;
    size_t tab_used = 0;
    hipEvent_t images_done = nullptr;
    void release_all() { arena.release(); tables.release(); hist.release(); luts.release(); h_tables.release(); if (images_done) { (void)hipEventDestroy(images_done); images_done = nullptr; } }
};

// A few helper threads for host work that is independent per job (the candidate lists of a round's face-region searches are
// converted, replayed and grouped job by job: 96 jobs of ~50 us on the calling thread were most of a loaded part batch).
// The caller takes part; run() returns when every index has been handled.  Created on first use, joined with the context.
struct WorkPool;
WorkPool *work_pool_create(int threads);
void work_pool_destroy(WorkPool *p);
void work_pool_run(WorkPool *p, int n, void (*fn)(void *arg, int i), void *arg);     // p == nullptr: serial
int work_pool_threads(const WorkPool *p);          // helper threads (0 for nullptr)

struct DetectPlan;   // plan.cpp
struct ScaleTable;   // plan.cpp: one cascade at one scale factor (geometry-independent stump records), cached in the context
struct FaceTicket;   // api.cpp
void free_face_ticket(FaceTicket *t);
struct GeomPlan;     // api.cpp
struct Workspace;    // api.cpp

struct KernelTimer {
    bool on = false;
    int stride = 1;                 // events ride on every stride-th batch of an entry point (they serialise consecutive launches)
    uint64_t seq[2] = {0, 0};       // batches seen: [0] face detector, [1] tracker
    bool sample = true;             // the batch being queued carries events
    void tick(int which) { sample = stride <= 1 || (seq[which]++ % (uint64_t)stride) == 0; }
    struct Ev { hipEvent_t a, b; int k; bool first; };
    std::vector<Ev> pending;
    std::vector<hipEvent_t> pool;
    double total_ms[NVCA_K_COUNT] = {0};
    int64_t launches[NVCA_K_COUNT] = {0};
};

} // namespace nvca

struct nvca_cascade { nvca::Cascade c; nvca_ctx *ctx; };

namespace nvca {
static constexpr int kLanes = 10;         // lane 0: the context's stream; 1 .. 7: the batched part detectors; 8: the face detector's second batch in flight; 9: the trackers
static constexpr int kPartLanes = 8;      // lanes [0, kPartLanes) are the ones the part detectors spread over
static constexpr int kFaceLane2 = 8;
static constexpr int kTrackerLane = 9;    // NuboTracker's kernels: its state and workspace are its own, so a face batch in flight (lane 0 / 8) and a tracker call overlap
}

struct nvca_ctx {
    int device = 0;
    hipStream_t stream = nullptr;             // lane 0
    hipStream_t lane_streams[nvca::kLanes] = {nullptr};   // [0] == stream; the others carry the batched part detectors' jobs (api.cpp, Lane)
    int cur_lane = 0;
    hipStream_t cs() const { return lane_streams[cur_lane]; }       // the stream of the lane that is being queued on
    hipStream_t pre_streams[2] = {nullptr, nullptr};   // CU-masked streams for the pre-processing of the two face batches in flight ("pre_cus"), created on first use
    int pre_streams_cus = 0;                  // the CU count they were created for
    hipEvent_t pre_done[2] = {nullptr, nullptr};
    nvca::HostRangeTable host_ranges;         // what the caller page-locked through nvca_host_register (host_ranges.h: the only caller memory a copy is handed as it stands)
    nvca::BounceRing bounce;                  // everything else crosses through here
    hipStream_t copy_stream = nullptr;        // H2D of the next chunk of host frames while the current one computes
    std::vector<hipEvent_t> chunk_events;
    nvca::FaceTicket *face_tickets[3] = {nullptr, nullptr, nullptr};   // [0] synchronous calls, [1] / [2] submit / collect
    uint64_t face_serial = 0;
    int ptr_ring_used = 0;                    // nvca_bgr2gray: entries of the frame-pointer ring handed out since the last drain
    int defer_device_sync = 0;                // > 0: primitives that write device memory return without draining the stream
                                              // (internal callers chaining primitives on the context's stream, parts.cpp)
    std::string err;
    int hit_cap = 16384;
    int hit_cap_wanted = 0;                   // > hit_cap: a launch set produced more raw candidates than its lists hold; the per-frame size that holds it
    int policy = NVCA_SUM_F32PAIR;
    uint64_t next_uid = 1;
    nvca::KernelTimer timer;
    std::map<std::string, std::unique_ptr<nvca::GeomPlan>> plans;
    std::map<std::pair<uint64_t, uint64_t>, nvca::ScaleTable *> scale_tables;   // (cascade uid, factor bits)
    std::unique_ptr<nvca::Workspace> ws;
    nvca::TrkWorkspace trk;           // tracker buffers live and die with the context
    // the working images / tables of a batched part-detector call; two sets: a submitted call (nvca_part_batch_submit) may be in
    // flight while the one before it is collected -- a call uses the set of its ticket's parity (parts.cpp sets part_set)
    nvca::PartWorkspace part_sets[2]; int part_set = 0;
    nvca::PartWorkspace &pw() { return part_sets[part_set]; }
    void *part_calls[2] = {nullptr, nullptr};  // submitted, not yet collected part-detector calls (parts.cpp: PartCall), by ticket parity
    void (*part_calls_abandon)(nvca_ctx *) = nullptr;      // gives up whatever is outstanding (newest first: rolled back, drained, deleted)
    int part_seq = 0;                         // the next ticket
    nvca::Switches sw;                // this context's switches: the process defaults (environment), nvca_ctx_set_option overrides
    int lds_grant[2] = {0, 0};        // dynamic LDS already granted to k_tile / k_band through this context (hipFuncSetAttribute)
    void *identity_lut = nullptr;     // 256 B on device
    nvca::DevBuf overlay_img;         // the caller's overlay image on the device (nvca_overlay_blend on device frames)
    // small-image detector (kernels_roi.hip): per-cascade stage records on the device, the tables / candidate list of a launch
    std::map<uint64_t, nvca::DevBuf *> roi_stage_recs;
    // (three sets: [0] the synchronous callers', [1] / [2] the part-detector calls in flight by ticket parity -- a round of theirs stays
    // queued between submit and collect)
    struct RoiBuffers { nvca::DevBuf tables, hits, rej; nvca::PinnedBuf h_tables, h_hits, h_rej; } roi_bufs[3]; int roi_set = 0;
    RoiBuffers &rbuf() { return roi_bufs[roi_set]; }
    size_t roi_first_hint = 0;              // candidates of the recent small-image rounds (+ a quarter): what the launch copies back with itself
    nvca::WorkPool *pool = nullptr; bool pool_tried = false;
    std::mutex err_mu;                // set_error may be called from the helper threads
#ifdef NVCA_STAMPS
    unsigned long long *stamps = nullptr;
#endif
    std::recursive_mutex mu;          // serialises entry points: elements on different streaming threads share one context
    void set_error(const std::string &s) { std::lock_guard<std::mutex> lk(err_mu); err = s; }
    nvca_ctx();
    ~nvca_ctx();
};

// Exception barrier of the ABI.  Every extern "C" entry point is a function-try-block whose handler is one of these macros:
// nothing thrown below it (std::bad_alloc / std::length_error from the host-side containers, anything else) crosses into the
// caller's C frames -- the call returns a status code instead, as include/nubovca.h promises.  api_catch() rethrows inside its
// own try block to tell the cases apart (it never throws itself).
namespace nvca { int api_catch(nvca_ctx *ctx) noexcept; hipError_t take_launch_error(const char **kernel); }
#define NVCA_API_CATCH(ctxexpr) catch (...) { return nvca::api_catch(ctxexpr); }
#define NVCA_API_CATCH_VOID catch (...) { (void)nvca::api_catch(nullptr); }

#define NVCA_LOCK_OR_FAIL(ctx) if (!(ctx)) return NVCA_ERR_ARG; std::lock_guard<std::recursive_mutex> nvca_lock__((ctx)->mu); (void)nvca::take_launch_error(nullptr)   /* a launch failure of an earlier call on this thread has been reported by that call */

namespace nvca {

// RAII bracket: while timing is enabled, every kernel launched inside the scope carries a start / stop event pair in
// its dispatch packet (hipExtLaunchKernelGGL) -- no separate event-record packets on the stream
struct TimedLaunch {
    nvca_ctx *ctx; int k; int n = 0; TimedLaunch *prev = nullptr; bool active = false;
    TimedLaunch(nvca_ctx *c, int kind);
    ~TimedLaunch();
};
bool launch_events(hipEvent_t *a, hipEvent_t *b);      // event pair for the next launch of the current scope, if any
// A refused launch (bad configuration, LDS grant missing) is not sticky: the status is read right behind the launch and the
// FIRST failure of the calling thread is kept, with the kernel's name, until the entry point's next NVCA_LAUNCH_CHECK turns it
// into NVCA_ERR_HIP -- a kernel that did not run never hands stale buffers to the host logic as if they were results.
void note_launch(const char *kernel);                  // reads hipGetLastError()
hipError_t take_launch_error(const char **kernel);     // returns and clears the thread's first recorded failure
#define NVCA_LAUNCH(kern, grid, block, shmem, st, ...)                                                            \
    do {                                                                                                          \
        hipEvent_t ea__, eb__;                                                                                    \
        if (nvca::launch_events(&ea__, &eb__)) hipExtLaunchKernelGGL(kern, grid, block, shmem, st, ea__, eb__, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__);                                       \
        nvca::note_launch(#kern);                                                                                 \
    } while (0)
#define NVCA_LAUNCH_CHECK(ctx)                                                                                    \
    do {                                                                                                          \
        const char *k__ = nullptr;                                                                                \
        const hipError_t le__ = nvca::take_launch_error(&k__);                                                    \
        if (le__ != hipSuccess) {                                                                                 \
            (ctx)->set_error(std::string("kernel launch failed (") + (k__ ? k__ : "?") + "): " + hipGetErrorString(le__)); \
            return NVCA_ERR_HIP;                                                                                  \
        }                                                                                                         \
    } while (0)

// --------------------------------------------------------------------------
// Kernel launch wrappers (kernels_pre.hip / kernels_cascade.hip)
// --------------------------------------------------------------------------
struct PreGeom {
    int sw, sh, sstride, cn;      // source frame
    int w, h, gpitch;             // working gray image (pitch in bytes)
    int spitch;                   // integral pitch (elements), rows = h+1
    int nbands;
    size_t src_slot, gray_slot, sum_slot, band_slot;   // strides between batch slots (elements of each plane)
};

// src[b] pointers are passed as a device array of pointers (frames need not be contiguous)
void launch_gray(hipStream_t st, const uint8_t *const *d_src, const PreGeom &g, int mode,
                 const int *d_xofs, const short *d_ialpha, const int *d_yofs, const short *d_ibeta, int xmax,
                 uint8_t *gray, unsigned *hist, int batch, bool aligned4);
// one pyramid level of a CV_HAAR_SCALE_IMAGE scan (device copy): all levels are resized / integrated by one launch each
struct PyrLevelDev {
    int szw, szh, gpitch, mode, xmax, plane_off, pad0, pad1;
    long long gray_off;
    const int *xofs; const short *ialpha; const int *yofs; const short *ibeta;
};
void launch_pyr_resize(hipStream_t st, const uint8_t *src, int sw, int sh, int sstride, size_t src_slot, const PyrLevelDev *levels,
                       int nlev, int nimg, int maxw, int maxh, uint8_t *aux, size_t aux_slot);
void launch_pyr_integral(hipStream_t st, const uint8_t *aux, size_t aux_slot, const PyrLevelDev *levels, int nlev, int nimg,
                         int *sum, unsigned *sq32, size_t sum_slot, int P);
void launch_resize1(hipStream_t st, const uint8_t *src, int sw, int sh, int sstride, int mode,
                    const int *d_xofs, const short *d_ialpha, const int *d_yofs, const short *d_ibeta,
                    int xmax, uint8_t *dst, int dw, int dh, int dstride, unsigned *hist, int batch = 1, size_t src_slot = 0, size_t dst_slot = 0);
void launch_resize3(hipStream_t st, const uint8_t *src, int sw, int sh, int sstride, int mode,
                    const int *d_xofs, const short *d_ialpha, const int *d_yofs, const short *d_ibeta,
                    int xmax, uint8_t *dst, int dw, int dh, int dstride);
void launch_flip_h(hipStream_t st, const uint8_t *src, int w, int h, int spitch, uint8_t *dst, int dpitch, int batch = 1, size_t src_slot = 0,
                   size_t dst_slot = 0);
void launch_hist(hipStream_t st, const uint8_t *gray, int w, int h, int pitch, unsigned *hist);
void launch_lut(hipStream_t st, unsigned *hist, int total, uint8_t *lut, int batch, int rezero = 0,
                unsigned long long *zero_a = nullptr, unsigned long long *zero_b = nullptr);
void launch_apply_lut(hipStream_t st, const uint8_t *src, int w, int h, int spitch, const uint8_t *lut,
                      uint8_t *dst, int dpitch, int batch = 1, size_t src_slot = 0, size_t dst_slot = 0);
void launch_work_resize(hipStream_t st, bool bgr, const uint8_t *const *d_srcs, const int *d_lut_idx, const uint8_t *d_luts, int sh, int sstride,
                        int mode, const int *d_xofs, const short *d_ialpha, const int *d_yofs, const short *d_ibeta, int xmax,
                        uint8_t *dst, int dw, int dh, int dstride, size_t dst_slot, unsigned *hist, int batch);
void launch_colsum(hipStream_t st, const uint8_t *gray, const uint8_t *lut, int lut_stride, const PreGeom &g,
                   unsigned *bandsum, unsigned *bandsq, int batch);
void launch_bandscan(hipStream_t st, const PreGeom &g, unsigned *bandsum, unsigned *bandsq, int batch);
void launch_integral(hipStream_t st, const uint8_t *gray, const uint8_t *lut, int lut_stride, const PreGeom &g,
                     const unsigned *bandsum, const unsigned *bandsq, int *sum, unsigned long long *sqsum,
                     int batch);

// integral pair of small images (rows x cols fit 64 KiB of LDS) in one launch, one workgroup per image
bool small_integral_fits(const PreGeom &g);
void launch_small_integral(hipStream_t st, const uint8_t *gray, const uint8_t *lut, int lut_stride, const PreGeom &g, int *sum,
                           unsigned long long *sqsum, int batch);

// ---- tracker (kernels_tracker.hip) ----
struct TrkSlot {                // per tracker in the batch
    const uint8_t *src;         // BGRA frame
    uint8_t *prev;              // previous gray  [h][w]
    float *mhi;                 // motion history [h][w]
    float ts, delbound;         // (float)timestamp, (float)(timestamp - duration)
    float seg;                  // (float)seg_thresh
    int threshold;
    int has_prev;               // num_frames > 0
    int sstride;
    int min_area;               // __join_objects drops boxes outside (min_area, max_area) before anything else:
    long long max_area;         // k_ccl_collect does not even report them
};
struct CompAcc { int minx, miny, maxx, maxy, seed, pad; };   // per root, stored at the root's pixel index
// out: [0] = component count, [1] unused, then 6 ints per component: slot, first seed index, x, y, w, h
// flags: one byte per 256-pixel row segment and slot, set by the pixel pass where the motion history holds anything -- the
// component kernels leave the other segments alone (a static scene with a few moving objects is mostly such segments)
// roots: [0] = tile roots listed, [1] = the list overflowed, then one entry (slot * w * h + pixel) per tile root; mode: 0 folded component
// path, 1 per-pixel component kernels, 2 the latter without the pixel pass (fallback behind an overflow of mode 0's list)
void launch_tracker(hipStream_t st, const void *d_slots, int batch, int w, int h, bool vec4, int *labels, void *acc,
                    int *out, int cap, bool run_ccl, uint8_t *flags, int order /* Switches::trk_order */, int *roots, int roots_cap, int mode, int *tiles, int tick);
inline size_t tracker_count_offset(int w, int h, int batch) { return ((size_t)((w + 255) / 256) * h * batch + 63) & ~(size_t)63; }
inline size_t tracker_flag_bytes(int w, int h, int batch) { return tracker_count_offset(w, h, batch) + sizeof(int) * (size_t)batch; }   // flag bytes, then a live-segment count per slot

struct CascadeArgs {
    const int *sum; const unsigned long long *sqsum;
    size_t sum_slot;               // elements between slots
    int spitch;
    const ScaleRec *scales; const StageRec *stages;
    const StripRec *strips; const int *pos;
    const int *order; int blocks_per_frame;   // k_strip dispatch slot -> strip
    const TileRec *tiles; const int *tile_order; int tile_blocks_per_frame;   // k_tile
    const unsigned short *tcoords; int tile_lds;
    const BandRec *bands; const int *band_order; int band_blocks_per_frame; int batch; int band_map;  // k_band
    const DeepRec *deeprecs;                  // [nscales] or null (k_deep: LDS patches)
    int nscales;
    const unsigned *tasks; int ntasks;        // k_stage0 wave tasks: scale << 20 | iy << 7 | word
    unsigned long long *failbits;             // [batch][ntasks] stage-0 reject bits
    double *vnf;                              // [batch][ntasks*64] variance normaliser per window
    int nstages; int pair_policy;  // 1 = F32PAIR
    int stage_order;               // Switches::stage_order
    int *stage_hint;               // [8] per plan: the stat words (order | entered << 16 | passed per stage) the last tile that finished left -- where a band's first tile and the per-tile kernel start from (an intentionally racy hint: plain stores, validated before use)
    int spec_pairs;                // Switches::spec_pairs
    int pair_max;                  // Switches::pair_max (<= kPairMax)
    const float *stage_thr;        // [nstages + 8]: StageRec::thr of every stage (tile_stages reads eight at once)
    const int *stage_first;        // [nstages + 1 + 8]: first stump of every stage, the stump count, then INT_MAX padding (tile_stages reads eight entries at once)
    int deep_stage;                // first stage evaluated by k_deep (== nstages: the tile kernels walk the whole cascade, k_deep is not launched)
    int deep_lds;                  // bytes of k_deep's largest window patch (dynamic LDS)
    unsigned long long *deep;      // deep[0] = count, then (slot << 32) | key
    unsigned deep_cap;
    int key_sy, key_ss;            // a candidate's key = scale << key_ss | iy << key_sy | ix: the plan sizes the three fields for its own grids (DetectPlan::key_sy / key_ss), so a ladder of hundreds of scales (multi-scale-factor 1 .. 4) fits next to small grids and a 4K grid next to 25 scales
    unsigned long long *hits;      // hits[0] = running count, hits[1..cap] = (slot << 32) | key
    unsigned hit_cap;
    // general cascades (kernels_cascade.hip)
    const int *tilted;             // tilted integral planes, laid out like sum (null: the cascade has no tilted feature)
    const float *galpha;           // leaf values of every weak classifier, concatenated
    const int *gcls_first;         // first node of weak classifier c
    int stump_based;
#ifdef NVCA_STAMPS
    unsigned long long *dbg;       // diagnostic build only: per-phase s_memtime stamps of the first workgroups (scripts/stamps.py)
#endif
};
// which: 0 = k_stage0, 1 = k_strip, 2 = k_deep, 3 = k_tile, 5 = k_band
// lds_grant: the calling context's record of the dynamic LDS already granted to k_tile ([0]) / k_band ([1]); returns a
// hipError_t (as int) when the grant is refused, 0 otherwise
int launch_cascade_sc(hipStream_t st, const CascadeArgs &a, int batch, int which, int *lds_grant);
// detectMultiScale calls in halves (api.cpp): many calls share one wait per round
// ---- view-* outlines (nvca_draw_shapes): one coverage rule for the host rasteriser and the kernel
#if defined(__HIPCC__)
#define NVCA_HD __host__ __device__
#else
#define NVCA_HD
#endif
NVCA_HD inline bool shape_covers(const nvca_shape &sh, int px, int py)
{
    if (sh.kind == NVCA_SHAPE_RING4) {
        if (sh.w < 0) return false;
        const long long ro = sh.w + 2, ri = sh.w - 2 > 0 ? sh.w - 2 : 0, dx = px - sh.x, dy = py - sh.y, d2 = dx * dx + dy * dy;
        return d2 <= ro * ro && d2 >= ri * ri;
    }
    int x0 = sh.x, y0 = sh.y, x1 = sh.x + sh.w, y1 = sh.y + sh.h;
    if (x0 > x1) { const int t = x0; x0 = x1; x1 = t; }
    if (y0 > y1) { const int t = y0; y0 = y1; y1 = t; }
    const int ax0 = px > x0 ? px - x0 : x0 - px, ax1 = px > x1 ? px - x1 : x1 - px;
    const int ay0 = py > y0 ? py - y0 : y0 - py, ay1 = py > y1 ? py - y1 : y1 - py;
    if (px >= x0 && px <= x1 && (ay0 <= 1 || ay1 <= 1)) return true;          // the two horizontal edges, 3 rows each
    if (py >= y0 && py <= y1 && (ax0 <= 1 || ax1 <= 1)) return true;          // the two vertical edges, 3 columns each
    const int mx = ax0 < ax1 ? ax0 : ax1, my = ay0 < ay1 ? ay0 : ay1;         // round joins: the 4-neighbourhood of a vertex
    return mx + my == 1;
}
void draw_shapes_host(uint8_t *data, int w, int h, int stride, int channels, const nvca_shape *shapes, int n);

// ---- image-to-overlay (nvca_overlay_blend): kms_face_detect_display_detections_overlay_img, FACE/kmsfacedetect.cpp:427-502.
// One arithmetic for the host loop and the kernel (as for the outlines above).
// Channel k of output pixel (x, y) of cvResize(costume, costumeAux, CV_INTER_LINEAR) on an 8-bit image with cn interleaved
// channels: cv::resize's fixed-point bilinear path (11-bit coefficients, tables from build_resize_tab), its 2 x 2 area
// shortcut, or the identity.
NVCA_HD inline int resize_sample_cn(const uint8_t *src, int sh, int sstride, int cn, int mode, const int *xofs, const short *ialpha,
                                    const int *yofs, const short *ibeta, int xmax, int x, int y, int k)
{
    if (mode == 0) return src[(size_t)y * sstride + (size_t)x * cn + k];
    if (mode == 2) {
        const uint8_t *s0 = src + (size_t)(2 * y) * sstride + (size_t)(2 * x) * cn + k, *s1 = s0 + sstride;
        return (s0[0] + s0[cn] + s1[0] + s1[cn] + 2) >> 2;
    }
    int sy0 = yofs[y], sy1 = sy0 + 1;
    sy0 = sy0 >= 0 ? (sy0 < sh ? sy0 : sh - 1) : 0;
    sy1 = sy1 >= 0 ? (sy1 < sh ? sy1 : sh - 1) : 0;
    const uint8_t *s0 = src + (size_t)sy0 * sstride + (size_t)xofs[x] * cn + k, *s1 = src + (size_t)sy1 * sstride + (size_t)xofs[x] * cn + k;
    const bool inner = x < xmax;
    const int a0 = inner ? ialpha[2 * x] : 2048, a1 = inner ? ialpha[2 * x + 1] : 0;
    const int h0 = s0[0] * a0 + (inner ? s0[cn] * a1 : 0), h1 = s1[0] * a0 + (inner ? s1[cn] * a1 : 0);
    return (((ibeta[2 * y] * (h0 >> 4)) >> 16) + ((ibeta[2 * y + 1] * (h1 >> 4)) >> 16) + 2) >> 2;
}
// the write of one overlay pixel v[0 .. cn) onto a BGR pixel of the frame (:467-490; SRC_OVERLAY is 1)
NVCA_HD inline void overlay_pixel(uint8_t *px, const int *v, int cn)
{
    if (cn == 1) { px[0] = px[1] = px[2] = (uint8_t)v[0]; return; }
    if (cn == 3) { px[0] = (uint8_t)v[0]; px[1] = (uint8_t)v[1]; px[2] = (uint8_t)v[2]; return; }
    const double proportion = (double)v[3] / (double)255;
    const double overlay = 1.0 * proportion, original = 1 - overlay;
    for (int k = 0; k < 3; k++) px[k] = (uint8_t)((v[k] * overlay) + (px[k] * original));
}
// where the reference puts the scaled image for a box, and how large (:441-444: the sums are truncated, not the products)
struct OverlayPlace { int x, y, w, h; };
inline OverlayPlace overlay_place(const nvca_rect &b, const nvca_overlay &ov)
{
    OverlayPlace p;
    p.x = (int)(b.x + (b.w * ov.offset_x_percent));
    p.y = (int)(b.y + (b.h * ov.offset_y_percent));
    p.h = (int)(b.h * ov.height_percent);
    p.w = (int)(b.w * ov.width_percent);
    return p;
}
void overlay_blend_host(uint8_t *frame, int W, int H, int stride, const nvca_rect *boxes, int n, const nvca_overlay &ov);
void launch_overlay(hipStream_t st, uint8_t *frame, int W, int H, int stride, const OverlayPlace &p, const uint8_t *img, int ih, int istride, int cn,
                    int mode, const int *xofs, const short *ialpha, const int *yofs, const short *ibeta, int xmax);
void launch_draw_shapes(hipStream_t st, uint8_t *data, int w, int h, int stride, int channels, const nvca_shape *d_shapes, int n,
                        int bx0, int by0, int bx1, int by1);

// ---- caller host memory <-> device (api.cpp): direct only inside a range the caller registered, otherwise through the bounce ring
int caller_h2d(nvca_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t st);
int caller_h2d_rows(nvca_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows, hipStream_t st);
int caller_d2h_rows(nvca_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows, hipStream_t st);   // returns with dst filled (the stream is drained)

// ---- working images of a batched part call (api.cpp), all on the current lane
// N images of one launch set: image k = [equalizeHist](resize(source k)) at dst + k * slot, pitch dw.  BGR sources: gray of the frame
// computed on the fly (cvtColor then resize); gray sources go through LUT lut_idx[k] of `luts` first when lut_idx is given
struct PartImageBatch {
    bool bgr = true, post_eq = true;
    int sw = 0, sh = 0, sstride = 0, dw = 0, dh = 0;
    std::vector<const void *> src; std::vector<int> lut_idx;
    uint8_t *dst = nullptr; size_t slot = 0;
};
int part_table(nvca_ctx *ctx, const void *host, size_t bytes, void **dev);   // a small table for the next launch on the current lane (upload ring)
int part_arena(nvca_ctx *ctx, size_t bytes, uint8_t **base);                 // grows the arena (before anything of the call is queued)
int part_luts(nvca_ctx *ctx, int n_keep, int n_scratch, uint8_t **keep);     // LUT storage: n_keep that live through the call + scratch
int part_gray_eq(nvca_ctx *ctx, const void *const *bgr, int n, int w, int h, int stride, uint8_t *gray, size_t slot, uint8_t *luts);   // gray images + their equalisation LUTs
int part_image_batch(nvca_ctx *ctx, const PartImageBatch &b, const uint8_t *luts);
int part_flip_batch(nvca_ctx *ctx, const uint8_t *src, uint8_t *dst, int w, int h, int n, size_t slot);
int part_images_done(nvca_ctx *ctx, const int *lanes, int n);                // the lanes in `lanes` wait for what the current lane has queued so far
struct DetectJob;
int make_detect_job(nvca_ctx *ctx, DetectJob &j, const nvca_cascade *casc, const void *gray, int w, int h, int stride, int mem,
                    double sf, int min_neighbors, int flags, int minw, int minh, int maxw, int maxh, bool raw_only);
int run_detect_jobs(nvca_ctx *ctx, DetectJob *const *jobs, int n, const int *lanes);      // lanes: per job, or null (current lane)
// ... in two halves: the first round queued and left in flight, then the rest (api.cpp)
struct JobRound;
JobRound *job_round_new();
void job_round_free(JobRound *r);
int detect_jobs_begin(nvca_ctx *ctx, DetectJob *const *jobs, int n, const int *lanes, JobRound *R, bool *queued);
int detect_jobs_finish(nvca_ctx *ctx, DetectJob *const *jobs, int n, const int *lanes, JobRound *R, bool queued);
DetectJob *detect_job_new();
void detect_job_free(DetectJob *j);
const std::vector<nvca_rect> &detect_job_out(const DetectJob *j, int k);
constexpr int kJobImages = 32;                                    // images of one geometry that a plain / SCALE_IMAGE job can carry
int detect_job_add_image(DetectJob *j, const void *image);       // one more image for the job's launch set; returns its index k (detect_job_out), -1: full / not possible
// detectMultiScale(CV_HAAR_SCALE_IMAGE) on two images of one geometry with shared launches (api.cpp; used by parts.cpp)
int detect_scale_image_pair(nvca_ctx *ctx, const nvca_cascade *casc, const void *img_a, const void *img_b, int w, int h, int stride,
                            int mem, double sf, int min_neighbors, int minw, int minh, std::vector<nvca_rect> *outs /* [2] */);
// general cascades: which = 0: variance + stage 0 for every window (reject bits + normaliser), 1: the remaining stages on the
// visited stage-0 survivors, window per lane
void launch_generic(hipStream_t st, const CascadeArgs &a, int batch, int which);
// tilted integral (cv::integral's third plane) of `batch` images / of every pyramid level: one workgroup per image
void launch_tilted(hipStream_t st, const uint8_t *gray, const uint8_t *lut, int lut_stride, const PreGeom &g, int *tilted, int batch);
void launch_pyr_tilted(hipStream_t st, const uint8_t *aux, size_t aux_slot, const PyrLevelDev *levels, int nlev, int nimg,
                       int *tilted, size_t sum_slot, int P, int maxw, int maxh);
// ---- detectMultiScale on a small image in one workgroup (kernels_roi.hip)
struct RoiStep {                  // one ladder step (scale-cascade scan) or one pyramid level (CV_HAAR_SCALE_IMAGE) of a job
    const TStumpRec *trecs;       // the cascade's stumps at this step's factor (levels: factor 1)
    int ex, ey, ew, eh;           // variance rectangle (window-relative)
    int startX, endX, startY, endY;   // scale-cascade: grid indices, window origin = cvRound(i * ystep); levels: origins 0 .. end, every `step` pixels
    int step, adaptive, job, key_step;    // job: index of the step's job in the launch; key_step: the step's number inside its job (a candidate's key carries it; a step with many rows goes out as several records -- one workgroup each -- with the same number)
    int key_x0, key_dx, key_y0, key_dy;   // a candidate's key: column key_x0 + gx * key_dx, row key_y0 + gy * key_dy (grid indices, or level origins)
    int szw, szh;                 // level size
    int mode, xmax, xofs_off, yofs_off, ialpha_off, ibeta_off;      // the level's cv::resize tables (byte offsets into the launch's table blob)
    double inv_area, ystep;
    // adaptive == 2 ("dense", FIND_BIGGEST searches): every window that passes stage 0 goes on, visited by the serial walk or not, and the
    // stage-0 reject bits of the step's grid are written out -- row r's chunk c at rej[rej_off + r * rej_wpr + c] (64 windows a word) -- so
    // that the host can replay the walk from ANY start column: a narrowed re-scan needs no second launch (api.cpp, fb_replay)
    int rej_off, rej_wpr;
};
struct RoiJobDev {
    const uint8_t *img; int w, h, stride;
    int first_step, nsteps, scale_image;
    const StageRec *stages; int nstages, pair_policy;
    int slot, pad;
};
static constexpr int kRoiMaxWin = 2048;           // windows of one ladder step / pyramid level of a small-image job
void launch_roi(hipStream_t st, const RoiJobDev *jobs, int nsteps, const RoiStep *steps, const unsigned char *tabs, unsigned long long *hits,
                unsigned hit_cap, int plane_words, int lds_bytes, unsigned long long *rej);
#ifdef NVCA_STAMPS
void roi_stamps_dump(const char *path);     // diagnostic build: k_roi's phase sums as text
#endif
int roi_grant_lds(int bytes);     // dynamic LDS above 64 KiB is granted per function and device (monotonic, process-wide); returns a hipError_t as int

// groupRectangles per frame on the device; out: [batch][2 + 4*out_cap] ints: count (-1 = host must group), raw count, boxes
void launch_group(hipStream_t st, const CascadeArgs &a, const int *group_thr, int *out, int out_cap, int batch);

} // namespace nvca
