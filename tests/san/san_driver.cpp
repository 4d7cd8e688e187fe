// san_driver.cpp -- test infrastructure (never shipped): runs the product's pure-host sources -- cascade_xml.cpp (the loader),
// plan.cpp (table builders) and host_logic.cpp (groupRectangles, track_faces, __join_objects, the part detectors' merging
// heuristics) -- under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU.  tests/test_host_sanitizers.py builds it
// (clang++ -fsanitize=address,undefined, the three product sources compiled as they are) and checks what it prints against the
// oracle.  The GPU pool has no sanitizer support (ASan / XNACK are refused there), so this is where memory errors of the host
// code are looked for.
//
// The three sources reach the HIP runtime through a handful of calls (table uploads); the driver links no HIP library and
// supplies host doubles for exactly those calls: a "device" buffer is a malloc'd block.  Nothing here runs a kernel.
#include "../../nubomedia-vca_amd/csrc/plan.h"
#include "../../nubomedia-vca_amd/csrc/host_logic.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <set>

// ---- host doubles of what api.cpp / the HIP runtime provide to these sources ------------------------------------------
extern "C" hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { memcpy(dst, src, n); return hipSuccess; }
extern "C" hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
extern "C" hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
namespace nvca {
struct Workspace { int unused; };
struct GeomPlan { int unused; };
int DevBuf::ensure(size_t n) { if (n <= bytes) return 0; free(p); p = malloc(n); bytes = p ? n : 0; return p ? 0 : 1; }
void DevBuf::release() { if (p && bytes) free(p); p = nullptr; bytes = 0; }
DetectPlan::~DetectPlan() { release_tables(); d_blob.release(); }
}
nvca_ctx::nvca_ctx() {}
nvca_ctx::~nvca_ctx() { plans.clear(); nvca::free_scale_tables(this); }

using namespace nvca;

static unsigned long long g_rng = 0x9E3779B97F4A7C15ull;
static unsigned rnd() { g_rng = g_rng * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(g_rng >> 33); }
static int rnd_in(int lo, int hi) { return lo + (int)(rnd() % (unsigned)(hi - lo + 1)); }

static std::string slurp(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss; ss << f.rdbuf();
    return ss.str();
}

static int fail(const char *what) { fprintf(stderr, "san_driver: FAILED: %s\n", what); return 1; }

// ---- loader ------------------------------------------------------------------------------------------------------------
static int run_loader(int argc, char **argv)
{
    for (int i = 0; i < argc; i++) {
        const std::string xml = slurp(argv[i]);
        Cascade c; std::string err;
        const int rc = parse_cascade_xml(xml.data(), xml.size(), c, err);
        printf("{\"loader\": \"%s\", \"rc\": %d, \"ow\": %d, \"oh\": %d, \"stages\": %zu, \"cls\": %zu, \"nodes\": %zu, \"alpha\": %zu, \"tilted\": %d, \"stumps\": %d}\n",
               argv[i], rc, c.ow, c.oh, c.stages.size(), c.cls.size(), c.nodes.size(), c.alpha.size(), c.has_tilted ? 1 : 0, c.stump_based ? 1 : 0);
        // damaged copies: any status will do, memory safety is the point.  Byte flips, truncations, digit runs blown up.
        int statuses[3] = {0, 0, 0};
        for (int t = 0; t < 160; t++) {
            std::string b = xml;
            const int kind = t % 4;
            if (kind == 0) for (int k = rnd_in(1, 10); k > 0; k--) b[rnd() % b.size()] = (char)rnd();
            else if (kind == 1) b.resize(rnd() % b.size());
            else if (kind == 2) { const size_t at = rnd() % b.size(); b.insert(at, "99999999999"); }
            else { const size_t at = rnd() % b.size(), n = rnd() % 200; b.erase(at, std::min(n, b.size() - at)); }
            Cascade d; std::string e2;
            const int r = parse_cascade_xml(b.data(), b.size(), d, e2);
            statuses[r == NVCA_OK ? 0 : (r == NVCA_ERR_PARSE ? 1 : 2)]++;
            if (r == NVCA_OK) {          // what the loader accepts must be internally consistent: the builders index it blindly
                for (const HaarClassifier &hc : d.cls) {
                    if (hc.first_node < 0 || hc.first_node + hc.nnodes > (int)d.nodes.size() || hc.first_alpha + hc.nnodes + 1 > (int)d.alpha.size()) return fail("accepted cascade with out-of-range classifier");
                    for (int l = 0; l < hc.nnodes; l++) {
                        const HaarNode &n = d.nodes[hc.first_node + l];
                        if (n.left > 0 && (n.left <= l || n.left >= hc.nnodes)) return fail("accepted tree with a backward child");
                        if (n.right > 0 && (n.right <= l || n.right >= hc.nnodes)) return fail("accepted tree with a backward child");
                        if (n.left <= 0 && -n.left > hc.nnodes) return fail("leaf index out of range");
                        if (n.right <= 0 && -n.right > hc.nnodes) return fail("leaf index out of range");
                    }
                }
                for (const HaarStage &st : d.stages) if (st.first_cls < 0 || st.first_cls + st.ncls > (int)d.cls.size()) return fail("accepted cascade with out-of-range stage");
            }
        }
        printf("{\"loader_fuzz\": \"%s\", \"ok\": %d, \"parse\": %d, \"other\": %d}\n", argv[i], statuses[0], statuses[1], statuses[2]);
    }
    return 0;
}

// ---- plans -------------------------------------------------------------------------------------------------------------
// every sample a window of a tile can touch (its origin + any corner offset of the early stages / the variance rectangle)
// must be one of the tile's staged columns / rows, inside the plane: the tile kernels look corners up through maps that are
// only filled for staged coordinates, and read the planes at those coordinates
static int check_plan(const Cascade &c, const DetectPlan &dp, int cols, int rows, int pitch)
{
    long long windows = 0, tiled = 0;
    for (size_t s = 0; s < dp.scales.size(); s++) windows += (long long)dp.scales[s].endX * dp.scales[s].endY;
    for (const TileRec &t : dp.tiles) {
        tiled += (long long)t.nx * t.ny;
        const ScaleRec &sc = dp.scales[t.scale];
        if (t.nx < 1 || t.ny < 1 || t.nx > kTileWin || t.ny > kTileRows || t.ix0 + t.nx > sc.endX || t.iy0 + t.ny > sc.endY) return fail("tile outside its scale's grid");
        if (t.ncol < 1 || t.ncol > kTileMaxCols || t.nrow < 1 || t.nrow > kTileThreads) return fail("tile sample counts");
        if (tile_lds_bytes(t.ncol, t.nrow, t.span_x, t.span_y) > dp.tile_lds || dp.tile_lds > kTileLdsBudget) return fail("tile LDS size");
        if ((size_t)t.col_off + t.ncol > dp.tcoords.size() || (size_t)t.row_off + t.nrow > dp.tcoords.size()) return fail("tile coordinate lists out of range");
        std::set<int> cs(dp.tcoords.begin() + t.col_off, dp.tcoords.begin() + t.col_off + t.ncol), rs(dp.tcoords.begin() + t.row_off, dp.tcoords.begin() + t.row_off + t.nrow);
        if ((int)cs.size() != t.ncol || (int)rs.size() != t.nrow) return fail("duplicate staged coordinate");
        if (*cs.begin() < t.x0 || *cs.rbegin() - t.x0 + 1 > t.span_x || *rs.begin() < t.y0 || *rs.rbegin() - t.y0 + 1 > t.span_y) return fail("map extents");
        if (*cs.rbegin() >= pitch || *rs.rbegin() >= rows + 1 + 4) return fail("staged coordinate outside the plane");
        const ScaleTable &tab = *dp.tabs[t.scale];
        const int early_last = std::min<int>(dp.deep_stage, (int)dp.stages.size());
        const int k1 = dp.stages[early_last - 1].first + dp.stages[early_last - 1].count;
        std::set<int> ox{tab.ex, tab.ex + tab.ew}, oy{tab.ey, tab.ey + tab.eh};
        for (int k = 0; k < k1; k++)
            for (int q = 0; q < (tab.host[k].nrect & 255); q++) { ox.insert(tab.host[k].x0[q]); ox.insert(tab.host[k].x1[q]); oy.insert(tab.host[k].y0[q]); oy.insert(tab.host[k].y1[q]); }
        for (int rx = 0; rx < t.nx; rx++) for (int o : ox) if (!cs.count(dp.pos[sc.xpos_off + t.ix0 + rx] + o)) return fail("a window's sample column is not staged");
        for (int ry = 0; ry < t.ny; ry++) for (int o : oy) if (!rs.count(dp.pos[sc.ypos_off + t.iy0 + ry] + o)) return fail("a window's sample row is not staged");
    }
    if (!dp.tiles.empty() && dp.strips.empty() && tiled != windows) return fail("tiles do not cover the grid exactly once");
    long long band_tiles = 0;
    for (const BandRec &b : dp.bands) {
        band_tiles += b.ntiles;
        if (b.first_tile < 0 || b.first_tile + b.ntiles > (int)dp.tiles.size()) return fail("band tile range");
        for (int k = 0; k < b.ntiles; k++) { const TileRec &t = dp.tiles[b.first_tile + k]; if (t.scale != b.scale || t.iy0 != b.iy0 || t.ny != b.ny || t.ix0 != k * dp.tiles[b.first_tile].nx) return fail("band is not a row of tiles"); }
    }
    if (!dp.bands.empty() && band_tiles != (long long)dp.tiles.size()) return fail("bands do not cover the tiles");
    for (const DeepRec &d : dp.deeprecs) if (d.ncol && ((size_t)d.col_off + d.ncol > dp.tcoords.size() || (size_t)d.row_off + d.nrow > dp.tcoords.size() || d.nrow * (d.ncol | 1) * 4 > dp.deep_lds)) return fail("deep patch");
    // candidate keys round-trip through hit_valid / hit_rect for every scale's last window, and junk keys are refused
    for (size_t s = 0; s < dp.specs.size(); s++) {
        if (dp.specs[s].xs.empty() || dp.specs[s].ys.empty()) continue;
        const unsigned key = ((unsigned)s << dp.key_ss) | ((unsigned)(dp.specs[s].ys.size() - 1) << dp.key_sy) | (unsigned)(dp.specs[s].xs.size() - 1);
        if (!dp.hit_valid(key)) return fail("hit_valid refuses a real window");
        const nvca_rect r = dp.hit_rect(key);
        if (r.x + r.w > cols || r.y + r.h > rows) return fail("window outside the image");
        // one past the grid in x / in y, where the key's field can express it (the fields are sized for the plan's largest grid: a
        // full field would carry into the next one and name another, real window)
        const size_t nx = dp.specs[s].xs.size(), ny = dp.specs[s].ys.size();
        if (nx < (1u << dp.key_sy) && dp.hit_valid(key + 1)) return fail("hit_valid accepts a window beyond the grid (x)");
        if (ny < (1u << (dp.key_ss - dp.key_sy)) && dp.hit_valid(key + (1u << dp.key_sy))) return fail("hit_valid accepts a window beyond the grid (y)");
    }
    if (dp.key_ss < 32 && dp.hit_valid((unsigned)dp.specs.size() << dp.key_ss)) return fail("hit_valid accepts an unknown scale");
    (void)c;
    return 0;
}

static int run_plans(const char *xml_path)
{
    const std::string xml = slurp(xml_path);
    nvca_ctx ctx;
    nvca_cascade casc; casc.ctx = &ctx;
    std::string err;
    if (parse_cascade_xml(xml.data(), xml.size(), casc.c, err)) return fail(err.c_str());
    casc.c.uid = ctx.next_uid++;
    struct G { int w, h; double sf; int minw, minh; } geoms[] = {
        {1920, 1080, 1.1, 96, 54}, {1280, 720, 1.1, 64, 36}, {640, 480, 1.25, 32, 24}, {160, 120, 1.25, 8, 6}, {160, 90, 1.25, 8, 4},
        {97, 83, 1.1, 3, 3}, {25, 25, 1.1, 0, 0}, {21, 400, 1.3, 0, 0}, {1400, 300, 1.2, 0, 0}, {3840, 2160, 1.2, 192, 108}, {320, 180, 1.1, 20, 20},
    };
    for (const G &g : geoms) {
        for (int variant = 0; variant < 3; variant++) {
            ctx.sw = Switches();
            if (variant == 1) ctx.sw.tiles = false;
            if (variant == 2) ctx.sw.deep_stage = 2;
            const int pitch = (g.w + 1 + 7) / 8 * 8;
            DetectPlan dp;
            const int rc = dp.build_scale_cascade(&ctx, casc.c, g.w, g.h, pitch, g.sf, g.minw, g.minh, g.w, g.h, err);
            if (rc) { printf("{\"plan\": [%d, %d, %.17g, %d, %d], \"variant\": %d, \"rc\": %d}\n", g.w, g.h, g.sf, g.minw, g.minh, variant, rc); continue; }
            if (check_plan(casc.c, dp, g.w, g.h, pitch)) return 1;
            printf("{\"plan\": [%d, %d, %.17g, %d, %d], \"variant\": %d, \"rc\": 0, \"factors\": [", g.w, g.h, g.sf, g.minw, g.minh, variant);
            for (size_t s = 0; s < dp.scales.size(); s++) printf("%s%.17g", s ? ", " : "", dp.scales[s].factor);
            printf("], \"grid\": [");
            for (size_t s = 0; s < dp.scales.size(); s++) printf("%s[%d, %d]", s ? ", " : "", dp.scales[s].endX, dp.scales[s].endY);
            printf("], \"tiles\": %zu, \"bands\": %zu, \"strips\": %zu, \"tile_lds\": %d}\n", dp.tiles.size(), dp.bands.size(), dp.strips.size(), dp.tile_lds);
        }
    }
    // resize tables over awkward ratios
    for (int t = 0; t < 200; t++) {
        ResizeTab tab;
        const int sw = rnd_in(1, 2000), sh = rnd_in(1, 1200), dw = rnd_in(1, 700), dh = rnd_in(1, 500);
        build_resize_tab(sw, sh, dw, dh, tab);
        if (tab.mode == 1)
            for (int dx = 0; dx < dw; dx++) if (tab.xofs[dx] < 0 || tab.xofs[dx] >= sw) return fail("resize column offset outside the source");
    }
    return 0;
}

// ---- glue --------------------------------------------------------------------------------------------------------------
static void print_rects(const char *tag, int id, const std::vector<nvca_rect> &v, const std::vector<int> *extra = nullptr)
{
    printf("{\"%s\": %d, \"out\": [", tag, id);
    for (size_t i = 0; i < v.size(); i++) printf("%s[%d, %d, %d, %d]", i ? ", " : "", v[i].x, v[i].y, v[i].w, v[i].h);
    printf("]");
    if (extra) { printf(", \"extra\": ["); for (size_t i = 0; i < extra->size(); i++) printf("%s%d", i ? ", " : "", (*extra)[i]); printf("]"); }
    printf("}\n");
}
static std::vector<nvca_rect> read_rects(std::istream &in)
{
    int n; in >> n;
    std::vector<nvca_rect> v(n);
    for (nvca_rect &r : v) in >> r.x >> r.y >> r.w >> r.h;
    return v;
}

// cases come from the pytest (which also hands them to the oracle): one per line,
//   G id thr eps n rects...            groupRectangles
//   J id min max dist n rects...       __join_objects
//   T id thr nframes {n rects...}      Faces::track over a sequence of detections
static int run_glue(const char *cases_path)
{
    std::ifstream in(cases_path);
    std::string kind;
    while (in >> kind) {
        int id; in >> id;
        if (kind == "G") {
            int thr; double eps; in >> thr >> eps;
            std::vector<nvca_rect> v = read_rects(in); std::vector<int> w;
            group_rectangles(v, thr, eps, &w);
            print_rects("group", id, v, &w);
        } else if (kind == "J") {
            int mn, dist; long mx; in >> mn >> mx >> dist;
            std::vector<nvca_rect> v = read_rects(in);
            join_objects(v, mn, mx, dist);
            print_rects("join", id, v);
        } else if (kind == "T") {
            int thr, nf; in >> thr >> nf;
            Faces f;
            for (int k = 0; k < nf; k++) {
                std::vector<nvca_rect> cur = read_rects(in);
                if (!cur.empty()) f.track(cur, thr); else if (k % 3 == 2) f.clear();
            }
            std::vector<nvca_rect> boxes; std::vector<int> ids;
            for (const TrackedFace &t : f.faces) { boxes.push_back(t.box); ids.push_back(t.id); }
            print_rects("track", id, boxes, &ids);
        } else return fail("unknown case kind");
    }
    // the part detectors' merging heuristics on random lists (their parity with the oracle is the GPU stream tests' business:
    // here they run under the sanitizers, with the list shapes -- 0..6 boxes, nested, coincident -- that exercise every erase)
    unsigned long long sum = 0;
    for (int t = 0; t < 20000; t++) {
        auto boxes = [&](int nmax) {
            std::vector<nvca_rect> v(rnd_in(0, nmax));
            for (nvca_rect &r : v) { r.x = rnd_in(0, 300); r.y = rnd_in(0, 200); r.w = rnd_in(1, 120); r.h = rnd_in(1, 120); }
            if (v.size() > 1 && rnd() % 3 == 0) v[1] = v[0];
            if (v.size() > 2 && rnd() % 3 == 0) { v[2].x = v[0].x + 2; v[2].y = v[0].y + 2; v[2].w = std::max(1, v[0].w - 4); v[2].h = std::max(1, v[0].h - 4); }
            return v;
        };
        const nvca_rect face{rnd_in(0, 200), rnd_in(0, 150), rnd_in(20, 200), rnd_in(20, 200)};
        const int scale = rnd_in(1, 6);
        std::vector<nvca_rect> a = boxes(6), b = boxes(6), old = boxes(4), res;
        to_global(a, face, scale);
        merge_eyes_current(face, b, a, scale, rnd() & 1);
        merge_eyes_consecutive(a, old, res);
        for (const nvca_rect &r : res) sum += (unsigned)(r.x * 3 + r.y * 5 + r.w * 7 + r.h * 11);
        std::vector<nvca_rect> cn = boxes(5);
        merge_consecutive_nm(cn, old, face, scale, rnd_in(1, 12), res);
        for (const nvca_rect &r : res) sum += (unsigned)(r.x + r.y + r.w + r.h);
    }
    // image-to-overlay on host frames: boxes that stick out of the frame on every side, every channel count, tiny and huge scales
    for (int t = 0; t < 300; t++) {
        const int W = rnd_in(1, 90), H = rnd_in(1, 70), stride = W * 3 + rnd_in(0, 5), cn = (int[]){1, 3, 4}[rnd() % 3], iw = rnd_in(1, 40), ih = rnd_in(1, 30);
        std::vector<uint8_t> frame((size_t)stride * H), img((size_t)iw * cn * ih);
        for (uint8_t &b : frame) b = (uint8_t)rnd();
        for (uint8_t &b : img) b = (uint8_t)rnd();
        nvca_overlay ov{img.data(), iw, ih, iw * cn, cn, (rnd_in(-20, 20)) / 10.0, (rnd_in(-20, 20)) / 10.0, rnd_in(0, 30) / 10.0, rnd_in(0, 30) / 10.0};
        std::vector<nvca_rect> bx(rnd_in(0, 3));
        for (nvca_rect &r : bx) { r.x = rnd_in(-60, 100); r.y = rnd_in(-60, 80); r.w = rnd_in(0, 120); r.h = rnd_in(0, 100); }
        overlay_blend_host(frame.data(), W, H, stride, bx.data(), (int)bx.size(), ov);
        for (uint8_t b : frame) sum += b;
    }
    printf("{\"merges_checksum\": %llu}\n", sum);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: san_driver loader <xml>... | plans <xml> | glue <cases>\n"); return 2; }
    const std::string mode = argv[1];
    if (mode == "loader") return run_loader(argc - 2, argv + 2);
    if (mode == "plans") return run_plans(argv[2]);
    if (mode == "glue") return run_glue(argv[2]);
    if (mode == "stages") {            // the per-stage summation-order proof of a cascade (StageRec flags), for DESIGN / experiments
        const std::string xml = slurp(argv[2]);
        Cascade c; std::string err;
        if (parse_cascade_xml(xml.data(), xml.size(), c, err)) return fail(err.c_str());
        std::vector<StageRec> st; build_stage_recs(c, st);
        for (size_t i = 0; i < st.size(); i++) printf("{\"stage\": %zu, \"count\": %d, \"flags\": %d, \"thr_i\": %d, \"vote_exp\": %d}\n", i, st[i].count, st[i].flags, st[i].thr_i, st[i].vote_exp);
        return 0;
    }
    return 2;
}
