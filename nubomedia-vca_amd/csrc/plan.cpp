// plan.cpp -- host-side table builders.  Everything cvRound / float-coefficient
// related is computed here, once per (cascade, geometry), so that the kernels do
// integer and IEEE add/mul work only.
//
// Replaces (OpenCV 2.4, called from FACE/kmsfacedetect.cpp:805,809-811):
//   cv::resize's coefficient tables            (imgwarp.cpp)
//   cvHaarDetectObjectsForROC's scale loop      (haar.cpp)
//   cvSetImagesForHaarClassifierCascade         (haar.cpp)
#include "plan.h"
#include <cmath>
#include <cfloat>
#include <cstring>
#include <algorithm>
#include <climits>
#include <cstdlib>

namespace nvca {

static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_floor(double v) { return (int)floor(v); }
static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

void build_resize_tab(int sw, int sh, int dw, int dh, ResizeTab &t)
{
    t.sw = sw; t.sh = sh; t.dw = dw; t.dh = dh;
    t.xofs.assign(dw, 0); t.yofs.assign(dh, 0);
    t.ialpha.assign(2 * (size_t)dw, 0); t.ibeta.assign(2 * (size_t)dh, 0);
    t.xmax = dw;
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int iscale_x = cv_round(scale_x), iscale_y = cv_round(scale_y);
    bool is_area_fast = std::fabs(scale_x - iscale_x) < DBL_EPSILON && std::fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (sw == dw && sh == dh) { t.mode = 0; return; }           // bilinear at scale 1 is the identity
    if (is_area_fast && iscale_x == 2 && iscale_y == 2) { t.mode = 2; return; }
    t.mode = 1;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            t.xmax = std::min(t.xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        t.xofs[dx] = sx;
        t.ialpha[2 * dx] = sat_short(cv_round((1.f - fx) * 2048));
        t.ialpha[2 * dx + 1] = sat_short(cv_round(fx * 2048));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        t.yofs[dy] = sy;
        t.ibeta[2 * dy] = sat_short(cv_round((1.f - fy) * 2048));
        t.ibeta[2 * dy + 1] = sat_short(cv_round(fy * 2048));
    }
}

// cvHaarDetectObjectsForROC, scale-cascade branch: which factors are evaluated.
void scale_grid(int ow, int oh, int cols, int rows, double scaleFactor, int minw, int minh,
                int maxw, int maxh, bool findBiggest, std::vector<double> &factors)
{
    factors.clear();
    if (maxh == 0 || maxw == 0) { maxh = rows; maxw = cols; }
    int n_factors = 0; double factor;
    for (n_factors = 0, factor = 1; factor * ow < cols - 10 && factor * oh < rows - 10; n_factors++, factor *= scaleFactor)
        ;
    if (findBiggest) { scaleFactor = 1. / scaleFactor; factor *= scaleFactor; } else factor = 1;
    for (; n_factors-- > 0; factor *= scaleFactor) {
        int winw = cv_round(ow * factor), winh = cv_round(oh * factor);
        if (winw < minw || winh < minh) { if (findBiggest) break; continue; }
        if (winw > maxw || winh > maxh) { if (!findBiggest) break; continue; }
        factors.push_back(factor);
    }
}

// cvSetImagesForHaarClassifierCascade for one factor, geometry-independent (see plan.h)
void build_scale_table(const Cascade &c, double factor, ScaleTable &t)
{
    t.factor = factor;
    t.ex = cv_round(factor); t.ey = t.ex;
    t.ew = cv_round((c.ow - 2) * factor); t.eh = cv_round((c.oh - 2) * factor);
    const double weight_scale = 1. / (t.ew * t.eh);
    t.winw = cv_round(c.ow * factor); t.winh = cv_round(c.oh * factor);
    t.inv_area = weight_scale;
    t.ghost.clear(); t.galpha.clear(); t.gcls_first.clear();
    if (c.generic()) {
        // every node of every weak classifier; corners in the order p0 - p1 - p2 + p3 of cvSetImagesForHaarClassifierCascade
        t.host.clear();
        t.galpha = c.alpha;
        for (const HaarClassifier &hc : c.cls) {
            t.gcls_first.push_back((int)t.ghost.size());
            for (int l = 0; l < hc.nnodes; l++) {
                const HaarNode &n = c.nodes[hc.first_node + l];
                GNodeRec r; memset(&r, 0, sizeof(r));
                double sum0 = 0, area0 = 0;
                for (int q = 0; q < n.nrect; q++) {
                    const int tx = cv_round(n.rect[q][0] * factor), tw = cv_round(n.rect[q][2] * factor);
                    const int ty = cv_round(n.rect[q][1] * factor), th = cv_round(n.rect[q][3] * factor);
                    const double correction_ratio = weight_scale * (!n.tilted ? 1 : 0.5);
                    if (!n.tilted) {
                        r.dx[q][0] = (short)tx;        r.dy[q][0] = (short)ty;
                        r.dx[q][1] = (short)(tx + tw); r.dy[q][1] = (short)ty;
                        r.dx[q][2] = (short)tx;        r.dy[q][2] = (short)(ty + th);
                        r.dx[q][3] = (short)(tx + tw); r.dy[q][3] = (short)(ty + th);
                    } else {
                        r.dx[q][0] = (short)tx;             r.dy[q][0] = (short)ty;
                        r.dx[q][1] = (short)(tx - th);      r.dy[q][1] = (short)(ty + th);
                        r.dx[q][2] = (short)(tx + tw);      r.dy[q][2] = (short)(ty + tw);
                        r.dx[q][3] = (short)(tx + tw - th); r.dy[q][3] = (short)(ty + tw + th);
                    }
                    r.w[q] = (float)(n.weight[q] * correction_ratio);
                    if (q == 0) area0 = tw * th;
                    else {
                        float tt = r.w[q] * tw;      // float * int -> float, evaluated left to right
                        tt = tt * th;
                        sum0 += tt;
                    }
                }
                r.w[0] = (float)(-sum0 / area0);
                r.thr = n.threshold;
                // children: > 0 a node of this classifier, <= 0 a leaf whose value is alpha[first_alpha - v]
                r.left = n.left > 0 ? n.left : -(hc.first_alpha + (-n.left));
                r.right = n.right > 0 ? n.right : -(hc.first_alpha + (-n.right));
                r.flags = n.nrect | (n.tilted ? 256 : 0);
                t.ghost.push_back(r);
            }
        }
        return;
    }
    t.host.assign(c.cls.size(), TStumpRec());
    size_t k = 0;
    for (const HaarClassifier &hc : c.cls) {
        const HaarNode &n = c.nodes[hc.first_node];     // stump
        TStumpRec &r = t.host[k++];
        memset(&r, 0, sizeof(r));
        double sum0 = 0, area0 = 0;
        for (int q = 0; q < n.nrect; q++) {
            int tx = cv_round(n.rect[q][0] * factor), tw = cv_round(n.rect[q][2] * factor);
            int ty = cv_round(n.rect[q][1] * factor), th = cv_round(n.rect[q][3] * factor);
            double correction_ratio = weight_scale;       // upright feature, CV_ADJUST_WEIGHTS 0
            r.x0[q] = tx; r.x1[q] = tx + tw; r.y0[q] = ty; r.y1[q] = ty + th;
            r.w[q] = (float)(n.weight[q] * correction_ratio);
            if (q == 0) area0 = tw * th;
            else {
                float tt = r.w[q] * tw;      // float * int -> float, evaluated left to right
                tt = tt * th;
                sum0 += tt;
            }
        }
        r.w[0] = (float)(-sum0 / area0);
        r.thr = (double)n.threshold;
        r.a0 = (double)c.alpha[hc.first_alpha]; r.a1 = (double)c.alpha[hc.first_alpha + 1];
        r.nrect = n.nrect;
        // an x2 / x3 feature's rectangles span the same rows, a y2 / y3 feature's the same columns (same y and height, or
        // same x and width, go through the same rounding): the tile kernels look the shared pair up once
        for (int q = 1; q < n.nrect; q++) {
            if (r.y0[q] == r.y0[0] && r.y1[q] == r.y1[0]) r.nrect |= 256 << (2 * (q - 1));
            if (r.x0[q] == r.x0[0] && r.x1[q] == r.x1[0]) r.nrect |= 512 << (2 * (q - 1));
        }
    }
    // the compact per-lane form of the same records (kernels_cascade.hip, lane_vote): coordinates as byte offsets into u16 maps
    t.lhost.assign(t.host.size(), LStumpRec());
    for (size_t i = 0; i < t.host.size(); i++) {
        const TStumpRec &r = t.host[i];
        LStumpRec &l = t.lhost[i];
        memset(&l, 0, sizeof(l));
        auto pk = [](int a, int b) { return (unsigned)(2 * a) | (unsigned)(2 * b) << 16; };
        l.xx0 = pk(r.x0[0], r.x1[0]); l.yy0 = pk(r.y0[0], r.y1[0]);
        l.xx1 = pk(r.x0[1], r.x1[1]); l.yy1 = pk(r.y0[1], r.y1[1]);
        if ((r.nrect & 255) == 3) { l.xx2 = pk(r.x0[2], r.x1[2]); l.yy2 = pk(r.y0[2], r.y1[2]); }
        l.w0 = r.w[0]; l.w1 = r.w[1]; l.w2 = (r.nrect & 255) == 3 ? r.w[2] : 0.f;
        l.thr = (float)r.thr; l.a0 = (float)r.a0; l.a1 = (float)r.a1;       // all three came from floats: the round trip is exact
    }
    // integer votes of the stages whose sums are provably exact as 32-bit integers (StageRec flag bit 2)
    std::vector<StageRec> st;
    build_stage_recs(c, st);
    for (const StageRec &sr : st) {
        if (sr.flags & 1)               // a stage of two-rectangle stumps: where the SSE2 pair form applies, marked in the per-lane records (lane_vote)
            for (int j = 0; j < sr.count; j++) t.lhost[sr.first + j].yy2 = 1;
        if (!(sr.flags & 4)) continue;
        for (int j = 0; j < sr.count; j++) {
            TStumpRec &r = t.host[sr.first + j];
            r.a0i = (int)std::ldexp(r.a0, -sr.vote_exp); r.a1i = (int)std::ldexp(r.a1, -sr.vote_exp);
        }
    }
}

// lowest set bit of a float as a power of two (INT_MAX for 0)
static int lsb_exponent(float x)
{
    if (x == 0.f || !std::isfinite(x)) return x == 0.f ? INT_MAX : INT_MIN;
    int ex; const double fr = std::frexp(std::fabs((double)x), &ex);      // fr in [0.5,1)
    unsigned long long m = (unsigned long long)std::ldexp(fr, 53);          // exact integer mantissa
    int tz = 0; while (!(m & 1ull)) { m >>= 1; tz++; }
    return ex - 53 + tz;
}

void build_stage_recs(const Cascade &c, std::vector<StageRec> &out)
{
    out.clear();
    for (const HaarStage &s : c.stages) {
        StageRec r; r.first = s.first_cls; r.count = s.ncls;
        r.thr = s.threshold - 0.0001f;      // icv_stage_threshold_bias, float arithmetic
        bool two_rects = true;
        // The stage sum is a left-to-right f64 accumulation of f32 votes.  If every vote is a multiple
        // of 2^e and the sum of |votes| stays below 2^(e+53), every partial sum of every subset is
        // exactly representable, so any summation order gives OpenCV's result bit for bit.
        int emin = INT_MAX; double bound = 0; bool finite = true;
        for (int j = 0; j < s.ncls; j++) {
            const HaarClassifier &hc = c.cls[s.first_cls + j];
            if (c.nodes[hc.first_node].nrect == 3) two_rects = false;
            const float a0 = c.alpha[hc.first_alpha], a1 = c.alpha[hc.first_alpha + 1];
            if (!std::isfinite(a0) || !std::isfinite(a1)) finite = false;
            emin = std::min(emin, std::min(lsb_exponent(a0), lsb_exponent(a1)));
            bound += std::max(std::fabs((double)a0), std::fabs((double)a1));
        }
        bool order_free = finite && (emin == INT_MAX || (bound == 0) || std::ilogb(bound) + 1 <= emin + 52);
        r.flags = (two_rects ? 1 : 0) | (order_free ? 2 : 0);
        // integer votes: every vote an exact multiple of 2^emin, the sum of their magnitudes (hence every partial sum of every
        // subset) below 2^31 multiples, and the threshold comparable as an integer:  !(S * 2^e < T)  <=>  S >= ceil(T / 2^e)
        r.thr_i = 0; r.vote_exp = 0; r.spec_run = 0; r.pad1 = 0;
        if (finite && emin != INT_MAX && emin != INT_MIN && emin > -1000 && emin < 1000 && std::ldexp(bound, -emin) < 2147483000.0) {
            const double tq = std::ceil(std::ldexp((double)r.thr, -emin));
            if (std::isfinite(tq) && std::fabs(tq) < 2147483000.0) { r.flags |= 4 | 2; r.thr_i = (int)tq; r.vote_exp = emin; }
        }
        out.push_back(r);
    }
    for (int i = (int)out.size() - 1; i >= 0; i--)
        out[i].spec_run = (out[i].flags & 2) ? 1 + (i + 1 < (int)out.size() ? out[i + 1].spec_run : 0) : 0;
}

// Dispatch order.  Workgroups are dealt round-robin over the 8 XCDs (b % 8 shares an XCD), each with its
// own 4 MiB L2.  Give every XCD one contiguous run of work items (scale-major, then y, x) of about equal
// weight, so that the integral rows an XCD reads stay resident in its L2 (speed only, never results).
static int build_xcd_order(const std::vector<long long> &weight, std::vector<int> &order)
{
    const int n = (int)weight.size();
    long long total = 0;
    for (long long w : weight) total += w;
    std::vector<int> start(9, n);
    start[0] = 0;
    long long acc = 0; int k = 1;
    for (int i = 0; i < n && k < 8; i++) {
        acc += weight[i];
        while (k < 8 && acc * 8 >= total * k) { start[k] = i + 1; k++; }
    }
    for (; k < 8; k++) start[k] = n;
    int maxlen = 0;
    for (int x = 0; x < 8; x++) maxlen = std::max(maxlen, start[x + 1] - start[x]);
    order.assign((size_t)8 * maxlen, -1);
    for (int x = 0; x < 8; x++)
        for (int j = 0; j < start[x + 1] - start[x]; j++) order[(size_t)j * 8 + x] = start[x] + j;
    return 8 * maxlen;
}

bool DetectPlan::hit_valid(unsigned key) const
{
    const size_t s = key >> key_ss, iy = (key >> key_sy) & ((1u << (key_ss - key_sy)) - 1u), ix = key & ((1u << key_sy) - 1u);
    return s < specs.size() && ix < specs[s].xs.size() && iy < specs[s].ys.size();
}

nvca_rect DetectPlan::hit_rect(unsigned key) const
{
    const int s = key >> key_ss, iy = (key >> key_sy) & ((1u << (key_ss - key_sy)) - 1u), ix = key & ((1u << key_sy) - 1u);
    const ScaleSpec &sp = specs[s];
    const int x = sp.xs[ix], y = sp.ys[iy];
    if (sp.out_factor != 0) return nvca_rect{cv_round(x * sp.out_factor), cv_round(y * sp.out_factor), sp.out_w, sp.out_h};
    return nvca_rect{x, y, sp.out_w, sp.out_h};
}

int DetectPlan::build_custom(nvca_ctx *ctx, const Cascade &c, std::vector<ScaleSpec> &&in, bool allow_tiles, std::string &err)
{
    specs = std::move(in);
    const int ns = (int)c.stages.size();
    if (ctx->sw.deep_stage > 0) { deep_stage = ctx->sw.deep_stage; return build_tables(ctx, c, allow_tiles, err); }
    if (!ctx->sw.tiles || c.generic()) { deep_stage = 6; return build_tables(ctx, c, allow_tiles, err); }      // row strips / the general evaluator: no tiles to walk
    // By default the tile kernels walk the WHOLE cascade (deep_stage == number of stages, no late-stage kernel): the corner offsets of
    // all stages fall on the same near-lattice as those of the first six, so a tile that holds them is a few per cent larger
    // (1080p, 22 stages: 56 KB instead of 53).  A cascade whose late stages do not fit -- a scale left without tiles, or tiles
    // squeezed below 20 windows a side -- keeps the split: early stages on tiles, the rest in k_deep.
    deep_stage = ns;
    int rc = build_tables(ctx, c, allow_tiles, err);
    if (rc == NVCA_OK && !generic && ctx->sw.tiles && ns > 6 && (!strips.empty() || (min_tile_side > 0 && min_tile_side < std::min(max_tile_side, 20)))) {
        deep_stage = 6;
        rc = build_tables(ctx, c, allow_tiles, err);
    }
    return rc;
}

int DetectPlan::build_tables(nvca_ctx *ctx, const Cascade &c, bool allow_tiles, std::string &err)
{
    generic = c.generic(); needs_tilted = c.has_tilted; generic_stumps = c.stump_based;
    nstumps = (int)c.cls.size();
    scales.clear(); strips.clear(); pos.clear(); tasks.clear(); tiles.clear(); release_tables();
    min_tile_side = 0; max_tile_side = 0;
    // stages 1 .. deep_stage-1 run on LDS lattice tiles (k_tile); NVCA_TILES=0 selects the older row strips (k_strip)
    bool use_tiles = ctx->sw.tiles;
    if (generic) use_tiles = false;              // the LDS tile kernels are built around upright stumps
    (void)allow_tiles;
    tcoords.clear(); tile_lds = 0; bands.clear(); band_order.clear(); band_blocks_per_frame = 0;
    {   // the candidate key: scale | row | column of the window, each field as wide as THIS plan's grids need (the reference installs
        // multi-scale-factor with range 0 .. 51 and no clamp, FACE/kmsfacedetect.cpp:540-542: 1.01 on a 640 x 360 working image is a
        // ladder of 288 scales -- next to grids of at most 310 x 170 windows that is 9 + 8 + 9 bits)
        auto bits = [](size_t n) { int b = 1; while ((1ull << b) < n) b++; return b; };
        size_t mx = 1, my = 1;
        for (const ScaleSpec &sp : specs) { mx = std::max(mx, sp.xs.size()); my = std::max(my, sp.ys.size()); }
        const int bx = bits(mx), by = bits(my), bs = bits(std::max<size_t>(specs.size(), 1));
        if (bx + by + bs > 32 || specs.size() > 4095) { err = "scan too large for the candidate key (scales x rows x columns of windows beyond 2^32)"; return NVCA_ERR_ARG; }
        key_sy = bx; key_ss = bx + by;
    }
    if (c.stage_rec_cache.empty()) {            // once per cascade
        std::vector<StageRec> tmp; build_stage_recs(c, tmp);
        c.stage_rec_cache.resize(tmp.size() * sizeof(StageRec));
        if (!tmp.empty()) memcpy(c.stage_rec_cache.data(), tmp.data(), c.stage_rec_cache.size());
    }
    stages.resize(c.stage_rec_cache.size() / sizeof(StageRec));
    if (!stages.empty()) memcpy(stages.data(), c.stage_rec_cache.data(), c.stage_rec_cache.size());
    stage_first.clear();
    for (const StageRec &sr : stages) stage_first.push_back(sr.first);
    stage_first.push_back(stages.empty() ? 0 : stages.back().first + stages.back().count);
    stage_first.insert(stage_first.end(), 8, INT_MAX);
    stage_thr.clear();
    for (const StageRec &sr : stages) stage_thr.push_back(sr.thr);
    stage_thr.insert(stage_thr.end(), 8, 0.f);
    std::vector<long long> strip_w, tile_w;
    // A scan with few windows (the part detectors' working images and ROIs) cannot fill the GPU with 32 x 32-window tiles: a
    // handful of workgroups would each walk a long chain of stages.  Smaller tiles give more workgroups and, with the same 1024
    // threads per tile, more stump partitions per window -- a shorter chain.  (Results do not depend on the tiling.)
    int max_tile = kTileWin;
    {
        long long tiles32 = 0;
        for (const ScaleSpec &sp : specs) tiles32 += (long long)((sp.xs.size() + kTileWin - 1) / kTileWin) * (long long)((sp.ys.size() + kTileRows - 1) / kTileRows);
        if (tiles32 < 128) max_tile = 16;
    }
    for (size_t s = 0; s < specs.size(); s++) {
        const ScaleSpec &sp = specs[s];
        const int pitch = sp.pitch;
        ScaleRec sr; memset(&sr, 0, sizeof(sr));
        ScaleTable *tabp = get_scale_table(ctx, c, sp.table_factor);     // pyramid levels (factor 1) all share one table
        if (!tabp) { err = "allocation failed (stump tables)"; return NVCA_ERR_NOMEM; }
        tabp->refs++; tabs.push_back(tabp);
        sr.winw = tabp->winw; sr.winh = tabp->winh; sr.inv_area = tabp->inv_area; sr.factor = tabp->factor;
        sr.sq32 = (unsigned long long)tabp->ew * (unsigned long long)tabp->eh * 65025ull < (1ull << 32) ? 1 : 0;
        sr.eq[0] = tabp->ey * pitch + tabp->ex;               sr.eq[1] = tabp->ey * pitch + tabp->ex + tabp->ew;
        sr.eq[2] = (tabp->ey + tabp->eh) * pitch + tabp->ex;  sr.eq[3] = (tabp->ey + tabp->eh) * pitch + tabp->ex + tabp->ew;
        sr.trecs = tabp->dev.as<TStumpRec>();
        sr.lrecs = tabp->d_lrecs;
        sr.grecs = tabp->d_grecs;
        if (generic) {
            // every corner of every node must stay inside the planes for every window of the grid (a faulting read takes the
            // whole GPU down): columns 0 .. pitch-1, rows 0 .. plane_rows-1
            int mnx = 0, mxx = 0, mny = 0, mxy = 0;
            for (const GNodeRec &g : tabp->ghost)
                for (int q = 0; q < (g.flags & 255); q++)
                    for (int e = 0; e < 4; e++) { mnx = std::min<int>(mnx, g.dx[q][e]); mxx = std::max<int>(mxx, g.dx[q][e]); mny = std::min<int>(mny, g.dy[q][e]); mxy = std::max<int>(mxy, g.dy[q][e]); }
            if (!sp.xs.empty() && !sp.ys.empty() &&
                (sp.xs.front() + mnx < 0 || sp.xs.back() + mxx >= pitch || sp.ys.front() + mny < 0 || sp.ys.back() + mxy >= sp.plane_rows + 2)) {      // the planes are allocated with a few spare rows (api.cpp)
                err = "a feature of the cascade leaves the image at this scale"; return NVCA_ERR_UNSUPPORTED;
            }
        }
        sr.plane_off = sp.plane_off; sr.pitch = pitch; sr.adaptive = sp.adaptive;
        sr.endX = (int)sp.xs.size(); sr.endY = (int)sp.ys.size();
        if (sr.endY > 8191) { err = "image too large for the task key"; return NVCA_ERR_ARG; }
        sr.xpos_off = (int)pos.size(); pos.insert(pos.end(), sp.xs.begin(), sp.xs.end());
        sr.ypos_off = (int)pos.size(); pos.insert(pos.end(), sp.ys.begin(), sp.ys.end());
        sr.task_off = (int)tasks.size();
        sr.wpr = sr.endX > 0 ? (sr.endX + 63) / 64 : 0;
        if (sr.wpr > 128) { err = "scan row too long for the task key"; return NVCA_ERR_ARG; }
        if (sr.endX > 0)
            for (int iy = 0; iy < sr.endY; iy++)
                for (int k = 0; k < sr.wpr; k++) tasks.push_back(((unsigned)s << 20) | ((unsigned)iy << 7) | (unsigned)k);
        scales.push_back(sr);
        if (sr.endX <= 0 || sr.endY <= 0) continue;
        const int *xp = &pos[sr.xpos_off], *yp = &pos[sr.ypos_off];
        // ---- LDS lattice tiles: stages 1 .. early_last-1 read only (window origin + scaled corner) samples; per tile of
        // n x n windows those are ~2.7 (n + 20) distinct columns and rows whatever the scale.  Stage exactly them.
        int tw = 0;
        std::vector<int> offx, offy;
        const int early_last = std::min<int>(deep_stage, (int)stages.size());
        if (use_tiles && early_last > 1) {
            // stage 0 and the variance rectangle included: the band kernel evaluates them from the tile as well
            const int k0 = stages[0].first, k1 = stages[early_last - 1].first + stages[early_last - 1].count;
            { const auto &co = tabp->corner_offsets(k0, k1, true); offx = co.first; offy = co.second; }
            // distinct sample coordinates of windows [i0, i1) along one axis
            auto coords = [](const int *p, int i0, int i1, const std::vector<int> &off, std::vector<int> &out) {
                out.clear();
                for (int i = i0; i < i1; i++) for (int o : off) out.push_back(p[i] + o);
                std::sort(out.begin(), out.end()); out.erase(std::unique(out.begin(), out.end()), out.end());
            };
            std::vector<int> cx, cy;
            for (tw = max_tile; tw >= 1; tw--) {         // largest tile side whose every tile fits the LDS budget
                bool ok = true;
                int worst_c = 0, worst_r = 0, worst_sx = 0, worst_sy = 0;
                for (int i0 = 0; i0 < sr.endX; i0 += tw) {
                    coords(xp, i0, std::min(i0 + tw, sr.endX), offx, cx);
                    worst_c = std::max(worst_c, (int)cx.size()); worst_sx = std::max(worst_sx, cx.back() - xp[i0] + 1);
                }
                const int th = std::min(tw, kTileRows);         // a tile is tw windows wide and up to kTileRows high (a window per thread)
                for (int i0 = 0; i0 < sr.endY; i0 += th) {
                    coords(yp, i0, std::min(i0 + th, sr.endY), offy, cy);
                    worst_r = std::max(worst_r, (int)cy.size()); worst_sy = std::max(worst_sy, cy.back() - yp[i0] + 1);
                }
                ok = worst_c <= kTileMaxCols && worst_r <= kTileThreads && worst_r * tile_pitch(worst_c) < 65536 &&
                     tile_lds_bytes(worst_c, worst_r, worst_sx, worst_sy) <= kTileLdsBudget;
                if (ok) {
                    if (ctx->sw.plan_debug)
                        fprintf(stderr, "[nvca plan] scale %zu factor %.3f windows %d x %d: tile side %d, %d x %d samples, %d B of LDS (budget %d)\n", s, sp.table_factor > 0 ? sp.table_factor : sp.out_factor,
                                sr.endX, sr.endY, tw, worst_c, worst_r, tile_lds_bytes(worst_c, worst_r, worst_sx, worst_sy), kTileLdsBudget);
                    break;
                }
            }
            if (tw >= 1) {
                min_tile_side = min_tile_side ? std::min(min_tile_side, tw) : tw; max_tile_side = max_tile;
                const int th = std::min(tw, kTileRows);
                for (int iy0 = 0; iy0 < sr.endY; iy0 += th) {
                    coords(yp, iy0, std::min(iy0 + th, sr.endY), offy, cy);
                    BandRec b; memset(&b, 0, sizeof(b));
                    b.scale = (int)s; b.iy0 = iy0; b.ny = std::min(th, sr.endY - iy0); b.first_tile = (int)tiles.size();
                    b.ntiles = (sr.endX + tw - 1) / tw;
                    bands.push_back(b);
                    const int row_off = (int)tcoords.size();
                    for (int v : cy) tcoords.push_back((unsigned short)v);
                    for (int ix0 = 0; ix0 < sr.endX; ix0 += tw) {
                        coords(xp, ix0, std::min(ix0 + tw, sr.endX), offx, cx);
                        TileRec t; memset(&t, 0, sizeof(t));
                        t.scale = (int)s; t.ix0 = ix0; t.iy0 = iy0;
                        t.nx = std::min(tw, sr.endX - ix0); t.ny = std::min(th, sr.endY - iy0);
                        t.x0 = xp[ix0]; t.y0 = yp[iy0];
                        t.ncol = (int)cx.size(); t.nrow = (int)cy.size();
                        t.span_x = cx.back() - t.x0 + 1; t.span_y = cy.back() - t.y0 + 1;
                        t.col_off = (int)tcoords.size(); t.row_off = row_off;
                        for (int v : cx) tcoords.push_back((unsigned short)v);
                        tile_lds = std::max(tile_lds, tile_lds_bytes(t.ncol, t.nrow, t.span_x, t.span_y));
                        tiles.push_back(t);
                        tile_w.push_back((long long)t.nx * t.ny + 256);
                    }
                }
            } else tw = 0;
        }
        if (!tw) {
            if (sr.endX <= kStripMaxWin) {
                const int rows_per = std::max(1, std::min(kStripMaxWin / sr.endX, 64));
                for (int iy = 0; iy < sr.endY; iy += rows_per) {
                    StripRec st; memset(&st, 0, sizeof(st));
                    st.scale = (int)s; st.iy0 = iy; st.nrows = std::min(rows_per, sr.endY - iy); st.ix0 = 0; st.ncols = sr.endX;
                    strips.push_back(st);
                    strip_w.push_back((long long)st.nrows * st.ncols);
                }
            } else {                                 // long rows: equal segments of one row each
                const int nseg = (sr.endX + kStripMaxWin - 1) / kStripMaxWin, seg = (sr.endX + nseg - 1) / nseg;
                for (int iy = 0; iy < sr.endY; iy++)
                    for (int x0 = 0; x0 < sr.endX; x0 += seg) {
                        StripRec st; memset(&st, 0, sizeof(st));
                        st.scale = (int)s; st.iy0 = iy; st.nrows = 1; st.ix0 = x0; st.ncols = std::min(seg, sr.endX - x0);
                        strips.push_back(st);
                        strip_w.push_back((long long)st.ncols);
                    }
            }
        }
    }
    blocks_per_frame = build_xcd_order(strip_w, order);
    tile_blocks_per_frame = build_xcd_order(tile_w, tile_order);
    if (!strips.empty()) bands.clear();          // the band kernel has no strip counterpart: all scales tiled, or none
    if (!strips.empty() && use_tiles && !generic && !ctx->sw.quiet) {
        // a stump cascade some scale of which does not fit the tiles' LDS budget runs on the row-strip kernel: same results,
        // several times slower on large scans -- said once, not silently
        static bool noted = false;
        if (!noted) {
            noted = true;
            fprintf(stderr, "nubovca: a scan (%d x %d image, %zu scales) does not fit the LDS tile kernels and runs on the row-strip fallback (k_strip): same results, slower; NVCA_PLAN_DEBUG=1 prints the per-scale tile sizes (this note is printed once)\n",
                    cols, rows, specs.size());
        }
    }
    // late stages: per scale the distinct corner columns / rows of their stumps (k_deep's LDS patch of one window)
    deeprecs.clear(); deep_lds = 0;
    if (strips.empty() && !tiles.empty() && deep_stage < (int)stages.size() && ctx->sw.deep_lds) {
        deeprecs.resize(scales.size());
        std::vector<char> tiled(scales.size(), 0);
        for (const TileRec &t : tiles) tiled[t.scale] = 1;
        const int k0 = stages[deep_stage].first, k1 = stages.back().first + stages.back().count;
        for (size_t s = 0; s < scales.size(); s++) {
            DeepRec d; memset(&d, 0, sizeof(d));
            if (tiled[s]) {
                const auto &co = tabs[s]->corner_offsets(k0, k1, false);
                const std::vector<int> &ox = co.first, &oy = co.second;
                if (!ox.empty() && (int)ox.size() <= kDeepMaxSide && (int)oy.size() <= kDeepMaxSide && ox.back() < kDeepMaxSpan && oy.back() < kDeepMaxSpan) {
                    d.col_off = (int)tcoords.size(); for (int v : ox) tcoords.push_back((unsigned short)v);
                    d.row_off = (int)tcoords.size(); for (int v : oy) tcoords.push_back((unsigned short)v);
                    d.ncol = (int)ox.size(); d.nrow = (int)oy.size(); d.span_x = ox.back() + 1; d.span_y = oy.back() + 1;
                    deep_lds = std::max(deep_lds, d.nrow * (d.ncol | 1) * 4);
                }
            }
            deeprecs[s] = d;
        }
    }
    {   // bands: longest first (a band is a serial walk over its tiles; the short ones fill the tail)
        band_order.resize(bands.size());
        for (size_t i = 0; i < bands.size(); i++) band_order[i] = (int)i;
        std::stable_sort(band_order.begin(), band_order.end(), [&](int x, int y) { return bands[x].ntiles > bands[y].ntiles; });
        band_blocks_per_frame = (int)bands.size();
    }
    // per-scale segments of the global survivor lists (sized per frame; api.cpp scales them by the batch)
    device_group_ok = true;
    for (size_t q = 0; q < specs.size(); q++)
        if (specs[q].out_factor != 0 || specs[q].out_w != scales[q].winw || specs[q].out_h != scales[q].winh) device_group_ok = false;
    return NVCA_OK;
}

// cvHaarDetectObjectsForROC, scale-cascade branch (flags without SCALE_IMAGE): one pair of integral planes,
// features scaled by each factor, stride max(2, factor), adaptive x step.
int DetectPlan::build_scale_cascade(nvca_ctx *ctx, const Cascade &c, int cols, int rows, int pitch, double scaleFactor,
                                    int minw, int minh, int maxw, int maxh, std::string &err)
{
    std::vector<double> factors;
    scale_grid(c.ow, c.oh, cols, rows, scaleFactor, minw, minh, maxw, maxh, false, factors);
    std::vector<ScaleSpec> sp;
    for (double factor : factors) {
        const double ystep = std::max(2., factor);
        ScaleSpec s;
        s.table_factor = factor; s.plane_off = 0; s.pitch = pitch; s.plane_rows = rows + 1; s.adaptive = 1;
        s.out_factor = 0; s.out_w = cv_round(c.ow * factor); s.out_h = cv_round(c.oh * factor);
        const int endX = cv_round((cols - s.out_w) / ystep), endY = cv_round((rows - s.out_h) / ystep);
        for (int ix = 0; ix < endX; ix++) s.xs.push_back(cv_round(ix * ystep));
        for (int iy = 0; iy < endY; iy++) s.ys.push_back(cv_round(iy * ystep));
        // cvRunHaarClassifierCascadeSum's own bound (x + w < cols + 1) always holds inside the loop limits
        if (!s.xs.empty() && !s.ys.empty() && (s.xs.back() + s.out_w >= cols + 1 || s.ys.back() + s.out_h >= rows + 1)) {
            err = "scan grid leaves the image"; return NVCA_ERR_ARG;
        }
        sp.push_back(std::move(s));
    }
    return build_custom(ctx, c, std::move(sp), true, err);
}

const std::pair<std::vector<int>, std::vector<int>> &ScaleTable::corner_offsets(int k0, int k1, bool with_eq)
{
    const auto key = std::make_pair(with_eq ? k0 : -1 - k0, k1);
    auto it = offsets.find(key);
    if (it != offsets.end()) return it->second;
    std::vector<int> ox, oy;
    if (with_eq) { ox.push_back(ex); ox.push_back(ex + ew); oy.push_back(ey); oy.push_back(ey + eh); }
    for (int k = k0; k < k1; k++)
        for (int q = 0; q < (host[k].nrect & 255); q++) { ox.push_back(host[k].x0[q]); ox.push_back(host[k].x1[q]); oy.push_back(host[k].y0[q]); oy.push_back(host[k].y1[q]); }
    std::sort(ox.begin(), ox.end()); ox.erase(std::unique(ox.begin(), ox.end()), ox.end());
    std::sort(oy.begin(), oy.end()); oy.erase(std::unique(oy.begin(), oy.end()), oy.end());
    return offsets.emplace(key, std::make_pair(std::move(ox), std::move(oy))).first->second;
}

void DetectPlan::release_tables()
{
    for (ScaleTable *t : tabs) if (t && t->refs > 0) t->refs--;
    tabs.clear();
}

// Stump tables depend on (cascade, factor) only -- not on the image: every scale-cascade plan draws on the same ladder
// of factors 1.1^k, and a new ROI size (the part detectors meet one per face per frame) costs no table work at all.
ScaleTable *get_scale_table(nvca_ctx *ctx, const Cascade &c, double factor)
{
    uint64_t fb; memcpy(&fb, &factor, sizeof(fb));
    const auto key = std::make_pair((uint64_t)c.uid, fb);
    auto it = ctx->scale_tables.find(key);
    if (it != ctx->scale_tables.end()) { it->second->last_use = ++ctx->next_uid; return it->second; }
    if (ctx->scale_tables.size() >= 768) {
        // bounded: the least recently used tables nobody references go -- a sixteenth of the cache at a time, behind ONE drain of the
        // device (any lane may still hold kernels that read them).  One table per miss meant a device-wide wait per miss once the cache
        // was full: a process that had worked through a few dozen cascades (bench.py's secondary table) met it on every new factor.
        std::vector<std::pair<uint64_t, std::pair<uint64_t, uint64_t>>> idle;       // (last use, key)
        for (auto &kv : ctx->scale_tables) if (kv.second->refs == 0) idle.push_back({kv.second->last_use, kv.first});
        const size_t drop = std::min<size_t>(idle.size(), 48);
        if (drop) {
            std::partial_sort(idle.begin(), idle.begin() + drop, idle.end());
            (void)hipDeviceSynchronize();
            for (size_t i = 0; i < drop; i++) { auto it = ctx->scale_tables.find(idle[i].second); delete it->second; ctx->scale_tables.erase(it); }
        }
    }
    std::unique_ptr<ScaleTable> t(new ScaleTable());
    build_scale_table(c, factor, *t);
    const size_t bytes = t->host.size() * sizeof(TStumpRec), lbytes = t->lhost.size() * sizeof(LStumpRec);
    if (t->dev.ensure(bytes + lbytes ? bytes + lbytes : 8)) { ctx->set_error("hipMalloc failed for a stump table"); return nullptr; }
    if (bytes && hipMemcpy(t->dev.p, t->host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { ctx->set_error("hipMemcpy failed for a stump table"); return nullptr; }
    if (lbytes && hipMemcpy((unsigned char *)t->dev.p + bytes, t->lhost.data(), lbytes, hipMemcpyHostToDevice) != hipSuccess) { ctx->set_error("hipMemcpy failed for a stump table"); return nullptr; }
    t->d_lrecs = lbytes ? (const LStumpRec *)((const unsigned char *)t->dev.p + bytes) : nullptr;
    if (!t->ghost.empty()) {             // general cascade: nodes | leaf values | first-node indices in one block
        const size_t nb = (t->ghost.size() * sizeof(GNodeRec) + 255) & ~(size_t)255, ab = (t->galpha.size() * sizeof(float) + 255) & ~(size_t)255;
        const size_t cb = t->gcls_first.size() * sizeof(int);
        std::vector<unsigned char> blob(nb + ab + cb);
        memcpy(blob.data(), t->ghost.data(), t->ghost.size() * sizeof(GNodeRec));
        memcpy(blob.data() + nb, t->galpha.data(), t->galpha.size() * sizeof(float));
        memcpy(blob.data() + nb + ab, t->gcls_first.data(), cb);
        if (t->gdev.ensure(blob.size())) { ctx->set_error("hipMalloc failed for a node table"); return nullptr; }
        if (hipMemcpy(t->gdev.p, blob.data(), blob.size(), hipMemcpyHostToDevice) != hipSuccess) { ctx->set_error("hipMemcpy failed for a node table"); return nullptr; }
        t->d_grecs = (const GNodeRec *)t->gdev.p;
        t->d_galpha = (const float *)((const unsigned char *)t->gdev.p + nb);
        t->d_gcls_first = (const int *)((const unsigned char *)t->gdev.p + nb + ab);
    }
    t->last_use = ++ctx->next_uid;
    ScaleTable *raw = t.release();
    ctx->scale_tables[key] = raw;
    return raw;
}

void free_scale_tables(nvca_ctx *ctx)
{
    for (auto &kv : ctx->scale_tables) delete kv.second;
    ctx->scale_tables.clear();
}

} // namespace nvca
