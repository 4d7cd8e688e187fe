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

// cvSetImagesForHaarClassifierCascade for one scale; offsets use `pitch` (elements).
void build_scale_tables(const Cascade &c, double factor, int pitch, ScaleRec &sr, StumpRec *out)
{
    int ex = cv_round(factor), ey = ex;
    int ew = cv_round((c.ow - 2) * factor), eh = cv_round((c.oh - 2) * factor);
    double weight_scale = 1. / (ew * eh);
    sr.winw = cv_round(c.ow * factor); sr.winh = cv_round(c.oh * factor);
    sr.inv_area = weight_scale; sr.factor = factor;
    sr.eq[0] = ey * pitch + ex;          sr.eq[1] = ey * pitch + ex + ew;
    sr.eq[2] = (ey + eh) * pitch + ex;   sr.eq[3] = (ey + eh) * pitch + ex + ew;
    size_t k = 0;
    for (const HaarClassifier &hc : c.cls) {
        const HaarNode &n = c.nodes[hc.first_node];     // stump
        StumpRec &r = out[k++];
        memset(&r, 0, sizeof(r));
        double sum0 = 0, area0 = 0;
        for (int q = 0; q < n.nrect; q++) {
            int tx = cv_round(n.rect[q][0] * factor), tw = cv_round(n.rect[q][2] * factor);
            int ty = cv_round(n.rect[q][1] * factor), th = cv_round(n.rect[q][3] * factor);
            double correction_ratio = weight_scale;       // upright feature, CV_ADJUST_WEIGHTS 0
            r.p[q][0] = ty * pitch + tx;        r.p[q][1] = ty * pitch + tx + tw;
            r.p[q][2] = (ty + th) * pitch + tx; r.p[q][3] = (ty + th) * pitch + tx + tw;
            r.w[q] = (float)(n.weight[q] * correction_ratio);
            if (q == 0) area0 = tw * th;
            else {
                float t = r.w[q] * tw;      // float * int -> float, evaluated left to right
                t = t * th;
                sum0 += t;
            }
        }
        r.w[0] = (float)(-sum0 / area0);
        r.thr = n.threshold;
        r.a0 = c.alpha[hc.first_alpha]; r.a1 = c.alpha[hc.first_alpha + 1];
        r.nrect = n.nrect;
    }
}

void build_stage_recs(const Cascade &c, std::vector<StageRec> &out)
{
    out.clear();
    for (const HaarStage &s : c.stages) {
        StageRec r; r.first = s.first_cls; r.count = s.ncls;
        r.thr = s.threshold - 0.0001f;      // icv_stage_threshold_bias, float arithmetic
        r.two_rects = 1;
        for (int j = 0; j < s.ncls; j++)
            if (c.nodes[c.cls[s.first_cls + j].first_node].nrect == 3) r.two_rects = 0;
        out.push_back(r);
    }
}

int DetectPlan::build_scale_cascade(const Cascade &c, int cols, int rows, int pitch, double scaleFactor,
                                    int minw, int minh, int maxw, int maxh, std::string &err)
{
    if (!c.stump_based) { err = "tree weak classifiers are not supported by the device evaluator yet"; return NVCA_ERR_UNSUPPORTED; }
    std::vector<double> factors;
    scale_grid(c.ow, c.oh, cols, rows, scaleFactor, minw, minh, maxw, maxh, false, factors);
    nstumps = (int)c.cls.size();
    scales.clear(); strips.clear(); pos.clear();
    stumps.assign(factors.size() * (size_t)nstumps, StumpRec());
    build_stage_recs(c, stages);
    for (size_t s = 0; s < factors.size(); s++) {
        double factor = factors[s];
        const double ystep = std::max(2., factor);
        ScaleRec sr; memset(&sr, 0, sizeof(sr));
        build_scale_tables(c, factor, pitch, sr, &stumps[s * (size_t)nstumps]);
        sr.stump_off = (int)(s * (size_t)nstumps);
        sr.startX = sr.startY = 0;
        sr.endX = cv_round((cols - sr.winw) / ystep);
        sr.endY = cv_round((rows - sr.winh) / ystep);
        if (sr.endX > 8191 || sr.endY > 8191) { err = "image too large for the candidate key"; return NVCA_ERR_ARG; }
        sr.xpos_off = (int)pos.size();
        for (int ix = 0; ix < std::max(sr.endX, 0); ix++) pos.push_back(cv_round(ix * ystep));
        sr.ypos_off = (int)pos.size();
        for (int iy = 0; iy < std::max(sr.endY, 0); iy++) pos.push_back(cv_round(iy * ystep));
        scales.push_back(sr);
        if (sr.endX <= 0 || sr.endY <= 0) continue;
        // cvRunHaarClassifierCascadeSum's own bound: windows must satisfy x + w < cols + 1
        // (always true inside the loop limits; checked so the kernel needs no test)
        if (pos[sr.xpos_off + sr.endX - 1] + sr.winw >= cols + 1 || pos[sr.ypos_off + sr.endY - 1] + sr.winh >= rows + 1) {
            err = "scan grid leaves the image"; return NVCA_ERR_ARG;
        }
        int rows_per = std::max(1, std::min(kStripMaxWin / sr.endX, 64));
        if (sr.endX > kStripMaxWin) { err = "row longer than a strip"; return NVCA_ERR_ARG; }
        for (int iy = 0; iy < sr.endY; iy += rows_per) {
            StripRec st; st.scale = (int)s; st.iy0 = iy; st.nrows = std::min(rows_per, sr.endY - iy); st.pad = 0;
            strips.push_back(st);
        }
    }
    // Dispatch order.  Workgroups are dealt round-robin over the 8 XCDs (b % 8 shares an XCD), each with
    // its own 4 MiB L2.  Give every XCD one contiguous run of strips (scale-major, then y) of about equal
    // window count, so that the integral rows an XCD gathers from stay resident in its L2
    // (placement only changes speed, never results).
    {
        long long total = 0;
        for (const StripRec &st : strips) total += (long long)st.nrows * scales[st.scale].endX;
        std::vector<int> start(9, (int)strips.size());
        start[0] = 0;
        long long acc = 0; int k = 1;
        for (size_t i = 0; i < strips.size() && k < 8; i++) {
            acc += (long long)strips[i].nrows * scales[strips[i].scale].endX;
            while (k < 8 && acc * 8 >= total * k) { start[k] = (int)i + 1; k++; }
        }
        for (; k < 8; k++) start[k] = (int)strips.size();
        int maxlen = 0;
        for (int x = 0; x < 8; x++) maxlen = std::max(maxlen, start[x + 1] - start[x]);
        blocks_per_frame = 8 * maxlen;
        order.assign(blocks_per_frame, -1);
        for (int x = 0; x < 8; x++)
            for (int j = 0; j < start[x + 1] - start[x]; j++) order[j * 8 + x] = start[x] + j;
    }
    return NVCA_OK;
}

} // namespace nvca
