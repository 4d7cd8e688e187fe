// host_logic.cpp -- the small sequential pieces of the path that stay on the
// host exactly as in the reference / OpenCV:
//   cv::groupRectangles + cv::partition   (cascadedetect.cpp, operations.hpp; inside
//                                          detectMultiScale, FACE/kmsfacedetect.cpp:809)
//   Faces::track_faces and helpers        (FACE/Faces.cpp:78-188, FACE/BaseFace.cpp:97-101)
//   __join_objects / __merge / calc_dist  (TRK/gstnubotracker.cpp:119-200)
// They are O(#boxes) integer code that decides the emitted boxes.
#include "host_logic.h"
#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <climits>

namespace nvca {

static inline int cv_round(double v) { return (int)lrint(v); }

// ------------------------------------------------------------ groupRectangles
namespace {
struct SimilarRects {
    double eps;
    bool operator()(const nvca_rect &r1, const nvca_rect &r2) const {
        double delta = eps * (std::min(r1.w, r2.w) + std::min(r1.h, r2.h)) * 0.5;
        return std::abs(r1.x - r2.x) <= delta && std::abs(r1.y - r2.y) <= delta &&
               std::abs(r1.x + r1.w - r2.x - r2.w) <= delta && std::abs(r1.y + r1.h - r2.y - r2.h) <= delta;
    }
};

// cv::partition: the classes are the connected components of the (symmetric) predicate, numbered in order of their first
// member.  OpenCV tests every ordered pair; the components -- all the result depends on -- come out the same from the pairs
// that CAN satisfy SimilarRects: |x1 - x2| <= delta with delta = eps (min w + min h) / 2 <= eps (w1 + h1) / 2, so a rectangle's
// partners lie within that distance of it in x.  Rectangles sorted by x, a window per rectangle, pairs already in one class
// skipped: a FIND_BIGGEST search's few hundred candidates per ladder step cost a few thousand tests instead of N^2.
int partition(const std::vector<nvca_rect> &v, std::vector<int> &labels, SimilarRects pred)
{
    const int N = (int)v.size();
    std::vector<int> parent(N, -1), rank(N, 0);
    auto find = [&](int i) { int r = i; while (parent[r] >= 0) r = parent[r]; for (int k = i, p; (p = parent[k]) >= 0; k = p) parent[k] = r; return r; };
    std::vector<int> byx(N);
    for (int i = 0; i < N; i++) byx[i] = i;
    std::sort(byx.begin(), byx.end(), [&](int a, int b) { return v[a].x < v[b].x; });
    for (int a = 0; a < N; a++) {
        const int i = byx[a];
        const double reach = pred.eps * (v[i].w + v[i].h) * 0.5;             // no partner of i is further away in x
        for (int b = a + 1; b < N; b++) {                                   // every pair is met once, from its left member: delta <= the reach of BOTH
            const int j = byx[b];
            if ((double)(v[j].x - v[i].x) > reach) break;                   // sorted by x: nobody further right is within reach either
            int root = find(i), root2 = find(j);
            if (root2 == root || !pred(v[i], v[j])) continue;
            if (rank[root] > rank[root2]) parent[root2] = root;
            else { parent[root] = root2; rank[root2] += rank[root] == rank[root2]; }
        }
    }
    labels.assign(N, 0);
    std::vector<int> cls(N, -1);
    int nclasses = 0;
    for (int i = 0; i < N; i++) {
        int root = find(i);
        if (cls[root] < 0) cls[root] = nclasses++;
        labels[i] = cls[root];
    }
    return nclasses;
}
} // namespace

void group_rectangles(std::vector<nvca_rect> &rects, int groupThreshold, double eps, std::vector<int> *weights)
{
    if (groupThreshold <= 0 || rects.empty()) {
        if (weights) weights->assign(rects.size(), 1);
        return;
    }
    std::vector<int> labels;
    const int nclasses = partition(rects, labels, SimilarRects{eps});
    std::vector<nvca_rect> rr(nclasses, nvca_rect{0, 0, 0, 0});
    std::vector<int> rw(nclasses, 0);
    for (size_t i = 0; i < rects.size(); i++) {
        nvca_rect &a = rr[labels[i]];
        a.x += rects[i].x; a.y += rects[i].y; a.w += rects[i].w; a.h += rects[i].h;
        rw[labels[i]]++;
    }
    for (int i = 0; i < nclasses; i++) {
        const float s = 1.f / rw[i];
        nvca_rect r = rr[i];
        rr[i] = nvca_rect{cv_round(r.x * s), cv_round(r.y * s), cv_round(r.w * s), cv_round(r.h * s)};
    }
    rects.clear();
    if (weights) weights->clear();
    for (int i = 0; i < nclasses; i++) {
        const nvca_rect r1 = rr[i];
        const int n1 = rw[i];
        if (n1 <= groupThreshold) continue;
        int j;
        for (j = 0; j < nclasses; j++) {
            const int n2 = rw[j];
            if (j == i || n2 <= groupThreshold) continue;
            const nvca_rect r2 = rr[j];
            const int dx = cv_round(r2.w * eps), dy = cv_round(r2.h * eps);
            if (r1.x >= r2.x - dx && r1.y >= r2.y - dy && r1.x + r1.w <= r2.x + r2.w + dx &&
                r1.y + r1.h <= r2.y + r2.h + dy && (n2 > std::max(3, n1) || n1 < 3))
                break;
        }
        if (j == nclasses) {
            rects.push_back(r1);
            if (weights) weights->push_back(n1);
        }
    }
}

// ------------------------------------------------------------ Faces
namespace {
inline int area(const nvca_rect &r) { return r.w * r.h; }
inline int centre_distance(const nvca_rect &a, const nvca_rect &b)
{   // Faces::calc_distance over BaseFace::calc_center: truncated Euclidean distance
    const int ax = a.x + a.w / 2, ay = a.y + a.h / 2, bx = b.x + b.w / 2, by = b.y + b.h / 2;
    return (int)std::sqrt(std::pow((double)(bx - ax), 2) + std::pow((double)(by - ay), 2));
}
inline int distance_limit(int s1, int s2) { const int big = std::max(s1, s2); return big > 5000 ? 8 : (big > 2500 ? 5 : 3); }
inline int diff_area_pct(int s1, int s2) { return (std::abs(s1 - s2) * 100) / s2; }
} // namespace

void Faces::track(const std::vector<nvca_rect> &current, int track_threshold)
{
    std::vector<nvca_rect> cf(current);
    std::vector<TrackedFace> next;
    for (const TrackedFace &f : faces) {
        int best = track_threshold, pos = -1;
        for (size_t i = 0; i < cf.size(); i++) {
            const int d = centre_distance(cf[i], f.box);
            if (best > d) { pos = (int)i; best = d; }
        }
        if (pos < 0) continue;                         // unmatched old faces are dropped
        const nvca_rect &n = cf[pos];
        const int d = centre_distance(f.box, n);
        if (distance_limit(area(f.box), area(n)) < d) next.push_back(TrackedFace{n, f.id});
        else if (15 < diff_area_pct(area(f.box), area(n))) next.push_back(TrackedFace{nvca_rect{f.box.x, f.box.y, n.w, n.h}, f.id});
        else next.push_back(f);
        cf.erase(cf.begin() + pos);
    }
    for (const nvca_rect &r : cf) next.push_back(TrackedFace{r, next_id++});
    faces.swap(next);
}

// ------------------------------------------------------------ tracker boxes
namespace {
inline float trk_dist(const nvca_rect &a, const nvca_rect &b)
{
    const int c1x = a.x + a.w / 2, c1y = a.y + a.h / 2, c2x = b.x + b.w / 2, c2y = b.y + b.h / 2;
    return (float)std::sqrt((double)((c1x - c2x) * (c1x - c2x) + (c1y - c2y) * (c1y - c2y)));
}
inline bool inside(int px, int py, const nvca_rect &r) { return r.x <= px && px < r.x + r.w && r.y <= py && py < r.y + r.h; }
nvca_rect merge(const nvca_rect &r1, const nvca_rect &r2)
{
    if (inside(r2.x, r2.y, r1) && inside(r2.x + r2.w, r2.y + r2.h, r1)) return r1;
    if (inside(r1.x, r1.y, r2) && inside(r1.x + r1.w, r1.y + r1.h, r2)) return r2;
    const int tx = std::min(r1.x, r2.x), ty = std::min(r1.y, r2.y);
    const int bx = std::max(r1.x + r1.w, r2.x + r2.w), by = std::max(r1.y + r1.h, r2.y + r2.h);
    nvca_rect r;            // cv::Rect(Point, Point)
    r.x = std::min(tx, bx); r.y = std::min(ty, by);
    r.w = std::max(tx, bx) - r.x; r.h = std::max(ty, by) - r.y;
    return r;
}
} // namespace

void join_objects(std::vector<nvca_rect> &sb, int min_area, long max_area, int distance)
{
    for (int a = (int)sb.size() - 1; a >= 0; a--) {
        if (area(sb[a]) > min_area && area(sb[a]) < max_area) {
            for (int b = a - 1; b >= 0; b--) {
                if (area(sb[b]) > min_area && area(sb[b]) < max_area)
                    if ((float)distance > trk_dist(sb[a], sb[b])) {
                        sb[b] = merge(sb[a], sb[b]);
                        sb.erase(sb.begin() + a);
                        break;
                    }
            }
        } else
            sb.erase(sb.begin() + a);
    }
}

// view-* outlines on a host frame: the bounding box of every shape is walked, the last shape that covers a pixel colours it
void draw_shapes_host(uint8_t *data, int w, int h, int stride, int channels, const nvca_shape *shapes, int n)
{
    for (int i = 0; i < n; i++) {
        const nvca_shape &sh = shapes[i];
        int x0, y0, x1, y1;
        if (sh.kind == NVCA_SHAPE_RING4) { const int r = (sh.w > 0 ? sh.w : 0) + 2; x0 = sh.x - r; x1 = sh.x + r; y0 = sh.y - r; y1 = sh.y + r; }
        else { x0 = std::min(sh.x, sh.x + sh.w) - 1; x1 = std::max(sh.x, sh.x + sh.w) + 1; y0 = std::min(sh.y, sh.y + sh.h) - 1; y1 = std::max(sh.y, sh.y + sh.h) + 1; }
        x0 = std::max(x0, 0); y0 = std::max(y0, 0); x1 = std::min(x1, w - 1); y1 = std::min(y1, h - 1);
        for (int y = y0; y <= y1; y++)
            for (int x = x0; x <= x1; x++)
                if (shape_covers(sh, x, y)) {
                    uint8_t *p = data + (size_t)y * stride + (size_t)x * channels;
                    for (int k = 0; k < channels; k++) p[k] = sh.bgra[k];
                }
    }
}

// ------------------------------------------------------------ part detectors: merging heuristics
// (moved here from parts.cpp: pure functions of box lists, O(#faces); std::vector idioms of the reference that rely on
// libstdc++ behaviour -- erase through a reverse iterator, erase(end()-i) inside a counting loop -- are written out as the
// index operations they perform)
typedef std::vector<nvca_rect> RectV;
static inline int cv_round_sat(double v)
{
    if (!(v > -2147483648.5 && v < 2147483647.5)) return INT_MIN;
    return (int)lrint(v);
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
        r.x = cv_round_sat((face.x + r.x) * scale); r.y = cv_round_sat((face.y + r.y) * scale);
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


// ------------------------------------------------------------ image-to-overlay on a host frame
// kms_face_detect_display_detections_overlay_img (FACE/kmsfacedetect.cpp:427-502) for every box in order; the arithmetic
// is nvca_internal.h's resize_sample_cn / overlay_pixel, shared with the kernel
void overlay_blend_host(uint8_t *frame, int W, int H, int stride, const nvca_rect *boxes, int n, const nvca_overlay &ov)
{
    if (ov.height_percent == 0 || ov.width_percent == 0) return;           // :436-439
    const uint8_t *img = (const uint8_t *)ov.data;
    for (int b = 0; b < n; b++) {
        const OverlayPlace p = overlay_place(boxes[b], ov);
        if (p.w <= 0 || p.h <= 0) continue;
        ResizeTab tab;
        build_resize_tab(ov.width, ov.height, p.w, p.h, tab);
        for (int h = 0; h < p.h; h++) {
            if (h + p.y < 0 || h + p.y >= H) continue;
            for (int w = 0; w < p.w; w++) {
                if (w + p.x < 0 || w + p.x >= W) continue;
                int v[4] = {0, 0, 0, 0};
                for (int k = 0; k < ov.channels; k++)
                    v[k] = resize_sample_cn(img, ov.height, ov.stride, ov.channels, tab.mode, tab.xofs.data(), tab.ialpha.data(), tab.yofs.data(), tab.ibeta.data(), tab.xmax, w, h, k);
                overlay_pixel(frame + (size_t)(h + p.y) * stride + (size_t)(w + p.x) * 3, v, ov.channels);
            }
        }
    }
}

} // namespace nvca
