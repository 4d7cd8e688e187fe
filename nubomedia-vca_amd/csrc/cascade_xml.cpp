// cascade_xml.cpp -- reader of OpenCV's old-format Haar cascade XML
// (type_id="opencv-haar-classifier"), the format of every cascade file the
// reference loads: FACE/kmsfacedetect.cpp:40,162-177, EYE/kmseyedetect.cpp:27-29,
// NOSE/kmsnosedetect.cpp:31-32, MOUTH/kmsmouthdetect.cpp:37-38, EAR/kmseardetect.cpp:29-31.
// Mirrors what OpenCV 2.4's icvReadHaarClassifier stores: numbers are parsed as
// double and kept as float; a *_val leaf becomes alpha[last++] with child index -last.
#include "nvca_internal.h"
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <cfloat>

namespace nvca {
namespace {

struct XNode {
    std::string name, type_id, text;
    std::vector<std::unique_ptr<XNode>> kids;
    const XNode *child(const char *n) const {
        for (auto &k : kids) if (k->name == n) return k.get();
        return nullptr;
    }
};

struct XParser {
    const char *p, *end; std::string err;
    bool fail(const std::string &m) { if (err.empty()) err = m; return false; }
    void skip_ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    bool starts(const char *s) { size_t n = strlen(s); return (size_t)(end - p) >= n && !memcmp(p, s, n); }
    bool skip_misc() {           // comments, PIs, doctype
        for (;;) {
            skip_ws();
            if (starts("<!--")) { const char *e = (const char *)memmem(p, end - p, "-->", 3); if (!e) return fail("unterminated comment"); p = e + 3; }
            else if (starts("<?")) { const char *e = (const char *)memmem(p, end - p, "?>", 2); if (!e) return fail("unterminated PI"); p = e + 2; }
            else if (starts("<!")) { const char *e = (const char *)memchr(p, '>', end - p); if (!e) return fail("unterminated decl"); p = e + 1; }
            else return true;
        }
    }
    bool parse_elem(XNode &n, int depth) {
        if (depth > 64) return fail("XML too deep");
        if (p >= end || *p != '<') return fail("expected '<'");
        ++p;
        const char *s = p;
        while (p < end && *p != ' ' && *p != '>' && *p != '/' && *p != '\n' && *p != '\t' && *p != '\r') ++p;
        n.name.assign(s, p);
        if (n.name.empty()) return fail("empty tag name");
        // attributes
        for (;;) {
            skip_ws();
            if (p >= end) return fail("unterminated tag");
            if (*p == '/') { if (p + 1 < end && p[1] == '>') { p += 2; return true; } return fail("bad '/'"); }
            if (*p == '>') { ++p; break; }
            const char *as = p;
            while (p < end && *p != '=' && *p != '>' && *p != ' ') ++p;
            std::string an(as, p);
            skip_ws();
            if (p >= end || *p != '=') return fail("attribute without value");
            ++p; skip_ws();
            if (p >= end || (*p != '"' && *p != '\'')) return fail("attribute value not quoted");
            char q = *p++;
            const char *vs = p;
            while (p < end && *p != q) ++p;
            if (p >= end) return fail("unterminated attribute");
            if (an == "type_id") n.type_id.assign(vs, p);
            ++p;
        }
        // content
        for (;;) {
            const char *ts = p;
            while (p < end && *p != '<') ++p;
            n.text.append(ts, p);
            if (p >= end) return fail("unterminated element <" + n.name + ">");
            if (starts("<!--")) { const char *e = (const char *)memmem(p, end - p, "-->", 3); if (!e) return fail("unterminated comment"); p = e + 3; continue; }
            if (starts("</")) {
                p += 2;
                const char *cs = p;
                while (p < end && *p != '>') ++p;
                if (p >= end) return fail("unterminated close tag");
                std::string cn(cs, p);
                while (!cn.empty() && (cn.back() == ' ' || cn.back() == '\n')) cn.pop_back();
                ++p;
                if (cn != n.name) return fail("mismatched </" + cn + "> for <" + n.name + ">");
                return true;
            }
            std::unique_ptr<XNode> k(new XNode());
            if (!parse_elem(*k, depth + 1)) return false;
            n.kids.push_back(std::move(k));
        }
    }
};

bool to_double(const std::string &s, double &v) {
    const char *b = s.c_str(); char *e = nullptr;
    v = strtod(b, &e);
    if (e == b) return false;
    while (*e == ' ' || *e == '\n' || *e == '\r' || *e == '\t') ++e;
    return *e == 0;
}
bool to_int(const std::string &s, int &v) {
    const char *b = s.c_str(); char *e = nullptr;
    long l = strtol(b, &e, 10);
    if (e == b) return false;
    while (*e == ' ' || *e == '\n' || *e == '\r' || *e == '\t') ++e;
    if (*e) return false;
    v = (int)l; return true;
}

} // namespace

int parse_cascade_xml(const char *text, size_t len, Cascade &out, std::string &err)
{
    XParser xp{text, text + len, {}};
    if (!xp.skip_misc()) { err = xp.err; return NVCA_ERR_PARSE; }
    XNode root;
    if (!xp.parse_elem(root, 0)) { err = xp.err; return NVCA_ERR_PARSE; }
    if (root.name != "opencv_storage") { err = "root element is not <opencv_storage>"; return NVCA_ERR_PARSE; }
    const XNode *cn = nullptr;
    for (auto &k : root.kids) if (k->type_id == "opencv-haar-classifier") { cn = k.get(); break; }
    if (!cn) { err = "no opencv-haar-classifier node (new-format cascades are not Haar old-format)"; return NVCA_ERR_PARSE; }

    const XNode *size = cn->child("size"), *stages = cn->child("stages");
    if (!size || !stages) { err = "missing <size> or <stages>"; return NVCA_ERR_PARSE; }
    if (sscanf(size->text.c_str(), "%d %d", &out.ow, &out.oh) != 2 || out.ow <= 2 || out.oh <= 2 || out.ow > 1024 || out.oh > 1024) {
        err = "bad <size>"; return NVCA_ERR_PARSE;             // stock cascades are 18..45 pixels a side; the bound keeps every later sum in range
    }
    out.stages.clear(); out.cls.clear(); out.nodes.clear(); out.alpha.clear();
    out.stump_based = true; out.has_tilted = false;
    int si = 0;
    for (auto &st : stages->kids) {
        const XNode *trees = st->child("trees"), *sthr = st->child("stage_threshold");
        const XNode *par = st->child("parent"), *nxt = st->child("next");
        if (!trees || !sthr || !par || !nxt) { err = "stage without trees/stage_threshold/parent/next"; return NVCA_ERR_PARSE; }
        double d; int parent, next;
        if (!to_double(sthr->text, d) || !to_int(par->text, parent) || !to_int(nxt->text, next)) { err = "bad stage scalar"; return NVCA_ERR_PARSE; }
        if (parent != si - 1 || next != -1) { err = "tree-structured stage graph is not supported"; return NVCA_ERR_UNSUPPORTED; }
        HaarStage hs; hs.first_cls = (int)out.cls.size(); hs.ncls = (int)trees->kids.size(); hs.threshold = (float)d;
        if (hs.ncls <= 0) { err = "empty stage"; return NVCA_ERR_PARSE; }
        for (auto &tree : trees->kids) {
            HaarClassifier hc; hc.first_node = (int)out.nodes.size(); hc.nnodes = (int)tree->kids.size();
            hc.first_alpha = (int)out.alpha.size();
            if (hc.nnodes <= 0) { err = "empty tree"; return NVCA_ERR_PARSE; }
            if (hc.nnodes != 1) out.stump_based = false;
            int last = 0;
            for (auto &nd : tree->kids) {
                HaarNode hn; memset(&hn, 0, sizeof(hn));
                const XNode *feat = nd->child("feature"), *thr = nd->child("threshold");
                if (!feat || !thr) { err = "node without feature/threshold"; return NVCA_ERR_PARSE; }
                const XNode *rects = feat->child("rects"), *tilted = feat->child("tilted");
                if (!rects || !tilted) { err = "feature without rects/tilted"; return NVCA_ERR_PARSE; }
                if (rects->kids.size() < 2 || rects->kids.size() > 3) { err = "feature must have 2 or 3 rects"; return NVCA_ERR_PARSE; }
                for (size_t k = 0; k < rects->kids.size(); k++) {
                    double wt; int r[4];
                    if (sscanf(rects->kids[k]->text.c_str(), "%d %d %d %d %lf", &r[0], &r[1], &r[2], &r[3], &wt) != 5) {
                        err = "bad rect"; return NVCA_ERR_PARSE;
                    }
                    for (int q = 0; q < 4; q++) hn.rect[k][q] = r[q];
                    hn.weight[k] = (float)wt;
                    if (r[0] < 0 || r[1] < 0 || r[2] <= 0 || r[3] <= 0 || r[0] > out.ow || r[2] > out.ow || r[1] > out.oh || r[3] > out.oh) { err = "rect outside the window"; return NVCA_ERR_PARSE; }
                }
                if (!to_int(tilted->text, hn.tilted)) { err = "bad <tilted>"; return NVCA_ERR_PARSE; }
                hn.tilted = hn.tilted != 0;
                for (size_t k = 0; k < rects->kids.size(); k++) {
                    const int *r = hn.rect[k];
                    // upright: x .. x+w, y .. y+h.  tilted: the rectangle rotated about (x, y) reads the tilted integral at
                    // columns x-h .. x+w and rows y .. y+w+h (cvSetImagesForHaarClassifierCascade)
                    const bool inside = !hn.tilted ? (r[0] + r[2] <= out.ow && r[1] + r[3] <= out.oh)
                                                   : (r[0] - r[3] >= 0 && r[0] + r[2] <= out.ow && r[1] + r[2] + r[3] <= out.oh);
                    if (!inside) { err = hn.tilted ? "tilted rect outside the window" : "rect outside the window"; return NVCA_ERR_PARSE; }
                }
                if (hn.tilted) out.has_tilted = true;
                // icvCreateHidHaarClassifierCascade: rect[2] is dropped when its weight or size is zero
                hn.nrect = (fabs(hn.weight[2]) < DBL_EPSILON || hn.rect[2][2] == 0 || hn.rect[2][3] == 0) ? 2 : 3;
                if (!to_double(thr->text, d)) { err = "bad <threshold>"; return NVCA_ERR_PARSE; }
                hn.threshold = (float)d;
                for (int side = 0; side < 2; side++) {
                    const XNode *cn2 = nd->child(side == 0 ? "left_node" : "right_node");
                    int &dst = side == 0 ? hn.left : hn.right;
                    if (cn2) {
                        // a child comes after its parent (haartraining writes trees in that order): every walk from the root ends,
                        // on the device too -- an index at or before the node itself would make the evaluator loop for ever
                        const int self = (int)out.nodes.size() - hc.first_node;
                        if (!to_int(cn2->text, dst) || dst <= self || dst >= hc.nnodes) { err = "bad child node index"; return NVCA_ERR_PARSE; }
                    } else {
                        const XNode *cv = nd->child(side == 0 ? "left_val" : "right_val");
                        if (!cv || !to_double(cv->text, d)) { err = "node without left/right value"; return NVCA_ERR_PARSE; }
                        dst = -last; out.alpha.push_back((float)d); last++;
                    }
                }
                out.nodes.push_back(hn);
            }
            if (last != hc.nnodes + 1) { err = "tree leaf count mismatch"; return NVCA_ERR_PARSE; }
            out.cls.push_back(hc);
        }
        out.stages.push_back(hs);
        si++;
    }
    if (out.stages.empty()) { err = "cascade has no stages"; return NVCA_ERR_PARSE; }
    return NVCA_OK;
}

} // namespace nvca
