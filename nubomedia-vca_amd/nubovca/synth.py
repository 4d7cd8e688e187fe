"""Synthetic inputs for the Haar hot path: a seeded stump cascade written in
OpenCV's old-format XML (the reference hard-codes
/usr/share/opencv/haarcascades/haarcascade_frontalface_alt.xml,
FACE/kmsfacedetect.cpp:40, which is not available offline) and seeded frames.

The cascade has the *shape* of haarcascade_frontalface_alt (20x20 window,
22 stages, 2135 stumps, first stages 3/16/21) and is constructed so that
  * i.i.d. noise windows are rejected with ~50 % probability per stage
    (the per-stage false-alarm rate haartraining aims for), and
  * windows resembling TEMPLATE (a 20x20 face-like pattern) pass every stage.
It is a workload generator, not a face model.
"""
import numpy as np

FRONTALFACE_ALT_STAGES = [3, 16, 21, 39, 33, 44, 50, 51, 56, 71, 80, 103, 111, 102,
                          135, 137, 140, 160, 177, 182, 211, 213]
WIN = 20


def template(win=WIN):
    """20x20 face-like luminance pattern, float64 in [0,255]."""
    y, x = np.mgrid[0:win, 0:win].astype(np.float64)
    s = win / 20.0

    def blob(cx, cy, sx, sy, amp):
        return amp * np.exp(-(((x - cx * s) / (sx * s)) ** 2 + ((y - cy * s) / (sy * s)) ** 2))

    t = np.full((win, win), 150.0)
    t += blob(9.5, 9.5, 9, 11, 40)           # bright face oval
    t += blob(5.5, 7.0, 2.2, 1.4, -95)       # left eye
    t += blob(13.5, 7.0, 2.2, 1.4, -95)      # right eye
    t += blob(5.5, 4.6, 2.8, 0.9, -45)       # brows
    t += blob(13.5, 4.6, 2.8, 0.9, -45)
    t += blob(9.5, 10.0, 1.3, 3.0, 45)       # nose bridge
    t += blob(9.5, 15.2, 3.4, 1.2, -80)      # mouth
    t += blob(9.5, 0.5, 9, 2.0, -60)         # hair line
    return np.clip(t, 0, 255)


def _rand_feature(rng, win):
    """One Haar-like feature inside the [1, win-1) interior.
    Returns list of (x, y, w, h, weight) with rect[0] the whole support."""
    lo, hi = 1, win - 1
    kind = rng.integers(0, 5)
    for _ in range(100):
        if kind == 0:      # x2
            w, h = rng.integers(1, 9), rng.integers(1, 17)
            W, H = 2 * w, h
        elif kind == 1:    # y2
            w, h = rng.integers(1, 17), rng.integers(1, 9)
            W, H = w, 2 * h
        elif kind == 2:    # x3
            w, h = rng.integers(1, 6), rng.integers(1, 17)
            W, H = 3 * w, h
        elif kind == 3:    # y3
            w, h = rng.integers(1, 17), rng.integers(1, 6)
            W, H = w, 3 * h
        else:              # x2_y2
            w, h = rng.integers(1, 9), rng.integers(1, 9)
            W, H = 2 * w, 2 * h
        if W * H < 8 or W > hi - lo or H > hi - lo:
            continue
        x = rng.integers(lo, hi - W + 1)
        y = rng.integers(lo, hi - H + 1)
        if kind == 0:
            return [(x, y, W, H, -1.0), (x + w, y, w, h, 2.0)]
        if kind == 1:
            return [(x, y, W, H, -1.0), (x, y + h, w, h, 2.0)]
        if kind == 2:
            return [(x, y, W, H, -1.0), (x + w, y, w, h, 3.0)]
        if kind == 3:
            return [(x, y, W, H, -1.0), (x, y + h, w, h, 3.0)]
        return [(x, y, W, H, -1.0), (x, y, w, h, 2.0), (x + w, y + h, w, h, 2.0)]
    raise RuntimeError("feature sampling failed")


def _pixel_coeffs(feat, win):
    """Per-pixel coefficient image of a feature at scale 1 with OpenCV's weight
    normalisation (rect0 weight recomputed so the feature is zero-mean)."""
    inv_area = 1.0 / ((win - 2) * (win - 2))
    c = np.zeros((win, win))
    w = [r[4] * inv_area for r in feat]
    w[0] = -sum(w[k] * feat[k][2] * feat[k][3] for k in range(1, len(feat))) / (feat[0][2] * feat[0][3])
    for k, (x, y, rw, rh, _) in enumerate(feat):
        c[y:y + rh, x:x + rw] += w[k]
    return c


def _norm_values(coefs, windows):
    """Normalised feature values v = sum(c*img)/std(img[1:-1,1:-1]) for a stack of
    windows [N,win,win] and coefficient images [F,win,win] -> [N,F]."""
    inner = windows[:, 1:-1, 1:-1].reshape(len(windows), -1)
    std = inner.std(axis=1)
    std = np.where(std > 0, std, 1.0)
    vals = windows.reshape(len(windows), -1) @ coefs.reshape(len(coefs), -1).T
    return vals / std[:, None]


def part_template(name, win=WIN):
    """20x20 crop of the 4x up-sampled face TEMPLATE around a face part: what the synthetic
    stand-ins for haarcascade_mcs_{righteye,lefteye,nose,mouth,leftear,rightear} respond to.
    (x0, y0) are in the 80x80 face; a part therefore spans a quarter of the face width."""
    big = template(4 * win)
    x0, y0 = {"righteye": (12, 18), "lefteye": (44, 18), "nose": (28, 32), "mouth": (28, 50),
              "leftear": (0, 28), "rightear": (60, 28)}[name]
    return big[y0:y0 + win, x0:x0 + win].copy()


def make_cascade(seed=2016, stages=None, win=WIN, n_mc=3000, pass_rate=0.5, open_stages=4,
                 agree_lo=0.52, agree_hi=0.60, tmpl=None, calib=None):
    """Returns a dict describing a stump cascade (see cascade_to_xml).

    Every stump separates TEMPLATE's normalised feature value v_T from zero
    (threshold 0.4*v_T, |v_T| >= 1.5 noise sigmas).  The first `open_stages`
    stages get the stage threshold that passes ~pass_rate of noise windows (the
    bulk of the work on any frame, as with a trained cascade); later stages
    additionally demand that agree_lo..agree_hi of the vote weight sides with
    the template, which makes the cascade specific.

    calib: a WindowSample (below) -- windows of real frame content at the scan's own scales.  The stage thresholds are then
    chosen stage by stage on the windows that passed every stage before (what haartraining does with its negatives): each
    stage lets ~pass_rate of what reaches it through, for as many stages as the sample still holds a few hundred windows;
    the stages behind them keep the template-agreement rule.  Features, stump thresholds and votes are the same as without
    calib (same seed, same draws).
    """
    stages = list(FRONTALFACE_ALT_STAGES if stages is None else stages)
    rng = np.random.default_rng(seed)
    T = template(win) if tmpl is None else np.asarray(tmpl, np.float64)
    negs = rng.integers(0, 256, size=(n_mc, win, win)).astype(np.float64)
    Tn = T[None]
    out_stages = []
    for si, ncls in enumerate(stages):
        feats, thr, lv, rv = [], [], [], []
        coefs = []
        while len(feats) < ncls:
            f = _rand_feature(rng, win)
            c = _pixel_coeffs(f, win)
            vT = _norm_values(c[None], Tn)[0, 0]
            sf = np.sqrt((c ** 2).sum())             # std of v on unit-variance noise
            if abs(vT) < 1.5 * sf:
                continue                             # not discriminative for the template
            t = 0.4 * vT
            a = rng.uniform(0.4, 1.0)
            feats.append(f); coefs.append(c); thr.append(t)
            if vT >= t:      # template on the "right" (>= t) side
                lv.append(-a); rv.append(a)
            else:
                lv.append(a); rv.append(-a)
        coefs = np.stack(coefs)
        s_T = sum(abs(a) for a in rv)
        calibrated = False
        if calib is not None and calib.alive_count() >= 256:
            # the votes are float32 in the file and in every evaluator: calibrate on exactly those
            votes = calib.stage_votes(feats, thr, np.float32(lv).astype(np.float64), np.float32(rv).astype(np.float64))
            cand = np.unique(votes)
            rates = np.array([(votes >= cnd).mean() for cnd in cand])
            st_thr = cand[np.argmin(np.abs(rates - pass_rate))] - 1e-3
            st_thr = min(st_thr, 0.6 * s_T)
            calib.keep(votes >= float(np.float32(st_thr)) - 1e-4)
            calibrated = True
        else:
            v = _norm_values(coefs, negs)
            votes = np.where(v >= np.array(thr)[None], np.array(rv)[None], np.array(lv)[None]).sum(axis=1)
            # achievable vote sums are discrete: take the cut whose noise pass rate is closest
            cand = np.unique(votes)
            rates = np.array([(votes >= cnd).mean() for cnd in cand])
            st_thr = cand[np.argmin(np.abs(rates - pass_rate))] - 1e-3
        if not calibrated and (si >= open_stages or calib is not None):
            frac = agree_lo + (agree_hi - agree_lo) * min(1.0, max(0, si - open_stages) / 6.0)
            st_thr = max(st_thr, s_T * (2 * frac - 1))
        st_thr = min(st_thr, 0.6 * s_T)              # the template keeps a wide margin
        out_stages.append(dict(features=feats, thresholds=thr, left=lv, right=rv,
                               stage_threshold=float(st_thr)))
    return dict(name="synthetic_frontalface", size=(win, win), stages=out_stages)


def _f(v):
    return "%.9g" % float(np.float32(v))


def cascade_to_xml(casc):
    """Old-format OpenCV Haar cascade XML (type_id opencv-haar-classifier).

    A stage is either the stump form {features, thresholds, left, right, stage_threshold [, tilted]} (one root node per
    weak classifier; `tilted`: per-feature 0/1) or the general form {trees, stage_threshold}: `trees` is a list of weak
    classifiers, each a list of nodes {feature, tilted, threshold, left, right} whose left / right is ("val", v) for a
    leaf or ("node", i) for the index of a later node of the same tree (icvReadHaarClassifier's <left_node>)."""
    L = ['<?xml version="1.0"?>', "<opencv_storage>",
         '<%s type_id="opencv-haar-classifier">' % casc["name"],
         "  <size>%d %d</size>" % casc["size"], "  <stages>"]
    for si, st in enumerate(casc["stages"]):
        if "trees" in st:
            trees = st["trees"]
        else:
            tl = st.get("tilted", [0] * len(st["features"]))
            trees = [[dict(feature=f, tilted=tl[j], threshold=st["thresholds"][j], left=("val", st["left"][j]),
                           right=("val", st["right"][j]))] for j, f in enumerate(st["features"])]
        L += ["    <_>", "      <!-- stage %d -->" % si, "      <trees>"]
        for j, tree in enumerate(trees):
            L += ["        <_>", "          <!-- tree %d -->" % j]
            for k, nd in enumerate(tree):
                L += ["          <_>", "            <!-- %s -->" % ("root node" if k == 0 else "node %d" % k), "            <feature>", "              <rects>"]
                for (x, y, w, h, wt) in nd["feature"]:
                    L.append("                <_>%d %d %d %d %d.</_>" % (x, y, w, h, int(wt)))
                L += ["              </rects>", "              <tilted>%d</tilted></feature>" % int(nd.get("tilted", 0)),
                      "            <threshold>%s</threshold>" % _f(nd["threshold"])]
                for side in ("left", "right"):
                    kind, v = nd[side]
                    L.append("            <%s_%s>%s</%s_%s>" % (side, kind, _f(v) if kind == "val" else "%d" % v, side, kind))
                L[-1] += "</_>"
            L[-1] += "</_>"
        L += ["      </trees>", "      <stage_threshold>%s</stage_threshold>" % _f(st["stage_threshold"]),
              "      <parent>%d</parent>" % (si - 1), "      <next>-1</next></_>"]
    L += ["  </stages></%s>" % casc["name"], "</opencv_storage>", ""]
    return "\n".join(L)


def _rand_tilted_feature(rng, ow, oh):
    """A tilted (45 degree) two-rectangle feature inside the window: rect (x, y, w, h) tilted covers columns x-h .. x+w and
    rows y .. y+w+h (OpenCV's convention: w along the down-right diagonal, h along the down-left one)."""
    for _ in range(200):
        w, h = int(rng.randint(1, max(2, ow // 3))) * 2, int(rng.randint(1, max(2, oh // 3)))
        x, y = int(rng.randint(0, ow)), int(rng.randint(0, oh))
        if x - h < 1 or x + w > ow - 1 or y < 1 or y + w + h > oh - 1:
            continue
        if rng.rand() < 0.5:       # halves along w
            return [(x, y, w, h, -1.0), (x, y, w // 2, h, 2.0)]
        if h % 2 == 0:             # halves along h
            return [(x, y, w, h, -1.0), (x, y, w, h // 2, 2.0)]
    raise RuntimeError("tilted feature sampling failed")


def _rand_upright_feature(rng, ow, oh):
    while True:
        w, h = int(rng.randint(2, ow + 1)), int(rng.randint(2, oh + 1))
        x, y = int(rng.randint(0, ow - w + 1)), int(rng.randint(0, oh - h + 1))
        kind = int(rng.randint(0, 4))
        if kind == 0 and w % 2 == 0:
            return [(x, y, w, h, -1.0), (x + w // 2, y, w // 2, h, 2.0)]
        if kind == 1 and h % 2 == 0:
            return [(x, y, w, h, -1.0), (x, y + h // 2, w, h // 2, 2.0)]
        if kind == 2 and w % 3 == 0:
            return [(x, y, w, h, -1.0), (x + w // 3, y, w // 3, h, 3.0)]
        if kind == 3 and w % 2 == 0 and h % 2 == 0:
            return [(x, y, w, h, -1.0), (x, y, w // 2, h // 2, 2.0), (x + w // 2, y + h // 2, w // 2, h // 2, 2.0)]


def make_generic_cascade(ow=20, oh=20, seed=1, stage_sizes=(3, 8, 12, 16, 20, 24, 28), tilt_frac=0.25, tree_frac=0.35, max_nodes=3):
    """A lenient cascade that exercises what haarcascade_profileface / the mcs_* files may contain:
    tilted features and tree-structured weak classifiers (up to `max_nodes` nodes, children always later nodes).  Votes are
    symmetric around zero and thresholds near zero, so about half of all windows pass each stage on any image and a handful
    survive every stage.  A workload / parity generator, not a detector."""
    rng = np.random.RandomState(seed)
    stages = []
    for n in stage_sizes:
        trees = []
        for _ in range(n):
            nn = int(rng.randint(2, max_nodes + 1)) if rng.rand() < tree_frac else 1
            nodes = []
            next_free = 1                       # nodes are handed out to parents in index order: a proper binary tree, nn + 1 leaves
            for k in range(nn):
                tilted = 1 if rng.rand() < tilt_frac else 0
                feat = _rand_tilted_feature(rng, ow, oh) if tilted else _rand_upright_feature(rng, ow, oh)
                a = float(rng.uniform(0.3, 1.0))
                avail = nn - next_free
                lo = 1 if (next_free == k + 1 and avail > 0) else 0      # the next node must hang somewhere before its turn comes
                take = int(rng.randint(lo, min(2, avail) + 1)) if avail > 0 else 0
                sides = [("node", next_free + j) for j in range(take)]
                next_free += take
                v = a if rng.rand() < 0.5 else -a
                while len(sides) < 2:
                    sides.append(("val", v)); v = -v
                if rng.rand() < 0.5:
                    sides.reverse()
                nodes.append(dict(feature=feat, tilted=tilted, threshold=float(rng.normal(0, 0.02)), left=sides[0], right=sides[1]))
            trees.append(nodes)
        stages.append(dict(trees=trees, stage_threshold=float(rng.uniform(-0.3, 0.1))))
    return dict(name="generic_%dx%d" % (ow, oh), size=(ow, oh), stages=stages)


def generic_cascade_xml(**kw):
    return cascade_to_xml(make_generic_cascade(**kw))


def synthetic_cascade_xml(seed=2016, stages=None):
    return cascade_to_xml(make_cascade(seed=seed, stages=stages))


# ------------------------------------------------------------------ calibration on frame content
def equalize_np(g):
    """cv::equalizeHist as numpy (workload generation only: the checker and the kernels have their own)."""
    hist = np.bincount(g.ravel(), minlength=256)
    nz = np.nonzero(hist)[0]
    if len(nz) <= 1:
        return g.copy()
    i0 = nz[0]
    scale = 255.0 / (g.size - hist[i0])
    c = np.cumsum(hist) - hist[:i0 + 1].sum()
    lut = np.clip(np.rint(np.maximum(c, 0) * scale), 0, 255).astype(np.uint8)
    lut[:i0 + 1] = 0
    return lut[g]


def scan_ladder(ow, oh, cols, rows, scale_factor=1.1, min_size=(0, 0)):
    """The factors cvHaarDetectObjectsForROC's scale-cascade branch evaluates."""
    n, f = 0, 1.0
    while f * ow < cols - 10 and f * oh < rows - 10:
        n += 1; f *= scale_factor
    out, f = [], 1.0
    for _ in range(n):
        if int(np.rint(ow * f)) >= min_size[0] and int(np.rint(oh * f)) >= min_size[1]:
            out.append(f)
        f *= scale_factor
    return out


class WindowSample:
    """Every window a scale-cascade scan visits on a few gray images (OpenCV's grid: stride max(2, factor), origins cvRound(i * stride)),
    with the arithmetic of cvSetImagesForHaarClassifierCascade / cvRunHaarClassifierCascadeSum in float64: scaled rectangles
    cvRound(v * factor), weights over the variance window's area with rectangle 0 re-balanced, value compared with
    threshold * std of the window.  make_cascade() walks it stage by stage, keeping the windows that pass."""

    def __init__(self, images, win=WIN, scale_factor=1.1, min_size=None):
        self.win = win
        self.planes = []          # (S flat, pitch)
        self.groups = []          # per (image, factor): dict(img, f, off [n] int64, vnf [n])
        rnd = lambda v: int(np.rint(v))
        for gi, g in enumerate(images):
            g = np.asarray(g, np.uint8)
            H, W = g.shape
            S = np.zeros((H + 1, W + 1), np.int64); S[1:, 1:] = g.astype(np.int64).cumsum(0).cumsum(1)
            Q = np.zeros((H + 1, W + 1), np.int64); Q[1:, 1:] = (g.astype(np.int64) ** 2).cumsum(0).cumsum(1)
            P = W + 1
            self.planes.append((S.ravel(), P))
            ms = (W // 20, H // 20) if min_size is None else min_size
            for f in scan_ladder(win, win, W, H, scale_factor, ms):
                step = max(2.0, f)
                ww = rnd(win * f)
                ex = np.rint(np.arange(rnd((W - ww) / step)) * step).astype(np.int64)
                ey = np.rint(np.arange(rnd((H - ww) / step)) * step).astype(np.int64)
                if not len(ex) or not len(ey):
                    continue
                off = (ey[:, None] * P + ex[None, :]).ravel()
                e0, ew = rnd(f), rnd((win - 2) * f)
                inv = 1.0 / (ew * ew)
                c = [off + e0 * P + e0, off + e0 * P + e0 + ew, off + (e0 + ew) * P + e0, off + (e0 + ew) * P + e0 + ew]
                Sf, Qf = S.ravel(), Q.ravel()
                mean = (Sf[c[0]] - Sf[c[1]] - Sf[c[2]] + Sf[c[3]]) * inv
                var = (Qf[c[0]] - Qf[c[1]] - Qf[c[2]] + Qf[c[3]]) * inv - mean * mean
                vnf = np.where(var >= 0, np.sqrt(np.maximum(var, 0)), 1.0)
                self.groups.append(dict(img=gi, f=f, off=off, vnf=vnf, inv=inv))
        self.total = self.alive_count()

    def alive_count(self):
        return int(sum(len(g["off"]) for g in self.groups))

    def stage_votes(self, feats, thr, lv, rv):
        rnd = lambda v: int(np.rint(v))
        out = []
        for g in self.groups:
            Sf, P = self.planes[g["img"]]
            f, off, inv = g["f"], g["off"], g["inv"]
            tot = np.zeros(len(off))
            if len(off):
                for feat, t, a0, a1 in zip(feats, thr, lv, rv):
                    rs, ws = [], []
                    for (x, y, w, h, wt) in feat:
                        tx, ty, tw, th = rnd(x * f), rnd(y * f), rnd(w * f), rnd(h * f)
                        rs.append((tx, ty, tw, th)); ws.append(wt * inv)
                    ws[0] = -sum(ws[k] * rs[k][2] * rs[k][3] for k in range(1, len(rs))) / (rs[0][2] * rs[0][3])
                    val = np.zeros(len(off))
                    for (tx, ty, tw, th), wk in zip(rs, ws):
                        o = off + ty * P + tx
                        val += wk * (Sf[o] - Sf[o + tw] - Sf[o + th * P] + Sf[o + th * P + tw])
                    tot += np.where(val >= np.float32(t) * g["vnf"], a1, a0)
            out.append(tot)
        return np.concatenate(out) if out else np.zeros(0)

    def keep(self, mask):
        k = 0
        for g in self.groups:
            n = len(g["off"]); m = mask[k:k + n]; k += n
            g["off"] = g["off"][m]; g["vnf"] = g["vnf"][m]


_CALIB_CACHE = {}


def calibrated_cascade_xml(seed=2016, stages=None, W=1920, H=1080, n_images=2, content="natural", scale_factor=1.1):
    """The stand-in cascade with its stage thresholds calibrated on the bench's own content: every stage lets about half of
    the windows that reach it through (windows of the equalised 1/f field at the scan's own scales -- haartraining's
    false-alarm target per stage), for as many stages as the sample supports (ten at 2 x 1080p); pasted templates still pass
    everything.  A trained cascade's selectivity profile: monotone, no stage that a re-ordering could exploit."""
    key = (seed, tuple(stages) if stages else None, W, H, n_images, content, scale_factor)
    if key not in _CALIB_CACHE:
        imgs = [equalize_np(make_gray(W, H, frame_seed(0, 1000 + i), content)) for i in range(n_images)]
        c = make_cascade(seed=seed, stages=stages, calib=WindowSample(imgs, scale_factor=scale_factor))
        c["name"] = "synthetic_frontalface_calibrated"
        _CALIB_CACHE[key] = cascade_to_xml(c)
    return _CALIB_CACHE[key]


PART_STAGES = [3, 9, 14, 19, 20, 27, 31, 34, 37, 42, 47, 50]     # a shorter cascade, like the mcs_* files


def synthetic_part_cascade_xml(name, seed=None):
    seed = {"righteye": 101, "lefteye": 102, "nose": 103, "mouth": 104, "leftear": 105, "rightear": 106}[name] if seed is None else seed
    c = make_cascade(seed=seed, stages=PART_STAGES, tmpl=part_template(name))
    c["name"] = "synthetic_" + name
    return cascade_to_xml(c)


# ------------------------------------------------------------------ frames
def _resize_bilinear_f(img, size):
    h, w = img.shape
    ys = (np.arange(size) + 0.5) * h / size - 0.5
    xs = (np.arange(size) + 0.5) * w / size - 0.5
    y0 = np.clip(np.floor(ys).astype(int), 0, h - 1); y1 = np.clip(y0 + 1, 0, h - 1)
    x0 = np.clip(np.floor(xs).astype(int), 0, w - 1); x1 = np.clip(x0 + 1, 0, w - 1)
    fy = np.clip(ys - np.floor(ys), 0, 1)[:, None]; fx = np.clip(xs - np.floor(xs), 0, 1)[None, :]
    a = img[y0][:, x0]; b = img[y0][:, x1]; c = img[y1][:, x0]; d = img[y1][:, x1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_gray(W, H, seed, kind="noise", faces=()):
    """Seeded luminance field, uint8 [H,W].
    kind: 'noise' uniform u8; 'natural' 1/f multi-octave noise; 'gradient' smooth ramp +
    mild noise; 'flat'.
    faces: iterable of (x, y, size): TEMPLATE pasted at that box."""
    rng = np.random.default_rng(seed)
    if kind == "noise":
        g = rng.integers(0, 256, size=(H, W)).astype(np.float64)
    elif kind == "gradient":
        yy, xx = np.mgrid[0:H, 0:W]
        g = 40 + 170.0 * (0.6 * xx / max(W - 1, 1) + 0.4 * yy / max(H - 1, 1)) + rng.normal(0, 3.0, size=(H, W))
    elif kind == "natural":
        # 1/f-like field: equal-amplitude octaves of bilinearly upsampled value noise, so that
        # Haar responses do not vanish at large window scales the way they do on white noise
        g = np.zeros((H, W))
        cell, n_oct = 2, 0
        while cell <= max(W, H):
            gh, gw = H // cell + 3, W // cell + 3
            grid = rng.uniform(-1, 1, size=(gh, gw))
            ys = np.arange(H) / cell; xs = np.arange(W) / cell
            y0 = ys.astype(int); x0 = xs.astype(int)
            fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
            top = grid[y0][:, x0] * (1 - fx) + grid[y0][:, x0 + 1] * fx
            bot = grid[y0 + 1][:, x0] * (1 - fx) + grid[y0 + 1][:, x0 + 1] * fx
            g += top * (1 - fy) + bot * fy
            cell *= 2; n_oct += 1
        g += rng.uniform(-1, 1, size=(H, W))
        g = 128 + g * (100.0 / np.sqrt(n_oct + 1))
    elif kind == "flat":
        g = np.full((H, W), 128.0)
    else:
        raise ValueError(kind)
    g = np.clip(np.rint(g), 0, 255).astype(np.uint8)
    return paste_faces(g, faces, seed) if len(faces) else g


def paste_faces(gray, faces, seed=0):
    """copy of `gray` with TEMPLATE pasted at every (x, y, size)"""
    rng = np.random.default_rng(seed ^ 0xFACE)
    g = gray.astype(np.float64)
    H, W = g.shape
    T = template()
    for (x, y, size) in faces:
        patch = _resize_bilinear_f(T, int(size)) + rng.normal(0, 2.0, size=(int(size), int(size)))
        g[y:y + size, x:x + size] = patch[:max(0, min(size, H - y)), :max(0, min(size, W - x))]
    return np.clip(np.rint(g), 0, 255).astype(np.uint8)


def make_bgr(W, H, seed, kind="noise", faces=(), channels=3):
    """BGR (or BGRA) frame whose BGR2GRAY is close to make_gray's field."""
    return gray_to_bgr(make_gray(W, H, seed, kind, faces), seed, channels)


def gray_to_bgr(g, seed, channels=3):
    H, W = g.shape
    rng = np.random.default_rng(seed ^ 0x5EED)
    out = np.empty((H, W, channels), np.uint8)
    jitter = rng.integers(-6, 7, size=(H, W, 2))
    out[..., 0] = np.clip(g.astype(np.int32) + jitter[..., 0], 0, 255)
    out[..., 1] = g
    out[..., 2] = np.clip(g.astype(np.int32) + jitter[..., 1], 0, 255)
    if channels == 4:
        out[..., 3] = 255
    return out


def frame_seed(stream_id, frame_idx):
    """seed = 0xC0FFEE + stream_id*1000 + frame_idx."""
    return 0xC0FFEE + stream_id * 1000 + frame_idx


def calibrated_part_cascade_xml(name, seed=None, n_images=8, pass_rate=0.65):
    """A part detector's stand-in cascade with its stage thresholds calibrated the way calibrated_cascade_xml does it, on what the part
    searches scan: face regions of the working image (the face TEMPLATE at 88 .. 104 pixels -- a 560 .. 620-pixel face of a 1080p frame
    on the 320-pixel working image -- on the 1/f field, equalised), windows from 20 pixels up at OpenCV's grid.  Every early stage
    lets about two thirds of what reaches it through (pass_rate 0.65: with 0.5 the parts of the bench's faces lose neighbours to the
    stricter stages and a quarter of them falls below minNeighbors; the mcs_* files are short, lenient cascades too); the part's
    own template keeps its margin."""
    key = ("part", name, seed, n_images, pass_rate)
    if key not in _CALIB_CACHE:
        pseed = {"righteye": 101, "lefteye": 102, "nose": 103, "mouth": 104, "leftear": 105, "rightear": 106}[name] if seed is None else seed
        imgs = []
        for i in range(n_images):
            size = 88 + 2 * i
            g = make_gray(176, 176, 7000 + 13 * i + pseed, "natural", [(176 // 2 - size // 2, 176 // 2 - size // 2 + (i % 3) - 1, size)])
            imgs.append(equalize_np(g))
        c = make_cascade(seed=pseed, stages=PART_STAGES, tmpl=part_template(name), calib=WindowSample(imgs, scale_factor=1.1, min_size=(20, 20)), pass_rate=pass_rate)
        c["name"] = "synthetic_%s_calibrated" % name
        _CALIB_CACHE[key] = cascade_to_xml(c)
    return _CALIB_CACHE[key]
