"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Bit-exact everywhere (integer / byte / index work;
the f32/f64 comparisons inside the cascade are reproduced operation by
operation, so raw candidate lists must be identical, in order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from nubovca import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def casc(ctx, synth_xml):
    return ctx.load_cascade_xml(synth_xml)


@pytest.fixture(scope="module")
def casc_small(ctx, small_xml):
    return ctx.load_cascade_xml(small_xml)


# ------------------------------------------------------------------ loader
def test_loader_matches_python_reader(casc, orc_cascade):
    d = casc.dump()
    assert d["size"] == (orc_cascade.ow, orc_cascade.oh)
    assert np.array_equal(d["stage_sizes"], orc_cascade.stage_ncls)
    assert np.array_equal(d["stage_thr"], orc_cascade.stage_thr)
    assert np.array_equal(d["rects"], orc_cascade.rects)
    assert np.array_equal(d["weights"], orc_cascade.rweights)
    assert np.array_equal(d["thr"], orc_cascade.node_thr)
    assert np.array_equal(d["left"], orc_cascade.alpha[0::2])
    assert np.array_equal(d["right"], orc_cascade.alpha[1::2])


def test_loader_rejects_garbage(ctx):
    from nubovca import capi
    for bad in ("", "<opencv_storage></opencv_storage>", "<a><b></a>", "not xml at all"):
        with pytest.raises(capi.NvcaError):
            ctx.load_cascade_xml(bad or " ")


# ------------------------------------------------------------------ primitives
@pytest.mark.parametrize("w,h,cn", [(1, 1, 3), (7, 5, 3), (64, 64, 3), (161, 33, 3), (640, 480, 3), (1920, 1080, 3),
                                    (333, 77, 4), (1280, 720, 4)])
def test_bgr2gray(ctx, w, h, cn):
    import orc
    img = np.random.default_rng(w * h + cn).integers(0, 256, size=(h, w, cn), dtype=np.uint8)
    assert np.array_equal(ctx.bgr2gray(img), orc.bgr2gray(img))


@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 160, 120), (1920, 1080, 160, 90), (1280, 720, 320, 180),
                                         (640, 480, 320, 240), (100, 80, 100, 80), (97, 61, 41, 29), (50, 40, 120, 90),
                                         (1920, 1080, 640, 360), (33, 2, 7, 1)])
def test_resize_gray(ctx, sw, sh, dw, dh):
    import orc
    img = np.random.default_rng(sw + dw).integers(0, 256, size=(sh, sw), dtype=np.uint8)
    assert np.array_equal(ctx.resize_linear(img, dw, dh), orc.resize_linear(img, dw, dh))


@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 160, 120), (1920, 1080, 160, 90), (322, 242, 161, 121), (97, 61, 41, 29),
                                         (50, 40, 120, 90), (64, 48, 64, 48)])
def test_resize_bgr(ctx, sw, sh, dw, dh):
    import orc
    img = np.random.default_rng(sw * dw).integers(0, 256, size=(sh, sw, 3), dtype=np.uint8)
    assert np.array_equal(ctx.resize_linear(img, dw, dh), orc.resize_linear(img, dw, dh))


@pytest.mark.parametrize("w,h,kind", [(160, 90, "natural"), (640, 480, "noise"), (1920, 1080, "natural"),
                                      (333, 77, "gradient"), (64, 64, "flat"), (5, 3, "noise")])
def test_equalize_hist(ctx, w, h, kind):
    import orc
    from nubovca import synth
    img = synth.make_gray(w, h, 5, kind)
    assert np.array_equal(ctx.equalize_hist(img), orc.equalize_hist(img))


def test_equalize_lut_division_sweep(ctx):
    """every LUT entry goes through a device f32 divide + multiply + round: sweep many totals."""
    import orc
    rng = np.random.default_rng(0)
    for _ in range(40):
        w, h = int(rng.integers(8, 400)), int(rng.integers(8, 300))
        lo, hi = sorted(rng.integers(0, 256, size=2))
        img = rng.integers(lo, hi + 1, size=(h, w)).astype(np.uint8)
        assert np.array_equal(ctx.equalize_hist(img), orc.equalize_hist(img))


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (64, 16), (65, 17), (640, 480), (1920, 1080), (2047, 20), (2048, 20),
                                 (2500, 37), (4100, 33),
                                 (125, 128), (126, 127), (255, 63), (1000, 16), (63, 200), (160, 90), (3, 1000)])     # around the LDS-resident small-image path
def test_integral(ctx, w, h):
    import orc
    img = np.random.default_rng(w + h).integers(0, 256, size=(h, w), dtype=np.uint8)
    s, q = ctx.integral(img)
    es, eq = orc.integral(img)
    assert np.array_equal(s, es)
    assert np.array_equal(q, eq)


def test_integral_all_white_1080p(ctx):
    img = np.full((1080, 1920), 255, np.uint8)
    s, q = ctx.integral(img)
    assert s[-1, -1] == 255 * 1920 * 1080 and q[-1, -1] == 255.0 * 255 * 1920 * 1080


# ------------------------------------------------------------------ detectMultiScale
CASES = [
    (160, 120, "natural", [(40, 30, 60)], 1.25, (8, 6)),
    (320, 240, "natural", [(60, 40, 100), (200, 120, 50)], 1.1, (16, 12)),
    (640, 480, "noise", [(100, 80, 120)], 1.1, (32, 24)),
    (640, 480, "gradient", [(100, 80, 120), (300, 200, 60)], 1.2, (0, 0)),
    (333, 251, "natural", [(30, 20, 90)], 1.1, (0, 0)),
    (1280, 720, "natural", [(200, 150, 300), (900, 400, 180)], 1.1, (64, 36)),
]


@pytest.mark.parametrize("w,h,kind,faces,sf,ms", CASES)
def test_detect_raw_and_grouped(ctx, casc, orc_cascade, w, h, kind, faces, sf, ms):
    import orc
    from nubovca import synth
    g = orc.equalize_hist(synth.make_gray(w, h, 3, kind, faces))
    raw = ctx.detect_raw(casc, g, sf, 0, ms)
    eraw = orc.detect_raw(orc_cascade, g, sf, 0, ms)
    assert np.array_equal(raw, eraw), (len(raw), len(eraw))
    assert len(eraw) > 0
    det = ctx.detect_multiscale(casc, g, sf, 3, 0, ms)
    edet = orc.detect_multiscale(orc_cascade, g, sf, 3, 0, ms)
    assert np.array_equal(det, edet)


def test_detect_1080p_fullres(ctx, casc, orc_cascade):
    """BASELINE config 2 geometry: 1920x1080, sf 1.1, minSize (96,54): 25 scales."""
    import orc
    from nubovca import synth
    faces = [(200, 150, 300), (900, 400, 180), (1400, 100, 120), (1500, 700, 240)]
    g = orc.equalize_hist(synth.make_gray(1920, 1080, 3, "natural", faces))
    raw = ctx.detect_raw(casc, g, 1.1, 0, (96, 54))
    eraw, st = orc.detect_raw(orc_cascade, g, 1.1, 0, (96, 54), return_stats=True)
    assert st.n_scales == 25
    assert np.array_equal(raw, eraw)
    det = ctx.detect_multiscale(casc, g, 1.1, 3, 0, (96, 54))
    assert np.array_equal(det, orc.detect_multiscale(orc_cascade, g, 1.1, 3, 0, (96, 54)))
    assert len(det) == 4


def test_detect_lenient_cascade_many_hits(ctx, casc_small, orc_small):
    """a 6-stage cascade lets thousands of windows through: stresses the queue / candidate path."""
    import orc
    from nubovca import synth
    g = orc.equalize_hist(synth.make_gray(400, 300, 8, "gradient", [(50, 40, 150)]))
    raw = ctx.detect_raw(casc_small, g, 1.1, 0, (0, 0))
    eraw = orc.detect_raw(orc_small, g, 1.1, 0, (0, 0))
    assert len(eraw) > 100
    assert np.array_equal(raw, eraw)
    det = ctx.detect_multiscale(casc_small, g, 1.1, 3, 0, (0, 0))
    assert np.array_equal(det, orc.detect_multiscale(orc_small, g, 1.1, 3, 0, (0, 0)))


@pytest.mark.parametrize("w,h,kind,seed", [(400, 300, "gradient", 8), (640, 480, "noise", 5), (800, 600, "natural", 11),
                                           (1280, 720, "noise", 2)])
def test_device_grouping_sweep(ctx, casc_small, orc_small, w, h, kind, seed):
    """groupRectangles runs on the device for the plain scan (k_group): many clusters, every threshold, and the
    frames with more raw candidates than the kernel takes (host grouping takes over) must all match."""
    import orc
    from nubovca import synth
    g = orc.equalize_hist(synth.make_gray(w, h, seed, kind, [(50, 40, 150), (220, 60, 90)]))
    nraw = len(orc.detect_raw(orc_small, g, 1.1, 0, (0, 0)))
    assert nraw > 50
    for mn in (1, 2, 3, 5, 9):
        det = ctx.detect_multiscale(casc_small, g, 1.1, mn, 0, (0, 0))
        edet = orc.detect_multiscale(orc_small, g, 1.1, mn, 0, (0, 0))
        assert np.array_equal(det, edet), (mn, nraw, len(det), len(edet))


def test_detect_random_geometries(ctx, casc, casc_small, orc_cascade, orc_small):
    """seeded sweep over odd image sizes, scale factors and size limits: tile / band planning (edge tiles, rows shorter
    than a tile, scales with a handful of windows) must never change a candidate; both batch-1 and forced-band paths"""
    import os
    import orc
    from nubovca import synth
    rng = np.random.RandomState(20240611)
    kinds = ["natural", "noise", "gradient"]
    checked = 0
    for it in range(28):
        w, h = int(rng.randint(41, 700)), int(rng.randint(41, 500))
        sf = float(rng.choice([1.1, 1.15, 1.2, 1.25, 1.3, 1.5]))
        ms = (int(rng.randint(0, max(1, w // 6))), int(rng.randint(0, max(1, h // 6)))) if it % 3 else (0, 0)
        mx = (0, 0) if it % 4 else (int(rng.randint(w // 3, w + 1)), int(rng.randint(h // 3, h + 1)))
        s = int(min(w, h) * rng.uniform(0.3, 0.8))
        faces = [(int(rng.randint(0, max(1, w - s))), int(rng.randint(0, max(1, h - s))), s)] if s >= 24 else []
        g = orc.equalize_hist(synth.make_gray(w, h, 1000 + it, kinds[it % 3], faces))
        c, oc = (casc_small, orc_small) if it % 5 == 0 else (casc, orc_cascade)
        eraw = orc.detect_raw(oc, g, sf, 0, ms, mx)
        for band in (0, 1):
            with ctx.options(band=band):
                raw = ctx.detect_raw(c, g, sf, 0, ms, mx)
            assert np.array_equal(raw, eraw), (it, band, w, h, sf, ms, mx, len(raw), len(eraw))
        det = ctx.detect_multiscale(c, g, sf, 2, 0, ms, mx)
        assert np.array_equal(det, orc.detect_multiscale(oc, g, sf, 2, 0, ms, mx)), (it, w, h)
        checked += len(eraw)
    assert checked > 100


def _rect_cascade_xml(ow, oh, seed, stage_sizes=(3, 8, 12, 16, 20, 24, 28)):
    """a stump cascade with a RECTANGULAR window, like haarcascade_mcs_mouth (25x15), _nose (18x15), _eyes (18x12), _ears
    (12x20): random 2- and 3-rect features inside ow x oh, vote sums symmetric around 0 (about half of all windows pass a
    stage), so a handful of windows survive seven stages on any image."""
    from nubovca import synth
    rng = np.random.RandomState(seed)
    stages = []
    for n in stage_sizes:
        feats, thr, lv, rv = [], [], [], []
        for _ in range(n):
            while True:
                w, h = int(rng.randint(2, ow + 1)), int(rng.randint(2, oh + 1))
                x, y = int(rng.randint(0, ow - w + 1)), int(rng.randint(0, oh - h + 1))
                kind = int(rng.randint(0, 4))
                if kind == 0 and w % 2 == 0:
                    f = [(x, y, w, h, -1.0), (x + w // 2, y, w // 2, h, 2.0)]; break
                if kind == 1 and h % 2 == 0:
                    f = [(x, y, w, h, -1.0), (x, y + h // 2, w, h // 2, 2.0)]; break
                if kind == 2 and w % 3 == 0:
                    f = [(x, y, w, h, -1.0), (x + w // 3, y, w // 3, h, 3.0)]; break
                if kind == 3 and w % 2 == 0 and h % 2 == 0:
                    f = [(x, y, w, h, -1.0), (x, y, w // 2, h // 2, 2.0), (x + w // 2, y + h // 2, w // 2, h // 2, 2.0)]; break
            a = float(rng.uniform(0.3, 1.0))
            feats.append(f); thr.append(float(rng.normal(0, 0.02))); lv.append(-a if rng.rand() < 0.5 else a); rv.append(-lv[-1])
        stages.append(dict(features=feats, thresholds=thr, left=lv, right=rv, stage_threshold=float(rng.uniform(-0.3, 0.1))))
    return synth.cascade_to_xml(dict(name="rect_window_%dx%d" % (ow, oh), size=(ow, oh), stages=stages))


@pytest.mark.parametrize("ow,oh", [(25, 15), (18, 12), (12, 20), (18, 15)])
def test_rectangular_window_cascades(ctx, ow, oh):
    """the mcs_* cascades the part detectors load have non-square windows: all three scan variants, both policies"""
    import orc
    from nubovca import capi, synth
    xml = _rect_cascade_xml(ow, oh, 7 * ow + oh)
    c, oc = ctx.load_cascade_xml(xml), orc.parse_cascade_xml(xml)
    total = 0
    for it, (w, h, kind, sf) in enumerate([(333, 251, "natural", 1.1), (200, 150, "noise", 1.2), (640, 360, "gradient", 1.15)]):
        g = orc.equalize_hist(synth.make_gray(w, h, 50 + it, kind))
        eraw = orc.detect_raw(oc, g, sf, 0, (0, 0))
        assert np.array_equal(ctx.detect_raw(c, g, sf, 0, (0, 0)), eraw), (ow, oh, w, h)
        total += len(eraw)
        assert np.array_equal(ctx.detect_multiscale(c, g, sf, 2, 0, (ow + 5, oh + 3)), orc.detect_multiscale(oc, g, sf, 2, 0, (ow + 5, oh + 3)))
        assert np.array_equal(ctx.detect_multiscale(c, g, sf, 2, capi.HAAR_SCALE_IMAGE, (3, 3)),
                              orc.detect_multiscale(oc, g, sf, 2, capi.HAAR_SCALE_IMAGE, (3, 3)))
        assert np.array_equal(ctx.detect_multiscale(c, g, sf, 3, capi.HAAR_FIND_BIGGEST_OBJECT, (1, 1)),
                              orc.detect_multiscale(oc, g, sf, 3, capi.HAAR_FIND_BIGGEST_OBJECT, (1, 1)))
    assert total > 20
    ctx.set_sum_policy(capi.SUM_F64)
    try:
        g = orc.equalize_hist(synth.make_gray(300, 200, 9, "natural"))
        assert np.array_equal(ctx.detect_raw(c, g, 1.1, 0, (0, 0)), orc.detect_raw(oc, g, 1.1, 0, (0, 0), policy=orc.SUM_F64))
    finally:
        ctx.set_sum_policy(capi.SUM_F32PAIR)


def test_detect_f64_policy(ctx, casc, orc_cascade):
    import orc
    from nubovca import capi, synth
    g = orc.equalize_hist(synth.make_gray(320, 240, 12, "natural", [(60, 40, 100)]))
    ctx.set_sum_policy(capi.SUM_F64)
    try:
        raw = ctx.detect_raw(casc, g, 1.1, 0, (0, 0))
    finally:
        ctx.set_sum_policy(capi.SUM_F32PAIR)
    assert np.array_equal(raw, orc.detect_raw(orc_cascade, g, 1.1, 0, (0, 0), policy=orc.SUM_F64))


def test_detect_degenerate_images(ctx, casc, orc_cascade):
    import orc
    for img in (np.zeros((60, 80), np.uint8), np.full((31, 31), 200, np.uint8), np.zeros((25, 400), np.uint8)):
        assert np.array_equal(ctx.detect_raw(casc, img, 1.1, 0, (0, 0)), orc.detect_raw(orc_cascade, img, 1.1, 0, (0, 0)))


def test_hit_capacity_overflow_is_answered_or_loud(ctx, casc_small, orc_small):
    """more raw candidates than the lists hold: a detectMultiScale call re-runs its launch set with lists of the exact size and
    answers like the reference (every scan variant); the batched face path reports NVCA_ERR_OVERFLOW for that batch -- never a
    truncated list -- and answers the following batches, whose lists it sizes for what the refused one produced"""
    import orc
    from nubovca import capi, synth
    g = orc.equalize_hist(synth.make_gray(400, 300, 8, "gradient", [(50, 40, 150)]))
    ctx.set_hit_capacity(16)
    try:
        eraw = orc.detect_raw(orc_small, g, 1.1, 0, (0, 0))
        assert len(eraw) > 16
        assert np.array_equal(ctx.detect_raw(casc_small, g, 1.1, 0, (0, 0)), eraw)
        assert np.array_equal(ctx.detect_raw(casc_small, g, 1.1, capi.HAAR_SCALE_IMAGE, (0, 0)), orc.detect_raw(orc_small, g, 1.1, orc.HAAR_SCALE_IMAGE, (0, 0)))
        for fl in (0, capi.HAAR_SCALE_IMAGE, capi.HAAR_FIND_BIGGEST_OBJECT):
            assert np.array_equal(ctx.detect_multiscale(casc_small, g, 1.1, 2, fl, (0, 0)), orc.detect_multiscale(orc_small, g, 1.1, 2, fl, (0, 0))), fl
        bgr = synth.make_bgr(400, 300, 8, "gradient", [(50, 40, 150)])
        fs = capi.FaceStream(ctx, casc_small, width_to_process=400, multi_scale_factor=10)
        with pytest.raises(capi.NvcaError) as e:
            fs.process(bgr)
        assert e.value.code == capi.ERR_OVERFLOW
        b1, i1 = fs.process(bgr)                          # lists sized for what the refused batch produced: the stream goes on
        eb, _ = orc.FaceStream(orc_small, width_to_process=400, scale_factor_pct=10).process(bgr)
        assert len(eb) >= 1 and np.array_equal(b1, eb)    # (first tracked frame of either stream: the detections themselves)
    finally:
        ctx.set_hit_capacity(16384)


def test_group_rectangles_abi(ctx):
    import orc
    rng = np.random.default_rng(1)
    base = rng.integers(0, 300, size=(12, 2))
    rects = []
    for bx, by in base:
        for _ in range(int(rng.integers(1, 9))):
            s = int(rng.integers(30, 34))
            rects.append([bx + int(rng.integers(-2, 3)), by + int(rng.integers(-2, 3)), s, s])
    rects = np.array(rects, np.int32)[rng.permutation(len(rects))]
    for thr in (0, 1, 2, 3):
        exp, _ = orc.group_rectangles(rects, thr)
        assert np.array_equal(ctx.group_rectangles(rects, thr), exp)


# ------------------------------------------------------------------ NuboFaceDetector stream
def _sequence(W, H, n, seed0):
    from nubovca import synth
    frames = []
    for i in range(n):
        if i % 5 == 3:
            frames.append(synth.make_bgr(W, H, seed0 + i, "natural"))          # no face
        else:
            x = 40 + 6 * i
            frames.append(synth.make_bgr(W, H, seed0 + i, "natural", [(x, H // 6, H // 2)]))
    return frames


@pytest.mark.parametrize("W,H,props", [
    (640, 480, {}),                                                   # BASELINE config 1: w2p 160 -> 160x120, sf 1.25
    (1280, 720, {"width_to_process": 320}),
    (320, 240, {"width_to_process": 320, "multi_scale_factor": 10}),  # full-res mode
    (640, 480, {"process_x_every_4_frames": 2}),
    (640, 480, {"process_x_every_4_frames": 3, "width_to_process": 213}),
    (322, 242, {"width_to_process": 161}),                            # exact 2x -> area-fast resize
])
def test_face_stream_sequence(ctx, casc, orc_cascade, W, H, props):
    import orc
    from nubovca import capi
    kw = {"width_to_process": "width_to_process", "process_x_every_4_frames": "process_x_every_4",
          "multi_scale_factor": "scale_factor_pct"}
    fs = capi.FaceStream(ctx, casc, **props)
    ofs = orc.FaceStream(orc_cascade, **{kw[k]: v for k, v in props.items()})
    seen = 0
    for f in _sequence(W, H, 12, 100):
        boxes, ids = fs.process(f)
        eb, eid = ofs.process(f)
        assert np.array_equal(boxes, eb), (boxes, eb)
        assert np.array_equal(ids, eid)
        seen += len(eb)
    assert seen > 0
    fs.close()


@pytest.mark.parametrize("W,H,w2p", [
    (1920, 1080, 160),       # ratio 12: rows 12k+5, 12k+6 cross PCIe as one strided copy
    (1920, 1080, 320),       # ratio 6
    (1280, 720, 160),        # ratio 8
    (1000, 562, 160),        # 1000/166: irregular row pattern -> whole frames
    (644, 484, 160),         # ratio 4 + a remainder (scale 4, 161 x 121): the run pattern reaches the last row
])
def test_face_host_frames_shrinking_ingest(ctx, casc, orc_cascade, W, H, w2p):
    """host frames in the reference's shrink-first mode: only the source rows the bilinear resize reads are copied to the
    device (into their natural places of the staged frame); every frame of the sequence differs, so a row that was
    needed but not copied would show up as stale data"""
    import orc
    from nubovca import capi, synth
    fs = capi.FaceStream(ctx, casc, width_to_process=w2p)
    ofs = orc.FaceStream(orc_cascade, width_to_process=w2p)
    seen = 0
    for i in range(6):
        f = synth.make_bgr(W, H, 4000 + 17 * i, "natural", [(W // 8 + 20 * i, H // 6, H // 2)])
        boxes, ids = fs.process(f)
        eb, eid = ofs.process(f)
        assert np.array_equal(boxes, eb), (i, boxes, eb)
        assert np.array_equal(ids, eid)
        seen += len(eb)
    assert seen > 0
    # the same through the batched entry point (chunked ingest on the copy stream)
    frames = [synth.make_bgr(W, H, 4200 + i, "natural", [(W // 8 + 10 * i, H // 6, H // 2)]) for i in range(16)]
    streams = [capi.FaceStream(ctx, casc, width_to_process=w2p) for _ in frames]
    res = ctx.face_batch_process(streams, [capi.make_frame(f) for f in frames], cap=64)
    for f, (b, _) in zip(frames, res):
        eb, _ = orc.FaceStream(orc_cascade, width_to_process=w2p).process(f)
        assert np.array_equal(b, eb)
    for st in streams:
        st.close()
    fs.close()


def test_face_batch_equals_sequential(ctx, casc, orc_cascade):
    """3 streams x 6 frames interleaved in ONE batched call == each stream run alone on the oracle."""
    import orc
    from nubovca import capi
    W, H = 640, 480
    seqs = [_sequence(W, H, 6, 1000 * s) for s in range(3)]
    streams = [capi.FaceStream(ctx, casc) for _ in range(3)]
    order = [(s, i) for i in range(6) for s in range(3)]
    res = ctx.face_batch_process([streams[s] for s, _ in order], [capi.make_frame(seqs[s][i]) for s, i in order])
    for s in range(3):
        ofs = orc.FaceStream(orc_cascade)
        for i in range(6):
            eb, eid = ofs.process(seqs[s][i])
            boxes, ids = res[order.index((s, i))]
            assert np.array_equal(boxes, eb) and np.array_equal(ids, eid)


def test_face_batch_chunked_ingest(ctx, casc, orc_cascade):
    """a large batch of HOST frames goes through in chunks (H2D of the next chunk overlaps the kernels of the current
    one, each chunk with its own candidate list / box table); device frames mixed in; min_neighbors differs per stream"""
    import orc
    import torch
    from nubovca import capi, synth
    W, H, N = 480, 360, 21
    frames = [synth.make_bgr(W, H, 300 + i, "natural", [(30 + 9 * i, 40 + (i % 5) * 20, 120 + 4 * i)] if i % 4 != 3 else []) for i in range(N)]
    props = [{"width_to_process": W, "multi_scale_factor": 10, "min_neighbors": 2 + (i % 3)} for i in range(N)]
    streams = [capi.FaceStream(ctx, casc, **props[i]) for i in range(N)]
    dev = {i: torch.from_numpy(frames[i]).cuda() for i in (5, 13)}
    torch.cuda.synchronize()
    fr = [capi.make_frame(dev[i].data_ptr(), W, H, W * 3, capi.MEM_DEVICE) if i in dev else capi.make_frame(frames[i]) for i in range(N)]
    for rep in range(2):                               # the second pass exercises the temporal state as well
        res = ctx.face_batch_process(streams, fr)
        for i in range(N):
            if rep == 0:
                kw = {"width_to_process": "width_to_process", "multi_scale_factor": "scale_factor_pct", "min_neighbors": "min_neighbors"}
                streams[i]._o = orc.FaceStream(orc_cascade, **{kw[k]: v for k, v in props[i].items()})
            eb, eid = streams[i]._o.process(frames[i])
            assert np.array_equal(res[i][0], eb) and np.array_equal(res[i][1], eid), (rep, i)
    assert sum(len(r[0]) for r in res) >= 10


def test_face_batch_mixed_geometries(ctx, casc, orc_cascade):
    """one batched call with streams of different frame sizes / properties: one launch set per distinct geometry,
    results per stream as if each ran alone; three ticks so gating and tracking state take part"""
    import orc
    from nubovca import capi, synth
    specs = [(640, 480, {}), (480, 360, {"width_to_process": 240}), (640, 480, {"process_x_every_4_frames": 2}),
             (800, 450, {"width_to_process": 400, "multi_scale_factor": 15}), (480, 360, {"width_to_process": 240}),
             (640, 480, {}), (322, 242, {"width_to_process": 161}), (800, 450, {"width_to_process": 400, "multi_scale_factor": 15})]
    kw = {"width_to_process": "width_to_process", "process_x_every_4_frames": "process_x_every_4", "multi_scale_factor": "scale_factor_pct"}
    streams = [capi.FaceStream(ctx, casc, **p) for _, _, p in specs]
    oracles = [orc.FaceStream(orc_cascade, **{kw[k]: v for k, v in p.items()}) for _, _, p in specs]
    seen = 0
    for tick in range(5):
        frames = [synth.make_bgr(W, H, 5000 + 37 * i + tick, "natural", [(W // 6 + 7 * tick, H // 7, H // 2)] if (i + tick) % 4 else [])
                  for i, (W, H, _) in enumerate(specs)]
        res = ctx.face_batch_process(streams, [capi.make_frame(f) for f in frames])
        for i in range(len(specs)):
            eb, eid = oracles[i].process(frames[i])
            assert np.array_equal(res[i][0], eb) and np.array_equal(res[i][1], eid), (tick, i)
            seen += len(eb)
    assert seen > 8


def test_face_batch_two_chunked_groups(ctx, casc, orc_cascade):
    """two geometry groups in one call, both large enough for the chunked ingest: their host frames share one staging
    buffer (the copy stream runs ahead of the kernels), pipelined as well"""
    import orc
    from nubovca import capi, synth
    geo = [(480, 360, {"width_to_process": 480, "multi_scale_factor": 15}), (400, 300, {"width_to_process": 400, "multi_scale_factor": 15})]
    kw = {"width_to_process": "width_to_process", "multi_scale_factor": "scale_factor_pct"}
    N = 17
    specs = [geo[i % 2] for i in range(2 * N)]
    streams = [capi.FaceStream(ctx, casc, **p) for _, _, p in specs]
    oracles = [orc.FaceStream(orc_cascade, **{kw[k]: v for k, v in p.items()}) for _, _, p in specs]
    frames = [[synth.make_bgr(W, H, 9000 + 13 * i + t, "natural", [(W // 6 + 3 * i, H // 7 + t, H // 2)] if i % 3 else [])
               for i, (W, H, _) in enumerate(specs)] for t in range(3)]
    res0 = ctx.face_batch_process(streams, [capi.make_frame(f) for f in frames[0]])
    t1 = ctx.face_batch_submit(streams, [capi.make_frame(f) for f in frames[1]])
    t2 = ctx.face_batch_submit(streams, [capi.make_frame(f) for f in frames[2]])
    got = [res0, ctx.face_batch_collect(t1), ctx.face_batch_collect(t2)]
    seen = 0
    for t in range(3):
        for i in range(len(specs)):
            eb, eid = oracles[i].process(frames[t][i])
            assert np.array_equal(got[t][i][0], eb) and np.array_equal(got[t][i][1], eid), (t, i)
            seen += len(eb)
    assert seen > 20


def test_face_batch_submit_collect(ctx, casc, orc_cascade):
    """two batches in flight (submit k+1 before collect k), the same streams in consecutive batches, host and device
    frames, mixed geometries: boxes and ids as if every frame went through the synchronous call"""
    import orc
    import torch
    from nubovca import capi, synth
    specs = [(640, 480, {"width_to_process": 640, "multi_scale_factor": 10}), (480, 360, {"width_to_process": 240}),
             (640, 480, {"width_to_process": 640, "multi_scale_factor": 10}), (640, 480, {"width_to_process": 320})]
    kw = {"width_to_process": "width_to_process", "multi_scale_factor": "scale_factor_pct"}
    streams = [capi.FaceStream(ctx, casc, **p) for _, _, p in specs]
    oracles = [orc.FaceStream(orc_cascade, **{kw[k]: v for k, v in p.items()}) for _, _, p in specs]
    T = 7
    frames = [[synth.make_bgr(W, H, 7000 + 31 * i + t, "natural", [(W // 5 + 9 * t, H // 6, H // 2)] if (i + t) % 5 else [])
               for i, (W, H, _) in enumerate(specs)] for t in range(T)]
    keep = [[torch.from_numpy(f).cuda() if (i + t) % 2 else None for i, f in enumerate(row)] for t, row in enumerate(frames)]
    torch.cuda.synchronize()

    def fr(t):
        return [capi.make_frame(keep[t][i].data_ptr(), specs[i][0], specs[i][1], specs[i][0] * 3, capi.MEM_DEVICE)
                if keep[t][i] is not None else capi.make_frame(frames[t][i]) for i in range(len(specs))]

    got = []
    pending = ctx.face_batch_submit(streams, fr(0))
    for t in range(1, T):
        nxt = ctx.face_batch_submit(streams, fr(t))
        got.append(ctx.face_batch_collect(pending))
        pending = nxt
    with pytest.raises(capi.NvcaError):
        ctx.face_batch_process(streams, fr(0))                 # synchronous call while a batch is in flight: refused
    g = orc.equalize_hist(synth.make_gray(320, 240, 5, "natural", [(60, 40, 120)]))
    assert np.array_equal(ctx.detect_multiscale(casc, g, 1.2, 3), orc.detect_multiscale(orc_cascade, g, 1.2, 3))   # other entry points may run meanwhile
    got.append(ctx.face_batch_collect(pending))
    seen = 0
    for t in range(T):
        for i in range(len(specs)):
            eb, eid = oracles[i].process(frames[t][i])
            assert np.array_equal(got[t][i][0], eb) and np.array_equal(got[t][i][1], eid), (t, i)
            seen += len(eb)
    assert seen > 8
    res = ctx.face_batch_process(streams, fr(0))                # and the synchronous form works again afterwards
    assert len(res) == len(specs)


def test_face_batch_registered_host_frames(ctx, casc, orc_cascade):
    """page-locked (nvca_host_register) host frames: asynchronous H2D in the chunked ingest path, same boxes"""
    import orc
    from nubovca import capi, synth
    W, H, N = 400, 300, 18
    frames = [synth.make_bgr(W, H, 900 + i, "natural", [(20 + 10 * i, 30 + (i % 4) * 25, 110 + 5 * i)]) for i in range(N)]
    for f in frames:
        ctx.host_register(f)
    try:
        streams = [capi.FaceStream(ctx, casc, width_to_process=W, multi_scale_factor=10) for _ in range(N)]
        res = ctx.face_batch_process(streams, [capi.make_frame(f) for f in frames])
        for i in range(N):
            eb, eid = orc.FaceStream(orc_cascade, width_to_process=W, scale_factor_pct=10).process(frames[i])
            assert np.array_equal(res[i][0], eb) and np.array_equal(res[i][1], eid), i
    finally:
        for f in frames:
            ctx.host_unregister(f)


@pytest.mark.parametrize("w2p", [160, 640])
@pytest.mark.parametrize("msf", [1, 2, 4])
def test_face_stream_every_legal_multi_scale_factor(ctx, casc, orc_cascade, w2p, msf):
    """multi-scale-factor is installed with range 0 .. 51 and no clamp (FACE/kmsfacedetect.cpp:540-542, 1084-1087): 1, 2 and 4 are ladders
    of 73 .. 290 scales on the 160 x 90 and 640 x 360 working images -- more than the 63 a candidate key used to hold.  The key's
    fields are sized per plan now; boxes and ids against the oracle.  Value 0 (scaleFactor 1.0: OpenCV's assertion) stays an error."""
    import orc
    from nubovca import capi, synth
    W, H = 640, 360
    frames = [synth.make_bgr(W, H, 4300 + i, "natural", [(60 + 12 * i, 40, 150 + 6 * i), (380, 120 + 5 * i, 110)]) for i in range(3)]
    fs = capi.FaceStream(ctx, casc, width_to_process=w2p, multi_scale_factor=msf)
    ofs = orc.FaceStream(orc_cascade, width_to_process=w2p, scale_factor_pct=msf)
    seen = 0
    for f in frames:
        b, ids = fs.process(f)
        eb, eid = ofs.process(f)
        assert np.array_equal(b, eb) and np.array_equal(ids, eid), (w2p, msf, b, eb)
        seen += len(eb)
    assert seen >= 2
    res = ctx.face_batch_process([fs] * 3, [capi.make_frame(f) for f in frames])       # the batched entry point on the same ladder
    for i, f in enumerate(frames):
        eb, eid = ofs.process(f)
        assert np.array_equal(res[i][0], eb) and np.array_equal(res[i][1], eid), (w2p, msf, i)
    fs.close()
    bad = capi.FaceStream(ctx, casc, width_to_process=w2p, multi_scale_factor=0)
    with pytest.raises(capi.NvcaError) as ei:
        bad.process(frames[0])
    assert ei.value.code == capi.ERR_ARG
    bad.close()


def test_face_stream_device_frames(ctx, casc, orc_cascade):
    """frames already resident in HBM (torch tensors) give the same boxes as host frames."""
    import orc
    import torch
    from nubovca import capi
    W, H = 640, 480
    frames = _sequence(W, H, 4, 7)
    dev = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    fs = capi.FaceStream(ctx, casc)
    ofs = orc.FaceStream(orc_cascade)
    fr = [capi.make_frame(d.data_ptr(), W, H, W * 3, capi.MEM_DEVICE) for d in dev]
    res = ctx.face_batch_process([fs] * 4, fr)
    for i in range(4):
        eb, eid = ofs.process(frames[i])
        assert np.array_equal(res[i][0], eb) and np.array_equal(res[i][1], eid)


# ------------------------------------------------------------------ SCALE_IMAGE / FIND_BIGGEST variants
VARIANT_CASES = [
    (200, 160, "natural", [(40, 30, 80)], 1.1, (20, 20), 9),
    (97, 83, "natural", [(10, 8, 60)], 1.1, (3, 3), 4),
    (320, 180, "natural", [(100, 20, 120), (10, 60, 50)], 1.25, (3, 3), 5),
    (160, 120, "noise", [(30, 20, 70)], 1.1, (1, 1), 6),
    (64, 48, "gradient", [(8, 4, 40)], 1.1, (0, 0), 7),
    (25, 25, "natural", [], 1.1, (0, 0), 8),
]


# images whose integral pair fits a workgroup's LDS take the one-launch small-image path (kernels_roi.hip); roi=0 sends them
# through the large-image path (plan, pre-pass, tiles) like every bigger image: both must agree with the oracle
@pytest.mark.parametrize("roi", [1, 0])
@pytest.mark.parametrize("w,h,kind,faces,sf,ms,seed", VARIANT_CASES)
def test_detect_scale_image(ctx, casc, orc_cascade, w, h, kind, faces, sf, ms, seed, roi):
    import orc
    from nubovca import capi, synth
    g = orc.equalize_hist(synth.make_gray(w, h, seed, kind, faces))
    with ctx.options(roi=roi):
        raw = ctx.detect_raw(casc, g, sf, capi.HAAR_SCALE_IMAGE, ms)
        eraw = orc.detect_raw(orc_cascade, g, sf, orc.HAAR_SCALE_IMAGE, ms)
        assert np.array_equal(raw, eraw), (len(raw), len(eraw))
        for mn in (2, 3):
            det = ctx.detect_multiscale(casc, g, sf, mn, capi.HAAR_SCALE_IMAGE, ms)
            assert np.array_equal(det, orc.detect_multiscale(orc_cascade, g, sf, mn, orc.HAAR_SCALE_IMAGE, ms))


@pytest.mark.parametrize("roi", [1, 0])
@pytest.mark.parametrize("w,h,kind,faces,sf,ms,seed", VARIANT_CASES)
def test_detect_find_biggest(ctx, casc, orc_cascade, w, h, kind, faces, sf, ms, seed, roi):
    import orc
    from nubovca import capi, synth
    g = orc.equalize_hist(synth.make_gray(w, h, seed, kind, faces))
    with ctx.options(roi=roi):
        for flags in (capi.HAAR_FIND_BIGGEST_OBJECT, capi.HAAR_FIND_BIGGEST_OBJECT | capi.HAAR_DO_ROUGH_SEARCH,
                      capi.HAAR_FIND_BIGGEST_OBJECT | capi.HAAR_SCALE_IMAGE):
            det = ctx.detect_multiscale(casc, g, sf, 3, flags, ms)
            exp = orc.detect_multiscale(orc_cascade, g, sf, 3, flags, ms)
            assert np.array_equal(det, exp), (flags, det, exp)


def test_small_image_path_random_geometries(ctx, casc, orc_cascade, casc_small, orc_small):
    """the one-launch small-image detector on random sizes up to its limit ((w + 1)(h + 2) <= 10240 words), every scan variant,
    raw lists in OpenCV's order: windows at the image border, rows longer and shorter than a 64-window chunk (the adaptive x
    step's parity is carried across chunks), pyramid levels down to one window"""
    import orc
    from nubovca import capi, synth
    rng = np.random.RandomState(99)
    launched = 0
    ctx.enable_kernel_timing(1)
    for it in range(40):
        w = int(rng.randint(21, 200))
        h = int(rng.randint(21, min(200, 10240 // (w + 1) - 2) + 1))
        s = int(min(w, h) * rng.uniform(0.4, 0.9))
        faces = [(int(rng.randint(0, max(1, w - s))), int(rng.randint(0, max(1, h - s))), s)] if s >= 24 else []
        g = orc.equalize_hist(synth.make_gray(w, h, 3000 + it, ["natural", "noise", "gradient"][it % 3], faces))
        c, oc = (casc_small, orc_small) if it % 2 else (casc, orc_cascade)
        sf = float(rng.choice([1.1, 1.2, 1.25]))
        ms = (int(rng.randint(0, 30)), int(rng.randint(0, 30))) if it % 3 else (0, 0)
        assert np.array_equal(ctx.detect_raw(c, g, sf, 0, ms), orc.detect_raw(oc, g, sf, 0, ms)), (it, w, h, sf, ms)
        assert np.array_equal(ctx.detect_raw(c, g, sf, capi.HAAR_SCALE_IMAGE, ms), orc.detect_raw(oc, g, sf, orc.HAAR_SCALE_IMAGE, ms)), (it, w, h, sf, ms)
        for fl in (0, capi.HAAR_SCALE_IMAGE, capi.HAAR_FIND_BIGGEST_OBJECT):
            assert np.array_equal(ctx.detect_multiscale(c, g, sf, 2, fl, ms), orc.detect_multiscale(oc, g, sf, 2, fl, ms)), (it, w, h, sf, ms, fl)
    kt = ctx.kernel_timing()
    ctx.enable_kernel_timing(0)
    assert kt.get("cascade_roi", (0, 0))[1] >= 150 and "cascade_tile" not in kt, kt          # every one of these calls was ONE k_roi launch per round


def test_detect_roi_view(ctx, casc, orc_cascade):
    """detectMultiScale on a sub-matrix (pointer + parent stride), as the part detectors call it"""
    import ctypes as C
    import orc
    from nubovca import capi, synth
    full = orc.equalize_hist(synth.make_gray(320, 240, 21, "natural", [(90, 60, 100)]))
    x0, y0, rw, rh = 70, 40, 160, 150
    roi = full[y0:y0 + rh, x0:x0 + rw]
    exp = orc.detect_multiscale(orc_cascade, np.ascontiguousarray(roi), 1.1, 2, orc.HAAR_SCALE_IMAGE, (20, 20))
    buf = (capi.Rect * 64)()
    n = C.c_int()
    ctx.check(ctx.L.nvca_detect_multiscale(ctx.h, casc.h, full.ctypes.data + y0 * full.strides[0] + x0, rw, rh,
                                           full.strides[0], capi.MEM_HOST, 1.1, 2, capi.HAAR_SCALE_IMAGE, 20, 20, 0, 0,
                                           buf, 64, C.byref(n)))
    got = np.array([[buf[i].x, buf[i].y, buf[i].w, buf[i].h] for i in range(n.value)], np.int32).reshape(-1, 4)
    assert len(exp) >= 1 and np.array_equal(got, exp)


# ------------------------------------------------------------------ optional evaluator paths stay correct
@pytest.mark.parametrize("opts", [{"tiles": 0}, {"deep_stage": 1},
                                  {"deep_stage": 2}, {"deep_stage": 30}, {"tiles": 0, "deep_stage": 30},
                                  {"band": 1}, {"band": 1, "deep_stage": 30}, {"band": 1, "deep_stage": 2},
                                  {"deep_lds": 0}, {"deep_stage": 20}])
def test_optional_evaluator_paths(ctx, casc, orc_cascade, opts):
    """k_strip (row strips, global gathers: the fallback of plans without tiles) and other deep-stage splits are kept as
    measured alternatives; nvca_ctx_set_option switches them per context (plan options drop the cached plans)"""
    import orc
    from nubovca import synth
    w, h = 1003, 611
    g = orc.equalize_hist(synth.make_gray(w, h, 77, "natural", [(200, 100, 260), (600, 300, 120)]))
    eraw = orc.detect_raw(orc_cascade, g, 1.1, 0, (40, 40))
    with ctx.options(**opts):
        raw = ctx.detect_raw(casc, g, 1.1, 0, (40, 40))
    assert len(eraw) > 0 and np.array_equal(raw, eraw)
    assert np.array_equal(ctx.detect_raw(casc, g, 1.1, 0, (40, 40)), eraw)          # and back on the default path


def test_unknown_option_is_refused(ctx):
    from nubovca import capi
    with pytest.raises(capi.NvcaError):
        ctx.set_option("no_such_switch", 1)


def test_detect_long_scan_rows(ctx, casc, orc_cascade):
    """minSize 0 on a wide image: scan rows of > 512 windows are cut into strip segments"""
    import orc
    from nubovca import capi, synth
    g = orc.equalize_hist(synth.make_gray(1400, 300, 31, "natural", [(300, 60, 150), (900, 100, 60)]))
    assert np.array_equal(ctx.detect_raw(casc, g, 1.2, 0, (0, 0)), orc.detect_raw(orc_cascade, g, 1.2, 0, (0, 0)))
    assert np.array_equal(ctx.detect_raw(casc, g, 1.2, capi.HAAR_SCALE_IMAGE, (0, 0)),
                          orc.detect_raw(orc_cascade, g, 1.2, orc.HAAR_SCALE_IMAGE, (0, 0)))


def test_detect_4k(ctx, casc, orc_cascade):
    """3840x2160 (w + 1 > 2048: multi-pass integral rows; sum close to the int32 limit)"""
    import orc
    from nubovca import synth
    g = orc.equalize_hist(synth.make_gray(3840, 2160, 41, "natural", [(500, 300, 700), (2500, 900, 400)]))
    raw = ctx.detect_raw(casc, g, 1.2, 0, (192, 108))
    eraw = orc.detect_raw(orc_cascade, g, 1.2, 0, (192, 108))
    assert len(eraw) > 0 and np.array_equal(raw, eraw)


def test_kernel_timing_every_batch_and_sampled(ctx, casc, orc_cascade):
    """nvca_ctx_enable_kernel_timing: 1 brackets every launch, N > 1 the launches of every N-th face batch (the first one
    included); results are the same either way"""
    import orc
    from nubovca import capi, synth
    W, H = 640, 480
    frames = [synth.make_bgr(W, H, 5100 + i, "natural", [(60 + 9 * i, H // 6, H // 2)]) for i in range(6)]
    exp = []
    ofs = orc.FaceStream(orc_cascade)
    for f in frames:
        exp.append(ofs.process(f)[0])
    for mode, want in ((1, 6), (3, 2)):
        fs = capi.FaceStream(ctx, casc)
        ctx.enable_kernel_timing(mode)
        for f, e in zip(frames, exp):
            b, _ = fs.process(f)
            assert np.array_equal(b, e)
        kt = ctx.kernel_timing()
        ctx.enable_kernel_timing(0)
        fs.close()
        assert kt["gray_resize_hist"][1] == want, (mode, kt)
        assert kt["gray_resize_hist"][0] > 0.0
    assert all(v[1] == 0 for v in ctx.kernel_timing().values())      # off: nothing accumulates


def test_stump_table_cache_turns_over(ctx, casc_small, orc_small):
    """Stump tables are cached per (cascade, factor), 768 of them a context; beyond that the least recently used ones that nobody
    references are dropped in groups.  Fifty scale factors on one image walk ~1 500 distinct factors: the cache turns over twice
    while calls keep being answered, and a factor that comes round again is rebuilt -- every answer against the oracle's."""
    import orc
    from nubovca import synth
    g = orc.equalize_hist(synth.make_gray(160, 120, 17, "natural", [(20, 15, 80)]))
    sfs = [1.02 + 0.0013 * k for k in range(50)]
    hits = 0
    for sf in sfs + sfs[:3]:
        raw = ctx.detect_raw(casc_small, g, sf, 0, (0, 0))
        eraw = orc.detect_raw(orc_small, g, sf, 0, (0, 0))
        assert np.array_equal(raw, eraw), (sf, len(raw), len(eraw))
        hits += len(eraw)
    assert hits > 0
