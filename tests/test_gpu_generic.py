"""General cascades on the device: tilted features (haarcascade_profileface.xml, EAR/kmseardetect.cpp:29) and
tree-structured weak classifiers (the haarcascade_mcs_* files, EYE/kmseyedetect.cpp:27-29, NOSE/kmsnosedetect.cpp:31-32,
MOUTH/kmsmouthdetect.cpp:37-38) -- SURVEY.md A.6 marks the structure of those files (U), so both forms must load and
evaluate bit-exactly.  Oracle: orc_haar.c's general branch (tree walk, tilted integral), itself pinned by hand-derived
answers in tests/test_oracle_haar.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from nubovca import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("h,w", [(1, 1), (1, 7), (9, 1), (7, 5), (64, 64), (90, 160), (180, 320), (300, 1023), (200, 1024), (240, 1500), (1080, 1920)])
def test_tilted_integral(ctx, h, w):
    import orc
    img = np.random.default_rng(h * 7 + w).integers(0, 256, size=(h, w)).astype(np.uint8)
    assert np.array_equal(ctx.integral_tilted(img), orc.integral_tilted(img))


KINDS = {
    "tilted_stumps": dict(tilt_frac=0.4, tree_frac=0.0),          # the SSE2 pair policy still applies to two-rect stages
    "trees": dict(tilt_frac=0.0, tree_frac=0.5),
    "tilted_trees": dict(tilt_frac=0.3, tree_frac=0.4),
}


@pytest.mark.parametrize("kind", sorted(KINDS))
@pytest.mark.parametrize("ow,oh", [(20, 20), (25, 15), (12, 20)])
def test_generic_cascade_all_variants(ctx, kind, ow, oh):
    """scale-cascade scan (adaptive x step), SCALE_IMAGE (tilted integral per pyramid level), FIND_BIGGEST and its rough
    search; raw lists identical in order, grouped boxes identical; both accumulation policies"""
    import orc
    from nubovca import capi, synth
    xml = synth.generic_cascade_xml(ow=ow, oh=oh, seed=ow * 100 + oh + len(kind), **KINDS[kind])
    c, oc = ctx.load_cascade_xml(xml), orc.parse_cascade_xml(xml)
    assert c.kind() == (bool(oc.tilted.any()), bool((oc.cls_nnodes > 1).any()))
    total = 0
    for it, (w, h, content, sf) in enumerate([(333, 251, "natural", 1.1), (200, 150, "noise", 1.2), (320, 180, "gradient", 1.25)]):
        g = orc.equalize_hist(synth.make_gray(w, h, 50 + it, content))
        eraw = orc.detect_raw(oc, g, sf, 0, (0, 0))
        assert np.array_equal(ctx.detect_raw(c, g, sf, 0, (0, 0)), eraw), (kind, ow, oh, w, h)
        total += len(eraw)
        assert np.array_equal(ctx.detect_multiscale(c, g, sf, 2, 0, (ow + 5, oh + 3)), orc.detect_multiscale(oc, g, sf, 2, 0, (ow + 5, oh + 3)))
        esi = orc.detect_raw(oc, g, sf, orc.HAAR_SCALE_IMAGE, (3, 3))
        assert np.array_equal(ctx.detect_raw(c, g, sf, capi.HAAR_SCALE_IMAGE, (3, 3)), esi)
        total += len(esi)
        assert np.array_equal(ctx.detect_multiscale(c, g, sf, 2, capi.HAAR_SCALE_IMAGE, (3, 3)),
                              orc.detect_multiscale(oc, g, sf, 2, orc.HAAR_SCALE_IMAGE, (3, 3)))
        for fl in (capi.HAAR_FIND_BIGGEST_OBJECT, capi.HAAR_FIND_BIGGEST_OBJECT | capi.HAAR_DO_ROUGH_SEARCH):
            assert np.array_equal(ctx.detect_multiscale(c, g, sf, 3, fl, (1, 1)), orc.detect_multiscale(oc, g, sf, 3, fl, (1, 1)))
    assert total > 5
    ctx.set_sum_policy(capi.SUM_F64)
    try:
        g = orc.equalize_hist(synth.make_gray(300, 200, 9, "natural"))
        assert np.array_equal(ctx.detect_raw(c, g, 1.1, 0, (0, 0)), orc.detect_raw(oc, g, 1.1, 0, (0, 0), policy=orc.SUM_F64))
    finally:
        ctx.set_sum_policy(capi.SUM_F32PAIR)


def test_generic_cascade_large_image_and_batch(ctx):
    """a tilted + tree cascade as the FACE cascade of NuboFaceDetector streams: 720p working image, batched call with
    different content per stream, temporal state over three ticks"""
    import orc
    from nubovca import capi, synth
    xml = synth.generic_cascade_xml(seed=77, stage_sizes=(3, 8, 12, 16, 20, 24, 28, 30))
    c, oc = ctx.load_cascade_xml(xml), orc.parse_cascade_xml(xml)
    W, H, S = 1280, 720, 4
    streams = [capi.FaceStream(ctx, c, width_to_process=W // 2, multi_scale_factor=20) for _ in range(S)]
    oracles = [orc.FaceStream(oc, width_to_process=W // 2, scale_factor_pct=20) for _ in range(S)]
    seen = 0
    for t in range(3):
        frames = [synth.make_bgr(W, H, 300 + 10 * s + t, ["natural", "gradient", "noise", "natural"][s]) for s in range(S)]
        res = ctx.face_batch_process(streams, [capi.make_frame(f) for f in frames])
        for s in range(S):
            eb, eid = oracles[s].process(frames[s])
            assert np.array_equal(res[s][0], eb) and np.array_equal(res[s][1], eid), (t, s, res[s][0], eb)
            seen += len(eb)
    assert seen > 20          # the oracle finds 53 boxes over these twelve frames: the comparison is not [] == []
    for st in streams:
        st.close()


@pytest.mark.parametrize("kind", ["eye", "nose", "mouth", "ear"])
def test_part_streams_with_generic_cascades(ctx, kind):
    """the part detectors with tree / tilted cascades in every role (profile-face for the ear element included): the chain
    gray -> face pass -> ROI passes goes through SCALE_IMAGE and FIND_BIGGEST on general cascades"""
    import orc
    from nubovca import capi, synth
    K = {"eye": capi.PART_EYE, "nose": capi.PART_NOSE, "mouth": capi.PART_MOUTH, "ear": capi.PART_EAR}[kind]
    OK = {"eye": orc.PART_EYE, "nose": orc.PART_NOSE, "mouth": orc.PART_MOUTH, "ear": orc.PART_EAR}[kind]
    fx = synth.generic_cascade_xml(seed=11, stage_sizes=(3, 6, 9, 12, 15), tilt_frac=0.3, tree_frac=0.3)
    ax = synth.generic_cascade_xml(ow=18, oh=12, seed=12, stage_sizes=(3, 6, 9, 12), tilt_frac=0.2, tree_frac=0.5)
    bx = synth.generic_cascade_xml(ow=12, oh=20, seed=13, stage_sizes=(3, 6, 9, 12), tilt_frac=0.4, tree_frac=0.2)
    face, a, b = (ctx.load_cascade_xml(x) for x in (fx, ax, bx))
    oface, oa, ob = (orc.parse_cascade_xml(x) for x in (fx, ax, bx))
    ps = capi.PartStream(ctx, K, face, a, b)
    ops = orc.PartStream(OK, oface, oa, ob)
    seen = 0
    for i in range(5):
        f = synth.make_bgr(640, 480, 900 + i, "natural")
        ga, gb = ps.process(f)
        ea, eb = ops.process(f)
        assert np.array_equal(ga, ea) and np.array_equal(gb, eb), (kind, i, ga, ea, gb, eb)
        seen += len(ea) + len(eb)
    assert seen > 0, kind     # eye 3, nose 41, mouth 45, ear 91 boxes on the oracle side: every kind compares real lists
    ps.close()


def test_loader_rejects_tilted_rect_outside_window(ctx):
    from nubovca import capi, synth
    bad = dict(name="t", size=(12, 12), stages=[dict(features=[[(2, 1, 4, 3, -1.0), (2, 1, 2, 3, 2.0)]], tilted=[1],      # x - h < 0
                                                     thresholds=[0.0], left=[-1.0], right=[1.0], stage_threshold=0.0)])
    with pytest.raises(capi.NvcaError) as e:
        ctx.load_cascade_xml(synth.cascade_to_xml(bad))
    assert e.value.code == capi.ERR_PARSE


def test_handwritten_oldformat_files(ctx):
    """tests/golden/oldformat_*.xml (cvSave's layout, written by hand): the product's loader reads the same cascade out of
    them as the oracle's (an independent XML parser), and every scan variant agrees on an image"""
    import os
    import orc
    from nubovca import capi, synth
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sx = open(os.path.join(gold, "oldformat_stumps_24x24.xml")).read()
    tx = open(os.path.join(gold, "oldformat_trees_tilted_20x20.xml")).read()
    cs, ocs = ctx.load_cascade_xml(sx), orc.parse_cascade_xml(sx)
    d = cs.dump()
    assert d["size"] == (24, 24) and list(d["stage_sizes"]) == [2, 3, 2]
    assert np.array_equal(d["rects"], np.asarray(ocs.rects)) and np.array_equal(d["weights"], np.asarray(ocs.rweights))
    assert np.array_equal(d["thr"], np.asarray(ocs.node_thr)) and np.array_equal(d["stage_thr"], np.asarray(ocs.stage_thr))
    assert np.array_equal(d["left"], np.asarray(ocs.alpha)[0::2]) and np.array_equal(d["right"], np.asarray(ocs.alpha)[1::2])
    ct, oct_ = ctx.load_cascade_xml(tx), orc.parse_cascade_xml(tx)
    assert ct.kind() == (True, True) and cs.kind() == (False, False) and ct.info() == (20, 20, 2, 4)
    total = 0
    for c, oc in ((cs, ocs), (ct, oct_)):
        for it, (w, h, sf) in enumerate([(160, 120, 1.2), (97, 83, 1.1)]):
            g = orc.equalize_hist(synth.make_gray(w, h, 21 + it, "natural"))
            for fl, ofl in ((0, 0), (capi.HAAR_SCALE_IMAGE, orc.HAAR_SCALE_IMAGE)):
                e = orc.detect_raw(oc, g, sf, ofl, (0, 0), cap=1 << 18)
                assert np.array_equal(ctx.detect_raw(c, g, sf, fl, (0, 0)), e), (w, h, fl, len(e))
                total += len(e)
            assert np.array_equal(ctx.detect_multiscale(c, g, sf, 3, capi.HAAR_FIND_BIGGEST_OBJECT, (1, 1)),
                                  orc.detect_multiscale(oc, g, sf, 3, orc.HAAR_FIND_BIGGEST_OBJECT, (1, 1)))
    assert total > 100
