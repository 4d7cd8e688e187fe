"""The configurations bench.py times, checked against the oracle AT THE TIMED SIZE and through the same entry point:
BASELINE.json configs[1] (one 1080p stream, consecutive frames in one nvca_face_batch_process call: the batch is large
enough for the band kernel k_band), configs[3] (32 x 720p streams per GPU in one batch) and configs[4] (8 x 1080p streams
through NuboFaceDetector + NuboTracker).  Reference path: FACE/kmsfacedetect.cpp:805-826, TRK/gstnubotracker.cpp:339-421.
Which cascade kernel ran is read back from the per-kernel timers, so a silent switch of evaluator cannot hide here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FACES_1080 = [(200, 150, 300), (900, 400, 180), (1400, 100, 120), (1500, 700, 240)]


@pytest.fixture(scope="module")
def ctx():
    from nubovca import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def casc(ctx, synth_xml):
    return ctx.load_cascade_xml(synth_xml)


def _device_frames(frames):
    import torch
    from nubovca import capi
    keep = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    H, W = frames[0].shape[:2]
    return keep, [capi.make_frame(t.data_ptr(), W, H, W * frames[0].shape[2], capi.MEM_DEVICE) for t in keep]


def _launched(kt, name):
    return kt.get(name, (0.0, 0))[1]


def test_face_batch_1080p_band_kernel_vs_oracle(ctx, casc, orc_cascade):
    """configs[1] as bench.py runs it: 12 device-resident consecutive 1920x1080 frames of ONE stream, full-resolution mode,
    sf 1.1, one nvca_face_batch_process call, no environment override.  68 bands x 12 frames >= 640 -> k_band."""
    import orc
    from nubovca import capi, synth
    W, H, N = 1920, 1080, 12
    frames = [synth.make_bgr(W, H, synth.frame_seed(0, i), "natural", [(x + 8 * i, y, s) for (x, y, s) in FACES_1080] if i % 5 != 3 else [])
              for i in range(N)]
    keep, fr = _device_frames(frames)
    props = dict(width_to_process=W, multi_scale_factor=10)
    fs = capi.FaceStream(ctx, casc, **props)
    ofs = orc.FaceStream(orc_cascade, width_to_process=W, scale_factor_pct=10)
    ctx.enable_kernel_timing(1)
    seen = 0
    for rep in range(2):                      # the second call exercises the carried temporal state
        res = ctx.face_batch_process([fs] * N, fr)
        for i in range(N):
            eb, eid = ofs.process(frames[i])
            assert np.array_equal(res[i][0], eb), (rep, i, res[i][0], eb)
            assert np.array_equal(res[i][1], eid), (rep, i)
            seen += len(eb)
    kt = ctx.kernel_timing()
    ctx.enable_kernel_timing(0)
    assert _launched(kt, "cascade_band") == 2 and _launched(kt, "cascade_tile") == 0, kt
    assert seen >= 4 * 2 * (N - 3)
    fs.close()


def test_face_batch_1080p_calibrated_cascade_vs_oracle(ctx, calibrated_xml):
    """the headline workload with the headline CASCADE (bench.py --cascade calibrated: every early stage lets about half of what
    reaches it through, so thousands of windows per frame are still alive behind stage 5 and the tile kernels' late-stage walk
    carries real load): 8 consecutive 1080p frames through k_band, then the same frames with the late stages split off to
    k_deep (deep_stage 6) and through the per-tile kernel pair -- boxes and ids against the oracle every time"""
    import orc
    from nubovca import capi, synth
    W, H, N = 1920, 1080, 8
    casc = ctx.load_cascade_xml(calibrated_xml)
    oc = orc.parse_cascade_xml(calibrated_xml)
    frames = [synth.make_bgr(W, H, synth.frame_seed(0, i), "natural", [(x + 8 * i, y, s) for (x, y, s) in FACES_1080] if i % 5 != 3 else [])
              for i in range(N)]
    keep, fr = _device_frames(frames)
    ofs = orc.FaceStream(oc, width_to_process=W, scale_factor_pct=10)
    exp = [ofs.process(f) for f in frames]
    assert sum(len(b) for b, _ in exp) >= 4 * (N - 2)
    for opts, kern in (({"band": 1}, "cascade_band"), ({"band": 1, "deep_stage": 6}, "cascade_band"), ({"band": 0}, "cascade_tile"), ({"band": 0, "deep_stage": 8}, "cascade_tile")):
        fs = capi.FaceStream(ctx, casc, width_to_process=W, multi_scale_factor=10)
        ctx.enable_kernel_timing(1)
        with ctx.options(**opts):
            res = ctx.face_batch_process([fs] * N, fr)
        kt = ctx.kernel_timing()
        ctx.enable_kernel_timing(0)
        assert _launched(kt, kern) == 1, (opts, kt)
        assert (_launched(kt, "cascade_deep") > 0) == ("deep_stage" in opts), (opts, kt)
        for i in range(N):
            assert np.array_equal(res[i][0], exp[i][0]) and np.array_equal(res[i][1], exp[i][1]), (opts, i, res[i][0], exp[i][0])
        fs.close()


def test_face_batch_1080p_serving_loop_vs_oracle(ctx, casc, orc_cascade):
    """the loop bench.py times by default: nvca_face_batch_submit / _collect with two batches of the ONE stream in flight
    (batch k + 1 is queued before batch k is unpacked; the temporal logic of a stream runs in collect order), 10 frames of
    1920x1080 per batch -> k_band; boxes and ids of every batch against the oracle fed the same frames in the same order"""
    import orc
    from nubovca import capi, synth
    W, H, N, B = 1920, 1080, 10, 4
    sets = [[synth.make_bgr(W, H, synth.frame_seed(7, 10 * b + i), "natural", [(x + 8 * i + 40 * b, y, s) for (x, y, s) in FACES_1080] if (i + b) % 6 != 3 else [])
             for i in range(N)] for b in range(2)]
    dev = [_device_frames(fs_) for fs_ in sets]
    fs = capi.FaceStream(ctx, casc, width_to_process=W, multi_scale_factor=10)
    ofs = orc.FaceStream(orc_cascade, width_to_process=W, scale_factor_pct=10)
    ctx.enable_kernel_timing(1)
    pending = ctx.face_batch_submit([fs] * N, dev[0][1])
    seen = 0
    for b in range(1, B + 1):
        nxt = ctx.face_batch_submit([fs] * N, dev[b % 2][1]) if b < B else None
        res = ctx.face_batch_collect(pending)
        for i in range(N):
            eb, eid = ofs.process(sets[(b - 1) % 2][i])
            assert np.array_equal(res[i][0], eb) and np.array_equal(res[i][1], eid), (b, i, res[i][0], eb)
            seen += len(eb)
        pending = nxt
    kt = ctx.kernel_timing()
    ctx.enable_kernel_timing(0)
    assert _launched(kt, "cascade_band") == B and _launched(kt, "cascade_tile") == 0, kt
    assert seen > 4 * B * (N - 4)
    fs.close()


def test_face_batch_32x720p_streams_vs_oracle(ctx, casc, orc_cascade):
    """configs[3], one GPU's shard: 32 streams of 1280x720, one frame each per tick, one batched call per tick"""
    import orc
    from nubovca import capi, synth
    W, H, S, T = 1280, 720, 32, 2
    props = dict(width_to_process=W, multi_scale_factor=10)
    streams = [capi.FaceStream(ctx, casc, **props) for _ in range(S)]
    oracles = [orc.FaceStream(orc_cascade, width_to_process=W, scale_factor_pct=10) for _ in range(S)]
    ctx.enable_kernel_timing(1)
    seen = 0
    for t in range(T):
        frames = [synth.make_bgr(W, H, synth.frame_seed(s, t), "natural",
                                 [(120 + 16 * (s % 7) + 8 * t, 100, 200), (600 + 5 * s, 300 + t, 120)] if (s + t) % 6 else [])
                  for s in range(S)]
        keep, fr = _device_frames(frames)
        res = ctx.face_batch_process(streams, fr)
        for s in range(S):
            eb, eid = oracles[s].process(frames[s])
            assert np.array_equal(res[s][0], eb) and np.array_equal(res[s][1], eid), (t, s, res[s][0], eb)
            seen += len(eb)
    kt = ctx.kernel_timing()
    ctx.enable_kernel_timing(0)
    assert _launched(kt, "cascade_band") == T, kt
    assert seen > S
    for st in streams:
        st.close()


def test_face_tracker_batch_8x1080p_vs_oracle(ctx, casc, orc_cascade):
    """configs[4], one GPU's shard: 8 streams of 1920x1080 through NuboFaceDetector (BGR) and NuboTracker (BGRA of the
    same field), one batched call each per tick; moving templates give the tracker something to segment"""
    import orc
    import torch
    from nubovca import capi, synth
    W, H, S, T = 1920, 1080, 8, 3
    props = dict(width_to_process=W, multi_scale_factor=10)
    streams = [capi.FaceStream(ctx, casc, **props) for _ in range(S)]
    oracles = [orc.FaceStream(orc_cascade, width_to_process=W, scale_factor_pct=10) for _ in range(S)]
    trackers = [capi.Tracker(ctx) for _ in range(S)]
    otrk = [orc.Tracker() for _ in range(S)]
    bgs = [synth.make_gray(W, H, synth.frame_seed(s, 0), "natural") for s in range(S)]
    seen_f = seen_t = 0
    for t in range(T):
        grays = [synth.paste_faces(bgs[s], [(x + 8 * t + 16 * (s % 7), y, sz) for (x, y, sz) in FACES_1080], s) for s in range(S)]
        bgr = [synth.gray_to_bgr(g, synth.frame_seed(s, 0)) for s, g in enumerate(grays)]
        bgra = [synth.gray_to_bgr(g, synth.frame_seed(s, 0), 4) for s, g in enumerate(grays)]
        keep, fr = _device_frames(bgr)
        keep4, fr4 = _device_frames(bgra)
        res = ctx.face_batch_process(streams, fr)
        tres = capi.tracker_batch_process(ctx, trackers, fr4, [33.3 * (t + 1)] * S, cap=256)
        for s in range(S):
            eb, eid = oracles[s].process(bgr[s])
            assert np.array_equal(res[s][0], eb) and np.array_equal(res[s][1], eid), (t, s)
            et = otrk[s].process(bgra[s], 33.3 * (t + 1))
            assert np.array_equal(tres[s], et), (t, s, len(tres[s]), len(et))
            seen_f += len(eb); seen_t += len(et)
        del keep, keep4
        torch.cuda.synchronize()
    assert seen_f >= 4 * S and seen_t > 0
    for st in streams:
        st.close()
    for tr in trackers:
        tr.close()


@pytest.mark.parametrize("env,N", [({"band": 1}, 5), ({"band": 1, "band_map": 1}, 8),
                                   ({"band": 1, "band_map": 2}, 16),
                                   # both tile kernels with the late stages split off to k_deep (plans whose tiles cannot hold every
                                   # stage's samples) and walking the whole cascade themselves (the default)
                                   ({"band": 1, "deep_stage": 6}, 6), ({"band": 0, "deep_stage": 6}, 6), ({"band": 0}, 6)])
def test_band_kernel_batched_slots(ctx, casc, orc_cascade, env, N):
    """k_band decodes (band, frame slot) from the block index (and NVCA_BAND_MAP remaps it): frames of DIFFERENT content
    in one geometry, forced through the band kernel, each checked against the oracle (a slot / plane mix-up would
    swap or smear boxes between frames)"""
    import orc
    from nubovca import capi, synth
    W, H = 800, 450
    frames = [synth.make_bgr(W, H, 8100 + 7 * i, ["natural", "gradient", "noise"][i % 3],
                             [(40 + 37 * i % 400, 30 + 11 * i % 150, 120 + 9 * (i % 8))] if i % 4 != 2 else []) for i in range(N)]
    props = dict(width_to_process=W, multi_scale_factor=10, min_neighbors=2)
    streams = [capi.FaceStream(ctx, casc, **props) for _ in range(N)]
    keep, fr = _device_frames(frames)          # device frames: one launch set for the whole batch (host frames go in chunks)
    ctx.enable_kernel_timing(1)
    with ctx.options(**env):
        res = ctx.face_batch_process(streams, fr)
    kt = ctx.kernel_timing()
    ctx.enable_kernel_timing(0)
    if env.get("band"):
        assert _launched(kt, "cascade_band") == 1 and _launched(kt, "cascade_tile") == 0, kt
    else:
        assert _launched(kt, "cascade_band") == 0 and _launched(kt, "cascade_tile") == 1, kt
    seen = 0
    for i in range(N):
        eb, eid = orc.FaceStream(orc_cascade, width_to_process=W, scale_factor_pct=10, min_neighbors=2).process(frames[i])
        assert np.array_equal(res[i][0], eb) and np.array_equal(res[i][1], eid), (env, i, res[i][0], eb)
        seen += len(eb)
    assert seen >= N // 2
    for st in streams:
        st.close()
