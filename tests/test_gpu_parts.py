"""GPU parity of the part detectors (BASELINE config 3: face -> eye / nose / mouth / ear
ROI chain): NuboEyeDetector, NuboNoseDetector, NuboMouthDetector, NuboEarDetector streams against the CPU oracle,
bit-exact box lists over multi-frame sequences (the merging heuristics carry state from frame to frame)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PARTS = ("righteye", "lefteye", "nose", "mouth", "leftear", "rightear")


@pytest.fixture(scope="module")
def env(synth_xml):
    import orc
    from nubovca import capi, synth
    ctx = capi.Context(0)
    xml = {n: synth.synthetic_part_cascade_xml(n) for n in PARTS}
    dev = {n: ctx.load_cascade_xml(x) for n, x in xml.items()}
    cpu = {n: orc.parse_cascade_xml(x) for n, x in xml.items()}
    dev["face"] = ctx.load_cascade_xml(synth_xml)
    cpu["face"] = orc.parse_cascade_xml(synth_xml)
    yield ctx, dev, cpu
    ctx.close()


KINDS = {"eye": (0, "righteye", "lefteye"), "nose": (1, "nose", None), "mouth": (2, "mouth", None), "ear": (3, "leftear", "rightear")}


def _streams(env, kind, **props):
    import orc
    from nubovca import capi
    ctx, dev, cpu = env
    k, a, b = KINDS[kind]
    names = {"width_to_process": "width_to_process", "process_x_every_4_frames": "process_x_every_4",
             "multi_scale_factor": "scale_factor_pct", "detect_event": "detect_event"}
    g = capi.PartStream(ctx, k, dev["face"], dev[a], dev[b] if b else None, **props)
    o = orc.PartStream(k, cpu["face"], cpu[a], cpu[b] if b else None, **{names[n]: v for n, v in props.items()})
    return g, o


def _scene(W, H, n, seed, two_faces=False):
    from nubovca import synth
    frames = []
    s = int(H * 0.5)
    for i in range(n):
        faces = [] if i % 6 == 4 else [(W // 5 + 5 * i, H // 5, s)]
        if two_faces and faces:
            faces.append((W // 2 + 30, H // 3 + 3 * i, int(s * 0.7)))
        frames.append(synth.make_bgr(W, H, seed + i, "natural", faces))
    return frames


@pytest.mark.parametrize("kind", ["eye", "nose", "mouth", "ear"])
@pytest.mark.parametrize("W,H,props,two", [
    (640, 480, {}, False),
    (1280, 720, {}, True),
    (800, 600, {}, False),                                   # width / 320 = 2.5: the int-scale truncation quirk
    (640, 480, {"process_x_every_4_frames": 2, "multi_scale_factor": 15}, False),
    (640, 480, {"width_to_process": 640}, False),
])
def test_part_stream_sequence(env, kind, W, H, props, two):
    g, o = _streams(env, kind, **props)
    seen = 0
    for i, f in enumerate(_scene(W, H, 9, 700 + W, two)):
        ga, gb = g.process(f)
        ea, eb = o.process(f)
        assert np.array_equal(ga, ea) and np.array_equal(gb, eb), (kind, i, ga, ea, gb, eb)
        seen += len(ea) + len(eb)
    assert seen > 0
    g.close()


@pytest.mark.parametrize("kind", ["eye", "nose", "mouth"])
def test_part_stream_detect_event_mode(env, kind):
    """faces arrive from an upstream nubofacedetector (original-frame pixels) instead of the own face pass"""
    import orc
    from nubovca import capi
    ctx, dev, cpu = env
    g, o = _streams(env, kind, detect_event=1)
    fs = capi.FaceStream(ctx, dev["face"])
    seen = 0
    for i, f in enumerate(_scene(640, 480, 8, 900)):
        boxes, _ = fs.process(f)
        if i % 3 != 2 and len(boxes):        # some frames come without an upstream message
            g.push_faces(boxes)
            o.push_faces(boxes)
        ga, gb = g.process(f)
        ea, eb = o.process(f)
        assert np.array_equal(ga, ea) and np.array_equal(gb, eb), (kind, i)
        seen += len(ea) + len(eb)
    assert seen > 0


def test_roi_chain_1080p(env):
    """BASELINE config 3: 1080p frame through the eye / nose / mouth / ear chain"""
    from nubovca import synth
    frames = [synth.make_bgr(1920, 1080, 40 + i, "natural", [(500 + 10 * i, 200, 600)]) for i in range(3)]
    for kind in ("eye", "nose", "mouth", "ear"):
        g, o = _streams(env, kind)
        tot = 0
        for f in frames:
            ga, gb = g.process(f)
            ea, eb = o.process(f)
            assert np.array_equal(ga, ea) and np.array_equal(gb, eb), kind
            tot += len(ea) + len(eb)
        assert tot > 0, kind


def test_roi_chain_1080p_batched(env):
    """the same chain through nvca_part_batch_process: four 1080p video streams x the four part detectors = 16 part
    streams in ONE call per tick (own face pass each), device-resident frames; every stream's lists as if it ran alone"""
    import torch
    from nubovca import capi, synth
    ctx = env[0]
    V, T = 4, 3
    kinds = ("eye", "nose", "mouth", "ear")
    pairs = [[_streams(env, k) for k in kinds] for _ in range(V)]
    tot = 0
    for t in range(T):
        frames = [synth.make_bgr(1920, 1080, 60 + 10 * v + t, "natural", [(300 + 100 * v + 10 * t, 150 + 20 * v, 500 + 30 * v)] if (v + t) % 5 else [])
                  for v in range(V)]
        keep = [torch.from_numpy(f).cuda() for f in frames]
        torch.cuda.synchronize()
        fr = [capi.make_frame(k.data_ptr(), 1920, 1080, 1920 * 3, capi.MEM_DEVICE) for k in keep]
        res = capi.part_batch_process(ctx, [pairs[v][j][0] for v in range(V) for j in range(4)], [fr[v] for v in range(V) for j in range(4)])
        for v in range(V):
            for j in range(4):
                ea, eb = pairs[v][j][1].process(frames[v])
                ga, gb = res[v * 4 + j]
                assert np.array_equal(ga, ea) and np.array_equal(gb, eb), (t, v, kinds[j], ga, ea, gb, eb)
                tot += len(ea) + len(eb)
    assert tot > 0


def test_roi_chain_submit_collect_two_ticks_in_flight(env):
    """nvca_part_batch_submit / _collect as a serving loop: tick k + 1's gates, working images and face passes are queued before tick k is
    collected (two calls in flight on the two buffer / lane sets), host and device frames, detect-event streams next to streams with
    their own face pass, ticks on which nothing runs (process-x-every-4-frames).  Every stream's lists as if it had run alone."""
    import torch
    from nubovca import capi, synth
    ctx = env[0]
    V, T = 3, 7
    kinds = ("eye", "nose", "mouth", "ear")
    pairs = [[_streams(env, k, process_x_every_4_frames=(2 if v == 1 else 4)) for k in kinds] for v in range(V)]
    scenes = [[synth.make_bgr(1920, 1080, 300 + 10 * v + t, "natural", [(300 + 100 * v + 12 * t, 150 + 20 * v, 500 + 30 * v)] if (v + t) % 4 else [])
               for v in range(V)] for t in range(T)]
    streams = [pairs[v][j][0] for v in range(V) for j in range(4)]
    keep = {}

    def frames_of(t, tag=0):
        out = []
        for v in range(V):
            if v == 0:                                     # host memory
                out += [scenes[t][v]] * 4
            else:
                d = torch.from_numpy(scenes[t][v]).cuda()
                keep[(t, v, tag)] = d                      # (alive until the test ends: a submitted call reads its frames until it is collected)
                out += [capi.make_frame(d.data_ptr(), 1920, 1080, 1920 * 3, capi.MEM_DEVICE)] * 4
        torch.cuda.synchronize()
        return out
    tot = 0
    tk = capi.part_batch_submit(ctx, streams, frames_of(0))
    for t in range(T):
        nxt = capi.part_batch_submit(ctx, streams, frames_of(t + 1)) if t + 1 < T else None
        if nxt is not None and t == 2:                     # a third call while two are in flight is refused, and changes nothing
            with pytest.raises(capi.NvcaError):
                capi.part_batch_submit(ctx, streams, frames_of(t + 1, tag=1))
            with pytest.raises(capi.NvcaError):            # ... and so is collecting out of order
                capi.part_batch_collect(ctx, nxt)
            with pytest.raises(capi.NvcaError):            # ... and a synchronous call on a stream that has a ticket outstanding
                capi.part_batch_process(ctx, streams[:1], [scenes[0][0]])
        res = capi.part_batch_collect(ctx, tk)
        for v in range(V):
            for j in range(4):
                ea, eb = pairs[v][j][1].process(scenes[t][v])
                ga, gb = res[v * 4 + j]
                assert np.array_equal(ga, ea) and np.array_equal(gb, eb), (t, v, kinds[j], ga, ea, gb, eb)
                tot += len(ea) + len(eb)
        tk = nxt
    assert tot > 0
    # the synchronous call still works next to it, and an abandoned ticket is rolled back with its context (nothing to assert but no crash)
    res = capi.part_batch_process(ctx, streams[:4], [scenes[0][0]] * 4)
    for j in range(4):
        ea, eb = pairs[0][j][1].process(scenes[0][0])
        assert np.array_equal(res[j][0], ea) and np.array_equal(res[j][1], eb)
    # a stream of an outstanding call is destroyed: the call is abandoned -- rolled back, drained -- and its ticket unknown; the other
    # streams are as they were before the submit (their oracle twins never saw the abandoned frame)
    tk = capi.part_batch_submit(ctx, streams[:4], [scenes[1][0]] * 4)
    pairs[0][0][0].close()
    with pytest.raises(capi.NvcaError):
        capi.part_batch_collect(ctx, tk)
    res = capi.part_batch_process(ctx, streams[1:4], [scenes[2][0]] * 3)
    for j in range(1, 4):
        ea, eb = pairs[0][j][1].process(scenes[2][0])
        assert np.array_equal(res[j - 1][0], ea) and np.array_equal(res[j - 1][1], eb)


@pytest.mark.parametrize("opts", [{"host_threads": 0}, {"roi": 0}, {"roi": 0, "host_threads": 0}])
def test_roi_chain_batched_on_the_other_paths(env, opts):
    """the batched chain with the per-job host work on the calling thread only (no helper threads), and with the face-region
    searches on the large-image path (plan + pre-pass + tiles per region) instead of the one-launch small-image detector:
    the same lists either way"""
    import torch
    from nubovca import capi, synth
    ctx = env[0]
    V, T = 3, 3
    kinds = ("eye", "nose", "mouth", "ear")
    pairs = [[_streams(env, k) for k in kinds] for _ in range(V)]
    tot = 0
    with ctx.options(**opts):
        for t in range(T):
            frames = [synth.make_bgr(1920, 1080, 160 + 10 * v + t, "natural", [(200 + 150 * v + 10 * t, 150 + 20 * v, 520 + 30 * v), (1150, 300 + 10 * t, 480)] if (v + t) % 4 else [])
                      for v in range(V)]
            keep = [torch.from_numpy(f).cuda() for f in frames]
            torch.cuda.synchronize()
            fr = [capi.make_frame(k.data_ptr(), 1920, 1080, 1920 * 3, capi.MEM_DEVICE) for k in keep]
            res = capi.part_batch_process(ctx, [pairs[v][j][0] for v in range(V) for j in range(4)], [fr[v] for v in range(V) for j in range(4)])
            for v in range(V):
                for j in range(4):
                    ea, eb = pairs[v][j][1].process(frames[v])
                    ga, gb = res[v * 4 + j]
                    assert np.array_equal(ga, ea) and np.array_equal(gb, eb), (opts, t, v, kinds[j], ga, ea, gb, eb)
                    tot += len(ea) + len(eb)
    assert tot > 20


def test_part_batch_mixed_sizes_and_event_mode(env):
    """one batched call with streams of different kinds, frame sizes and modes (own face pass / faces pushed by an upstream
    face detector / gated by process-x-every-4), host frames; five ticks so the merging state takes part"""
    from nubovca import capi
    ctx, dev, cpu = env
    specs = [("eye", 640, 480, {}), ("nose", 800, 600, {}), ("mouth", 640, 480, {"detect_event": 1}), ("ear", 1280, 720, {}),
             ("eye", 640, 480, {"detect_event": 1}), ("nose", 640, 480, {"process_x_every_4_frames": 2}), ("mouth", 800, 600, {"width_to_process": 400})]
    pairs = [_streams(env, k, **p) for k, _, _, p in specs]
    scenes = [_scene(W, H, 5, 1200 + 31 * i, two_faces=(i % 2 == 0)) for i, (_, W, H, _) in enumerate(specs)]
    fs = [capi.FaceStream(ctx, dev["face"]) for _ in specs]
    tot = 0
    for t in range(5):
        for i, (k, W, H, p) in enumerate(specs):
            if p.get("detect_event"):
                boxes, _ = fs[i].process(scenes[i][t])
                if t % 3 != 2 and len(boxes):
                    pairs[i][0].push_faces(boxes); pairs[i][1].push_faces(boxes)
        res = capi.part_batch_process(ctx, [g for g, _ in pairs], [scenes[i][t] for i in range(len(specs))])
        for i in range(len(specs)):
            ea, eb = pairs[i][1].process(scenes[i][t])
            assert np.array_equal(res[i][0], ea) and np.array_equal(res[i][1], eb), (t, i, specs[i][0])
            tot += len(ea) + len(eb)
    assert tot > 0
    with pytest.raises(capi.NvcaError):          # a stream twice in one call
        capi.part_batch_process(ctx, [pairs[0][0], pairs[0][0]], [scenes[0][0], scenes[0][0]])


def test_part_batch_streams_on_one_frame_share_their_work(env):
    """several part detectors handed the SAME frame in one call (the elements of one video stream): the upload, the gray
    image, the working images and the face pass are computed once per frame and shared where the requests are identical --
    and only there: different working widths, scale factors, the eye detector's equalized chain and event-mode streams all
    sit in the same group.  Host frames (one upload per frame) and device frames; lists as if every stream ran alone."""
    import torch
    from nubovca import capi
    ctx, dev, cpu = env
    specs = [("nose", {}), ("mouth", {}), ("ear", {}), ("eye", {}), ("nose", {"width_to_process": 160}), ("mouth", {"multi_scale_factor": 15}),
             ("nose", {}), ("eye", {"detect_event": 1}), ("mouth", {"detect_event": 1}), ("ear", {"width_to_process": 160})]
    for mem in ("host", "device"):
        V = 2
        pairs = [[_streams(env, k, **p) for k, p in specs] for _ in range(V)]
        scenes = [_scene(800, 600, 4, 4100 + 57 * v, two_faces=(v == 0)) for v in range(V)]
        fs = [capi.FaceStream(ctx, dev["face"]) for _ in range(V)]
        tot = 0
        for t in range(4):
            frames = [scenes[v][t] for v in range(V)]
            for v in range(V):
                boxes, _ = fs[v].process(frames[v])
                for (k, p), (g, o) in zip(specs, pairs[v]):
                    if p.get("detect_event") and len(boxes) and t != 2:
                        g.push_faces(boxes); o.push_faces(boxes)
            if mem == "device":
                keep = [torch.from_numpy(f).cuda() for f in frames]
                torch.cuda.synchronize()
                handed = [capi.make_frame(k.data_ptr(), 800, 600, 800 * 3, capi.MEM_DEVICE) for k in keep]
            else:
                handed = frames
            order = [(v, j) for j in range(len(specs)) for v in range(V)]          # the streams of a frame are not adjacent in the call
            res = capi.part_batch_process(ctx, [pairs[v][j][0] for v, j in order], [handed[v] for v, j in order])
            for (v, j), (ga, gb) in zip(order, res):
                ea, eb = pairs[v][j][1].process(frames[v])
                assert np.array_equal(ga, ea) and np.array_equal(gb, eb), (mem, t, v, specs[j], ga, ea, gb, eb)
                tot += len(ea) + len(eb)
        assert tot > 0


def test_part_batch_more_images_than_one_job_carries(env):
    """40 nose + 20 ear + 6 eye streams on distinct frames of one size in ONE call: the face passes are split over several
    N-image jobs (32 images per job, 16 image + mirror pairs for the ear detector), the working images come from single
    launch sets of 40+ frames; two ticks so that the ring of launch tables and the arena are reused"""
    from nubovca import capi
    ctx, dev, cpu = env
    kinds = ["nose"] * 40 + ["ear"] * 20 + ["eye"] * 6
    pairs = [_streams(env, k) for k in kinds]
    W, H = 320, 240
    tot = 0
    for t in range(2):
        frames = []
        for i in range(len(kinds)):
            from nubovca import synth
            faces = [] if i % 7 == 3 else [(40 + (5 * i + 3 * t) % 60, 30 + (3 * i) % 25, 130 + (i % 4) * 10)]
            frames.append(synth.make_bgr(W, H, 8800 + 11 * i + t, "natural", faces))
        res = capi.part_batch_process(ctx, [g for g, _ in pairs], frames)
        for i, (ga, gb) in enumerate(res):
            ea, eb = pairs[i][1].process(frames[i])
            assert np.array_equal(ga, ea) and np.array_equal(gb, eb), (t, i, kinds[i], ga, ea, gb, eb)
            tot += len(ea) + len(eb)
    assert tot > 0
    assert capi.part_batch_process(ctx, [], []) == []


def test_part_batch_late_failure_leaves_every_stream_untouched(env):
    """A batched call that fails AFTER the gates have advanced (one eye stream whose face pass cannot be planned: multi-scale-factor 0
    is scaleFactor 1.0, where OpenCV's assertion fires) returns an error and leaves every stream of the call as it found it: frame gates,
    queued face events (detect-event streams) and result lists.  The streams then run on -- one by one, as the GStreamer shim
    does after a refused batch -- exactly like streams that never saw the failed call."""
    from nubovca import capi
    ctx, dev, cpu = env
    frames = _scene(640, 480, 6, 5150)
    fs = capi.FaceStream(ctx, dev["face"])
    good = [_streams(env, "nose", process_x_every_4_frames=2), _streams(env, "mouth", detect_event=1), _streams(env, "ear")]
    bad = capi.PartStream(ctx, 0, dev["face"], dev["righteye"], dev["lefteye"], multi_scale_factor=0)
    tot = 0
    for t, f in enumerate(frames):
        boxes, _ = fs.process(f)
        if len(boxes) and t != 3:
            good[1][0].push_faces(boxes); good[1][1].push_faces(boxes)
        if t in (1, 2, 4):          # the refused batch: the bad stream last, so every good gate has advanced before it fails
            with pytest.raises(capi.NvcaError) as ei:
                capi.part_batch_process(ctx, [g for g, _ in good] + [bad], [f] * 4)
            assert ei.value.code in (capi.ERR_ARG, capi.ERR_UNSUPPORTED), ei.value
        for g, o in good:
            ga, gb = g.process(f)
            ea, eb = o.process(f)
            assert np.array_equal(ga, ea) and np.array_equal(gb, eb), (t, ga, ea, gb, eb)
            tot += len(ea) + len(eb)
    assert tot > 0


def test_part_batch_without_any_job_returns_drained(env):
    """detect-event streams whose upstream sent an EMPTY face list: the call queues the frame upload and the working images,
    but no face pass and no part search -- it must still return with nothing in flight (the caller recycles its buffer; the
    next single-stream call carves the same arena on another lane).  Checked by overwriting the host frame right after the
    call and by the following calls' results."""
    from nubovca import capi
    ctx, dev, cpu = env
    pairs = [_streams(env, "nose", detect_event=1), _streams(env, "mouth", detect_event=1)]
    frames = _scene(640, 480, 4, 6200)
    fs = capi.FaceStream(ctx, dev["face"])
    tot = 0
    for t, f in enumerate(frames):
        boxes, _ = fs.process(f)
        handed = f.copy()
        if t % 2 == 0:
            for g, o in pairs:
                g.push_faces(np.zeros((0, 4), np.int32)); o.push_faces(np.zeros((0, 4), np.int32))
        elif len(boxes):
            for g, o in pairs:
                g.push_faces(boxes); o.push_faces(boxes)
        res = capi.part_batch_process(ctx, [g for g, _ in pairs], [handed, handed])
        handed[:] = 0                                        # the buffer goes back to its pool
        for (g, o), (ga, gb) in zip(pairs, res):
            ea, eb = o.process(f)
            assert np.array_equal(ga, ea) and np.array_equal(gb, eb), (t, ga, ea)
            tot += len(ea) + len(eb)
        one = _streams(env, "nose")                          # a single-stream call right behind it (lane 0, same arena)
        ga, gb = one[0].process(f)
        ea, eb = one[1].process(f)
        assert np.array_equal(ga, ea) and np.array_equal(gb, eb)
        one[0].close()
    assert tot >= 0


def test_flip_primitive(env):
    import ctypes as C
    import orc
    from nubovca import capi
    ctx = env[0]
    img = np.random.default_rng(3).integers(0, 256, size=(37, 101), dtype=np.uint8)
    out = np.empty_like(img)
    ctx.check(ctx.L.nvca_flip_horizontal(ctx.h, img.ctypes.data, 101, 37, 101, capi.MEM_HOST, out.ctypes.data, 101))
    assert np.array_equal(out, orc.flip_h(img))
