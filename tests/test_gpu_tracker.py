"""GPU parity of the NuboTracker path (TRK/gstnubotracker.cpp:339-421) against the CPU oracle:
bit-exact boxes, in order, on seeded moving-rectangle sequences (SURVEY.md 8d content iii)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from nubovca import capi
    c = capi.Context(0)
    yield c
    c.close()


def moving_scene(W, H, n_frames, n_rects, seed, noise=0):
    """static textured background + rectangles moving 8 px/frame; BGRA"""
    rng = np.random.default_rng(seed)
    bg = rng.integers(60, 120, size=(H, W), dtype=np.uint8)
    rects = [(int(rng.integers(0, W - 80)), int(rng.integers(0, H - 60)), int(rng.integers(12, 70)), int(rng.integers(12, 50)),
              int(rng.choice([-8, 8])), int(rng.choice([-8, 0, 8]))) for _ in range(n_rects)]
    frames = []
    for f in range(n_frames):
        g = bg.copy()
        if noise:
            g = np.clip(g.astype(np.int32) + rng.integers(-noise, noise + 1, size=g.shape), 0, 255).astype(np.uint8)
        for (x, y, w, h, dx, dy) in rects:
            xx, yy = (x + dx * f) % (W - w), (y + dy * f) % (H - h)
            g[yy:yy + h, xx:xx + w] = 230
        img = np.empty((H, W, 4), np.uint8)
        img[..., 0] = g
        img[..., 1] = g
        img[..., 2] = g
        img[..., 3] = 255
        frames.append(img)
    return frames


@pytest.mark.parametrize("W,H,n_rects,noise,props", [
    (160, 120, 2, 0, {}),
    (640, 480, 4, 0, {}),
    (641, 479, 6, 0, {}),                                   # odd geometry: scalar pixel kernel
    (1280, 720, 32, 0, {}),
    (1920, 1080, 32, 0, {}),
    (640, 480, 4, 30, {}),                                  # sensor noise: thousands of tiny components
    (640, 480, 8, 0, {"set_threshold": 5, "set_min_area": 10, "set_distance": 100}),
    (320, 240, 3, 0, {"set_max_area": 600}),
])
def test_tracker_sequence(ctx, W, H, n_rects, noise, props):
    import orc
    from nubovca import capi
    names = {"set_threshold": "threshold", "set_min_area": "min_area", "set_max_area": "max_area", "set_distance": "distance"}
    trk = capi.Tracker(ctx, **props)
    otr = orc.Tracker(**{names[k]: v for k, v in props.items()})
    seen = 0
    for i, f in enumerate(moving_scene(W, H, 6, n_rects, 11 + W, noise)):
        ts = 1000.0 + 33.3 * i
        got = trk.process(f, ts)
        exp = otr.process(f, ts, cap=1 << 16)
        assert np.array_equal(got, exp), (i, got[:5], exp[:5])
        seen += len(exp)
    assert seen > 0
    trk.close()


def test_tracker_mhi_persistence(ctx):
    """frames closer together than MHI_DURATION: stale-but-recent pixels stay in the MHI and join
    components through the floating-range rule."""
    import orc
    from nubovca import capi
    trk = capi.Tracker(ctx, mhi_duration=100.0)
    otr = orc.Tracker(mhi_duration=100.0)
    for i, f in enumerate(moving_scene(320, 240, 8, 3, 5)):
        ts = 500.0 + 20.0 * i
        assert np.array_equal(trk.process(f, ts), otr.process(f, ts, cap=1 << 16))
    trk.close()


def test_tracker_batch_and_device_frames(ctx):
    import orc
    import torch
    from nubovca import capi
    W, H = 640, 480
    seqs = [moving_scene(W, H, 5, 4, 100 + s) for s in range(4)]
    trks = [capi.Tracker(ctx) for _ in range(4)]
    otrs = [orc.Tracker() for _ in range(4)]
    for i in range(5):
        dev = [torch.from_numpy(seqs[s][i]).cuda() for s in range(4)]
        torch.cuda.synchronize()
        frames = [capi.make_frame(d.data_ptr(), W, H, W * 4, capi.MEM_DEVICE) for d in dev]
        ts = [2000.0 + 33.3 * i] * 4
        res = capi.tracker_batch_process(ctx, trks, frames, ts)
        for s in range(4):
            assert np.array_equal(res[s], otrs[s].process(seqs[s][i], ts[s], cap=1 << 16))


def test_tracker_scene_that_changes_everywhere(ctx):
    """every pixel moves between frames (a cut, a camera pan): the motion mask is salt-and-pepper plus components as large
    as the frame -- thousands of pixels report to the same accumulators (k_ccl_reduce aggregates per wave and only issues
    atomics that can still improve a value).  Same boxes as the oracle, identical on a second tracker, and the batched call
    must not take milliseconds per frame"""
    import time
    import orc
    import torch
    from nubovca import capi, synth
    W, H, S = 1280, 720, 4
    frames = [np.concatenate([synth.make_bgr(W, H, 4300 + i, "natural" if i % 3 else "noise"), np.full((H, W, 1), 255, np.uint8)], axis=2) for i in range(6)]
    trk, twin, otr = capi.Tracker(ctx), capi.Tracker(ctx), orc.Tracker()
    seen = 0
    for i, f in enumerate(frames):
        got = trk.process(f, 900.0 + 33.3 * i)
        assert np.array_equal(got, otr.process(f, 900.0 + 33.3 * i, cap=1 << 16)), i
        assert np.array_equal(got, twin.process(f, 900.0 + 33.3 * i)), i
        seen += len(got)
    assert seen > 50
    dev = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    fr = [capi.make_frame(d.data_ptr(), W, H, W * 4, capi.MEM_DEVICE) for d in dev]
    trks = [capi.Tracker(ctx) for _ in range(S)]
    for i in range(3):
        capi.tracker_batch_process(ctx, trks, [fr[(i + s) % 6] for s in range(S)], [33.3 * i] * S)
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(3, 13):
        capi.tracker_batch_process(ctx, trks, [fr[(i + s) % 6] for s in range(S)], [33.3 * i] * S)
    ctx.synchronize()
    per_frame_ms = (time.perf_counter() - t0) / (10 * S) * 1e3
    assert per_frame_ms < 1.5, per_frame_ms          # 3.2 ms before the aggregation, 0.2 ms after (MI355X)


def test_tracker_resolution_change_resets_state(ctx):
    import orc
    from nubovca import capi
    trk, otr = capi.Tracker(ctx), orc.Tracker()
    a = moving_scene(320, 240, 3, 2, 1)
    for i, f in enumerate(a):
        assert np.array_equal(trk.process(f, 100.0 + 33 * i), otr.process(f, 100.0 + 33 * i))


@pytest.mark.parametrize("fold", [1, 0])
def test_tracker_component_paths_agree_and_root_list_overflow_falls_back(ctx, fold):
    """the per-tile reduction + fold of tile roots (default) and the per-pixel component kernels ("trk_fold" 0) give the oracle's boxes on
    the same frames; a frame of isolated moving pixels (a quarter of all pixels a component of its own: more tile roots than the list
    holds) takes the fallback inside the call and still answers as the oracle does"""
    import orc
    from nubovca import capi
    W, H = 640, 480
    with ctx.options(trk_fold=fold):
        trk, otr = capi.Tracker(ctx), orc.Tracker()
        seq = moving_scene(W, H, 4, 6, 77, noise=20)
        dots = np.zeros((H, W, 4), np.uint8); dots[..., 3] = 255
        lit = dots.copy(); lit[::2, ::2, :3] = 255                      # every other pixel of every other row jumps: 76 800 one-pixel components
        seq = seq[:2] + [dots, lit, dots] + seq[2:]
        seen = 0
        for i, f in enumerate(seq):
            ts = 2000.0 + 33.3 * i
            got = trk.process(f, ts)
            exp = otr.process(f, ts, cap=1 << 16)
            assert np.array_equal(np.asarray(got).reshape(-1, 4), np.asarray(exp).reshape(-1, 4)), (fold, i, len(got), len(exp))
            seen += len(exp)
        assert seen > 0
        trk.close()


def test_tracker_batches_of_changing_shape_share_one_workspace(ctx):
    """The live-tile list of the component kernels (marks stamped with the launch's tick, a count per slot) lives in the context's
    tracker workspace and is laid out per (frame size, batch): calls that alternate between batch sizes, and one call that mixes two
    frame sizes (two launch sets), must not read each other's marks -- every stream against its own oracle tracker, tick by tick."""
    import orc
    from nubovca import capi
    geos = [(640, 480), (640, 480), (640, 480), (352, 288), (352, 288)]
    n = 7
    seqs = [moving_scene(W, H, n, 5, 900 + s, noise=(3 if s % 2 else 0)) for s, (W, H) in enumerate(geos)]
    trks = [capi.Tracker(ctx) for _ in geos]
    otrs = [orc.Tracker() for _ in geos]
    # which streams take part in tick i: all five (two launch sets), then three of one size, then one alone, then the two small ones ...
    parts = [[0, 1, 2, 3, 4], [0, 1, 2], [3], [3, 4], [0, 1, 2, 3, 4], [1], [0, 1, 2, 3, 4]]
    seen = [0] * len(geos)
    found = 0
    for i in range(n):
        who = parts[i]
        frames, ts = [], []
        for s in who:
            W, H = geos[s]
            fr = seqs[s][seen[s]]
            frames.append(capi.make_frame(fr.ctypes.data, W, H, W * 4, capi.MEM_HOST))
            ts.append(5000.0 + 33.3 * seen[s])
        res = capi.tracker_batch_process(ctx, [trks[s] for s in who], frames, ts)
        for k, s in enumerate(who):
            exp = otrs[s].process(seqs[s][seen[s]], ts[k], cap=1 << 16)
            assert np.array_equal(res[k], exp), (i, s)
            found += len(exp)
            seen[s] += 1
    assert found > 0
    for t in trks:
        t.close()
