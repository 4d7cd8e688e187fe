"""Stress: several threads hammer one context with different entry points (synchronous batches, the submit / collect
loop, tracker, a part detector, detectMultiScale on changing ROI sizes) for a while; every thread's outputs must equal
what the same call sequence produced single-threaded beforehand.  Usage: stress_threads.py [seconds]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd"))
import numpy as np
from nubovca import capi, synth

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
ctx = capi.Context(0)
casc = ctx.load_cascade_xml(synth.synthetic_cascade_xml())
eye_r = ctx.load_cascade_xml(synth.synthetic_part_cascade_xml("righteye"))
eye_l = ctx.load_cascade_xml(synth.synthetic_part_cascade_xml("lefteye"))


def seq_face(seed):
    fr = [synth.make_bgr(640, 480, seed + i, "natural", [(80 + 9 * i, 60, 220)] if i % 4 else []) for i in range(6)]
    def run():
        s = [capi.FaceStream(ctx, casc, width_to_process=320) for _ in range(3)]
        out = []
        for f in fr:
            out.append(ctx.face_batch_process(s, [capi.make_frame(f)] * 3))
        return out
    return run


def seq_async(seed):
    fr = [[synth.make_bgr(480, 360, seed + 10 * i + k, "natural", [(60 + 5 * i, 40, 200)]) for k in range(4)] for i in range(6)]
    def run():
        s = [capi.FaceStream(ctx, casc, width_to_process=480, multi_scale_factor=15) for _ in range(4)]
        out = []
        pend = ctx.face_batch_submit(s, [capi.make_frame(f) for f in fr[0]])
        for i in range(1, len(fr)):
            nxt = None
            while nxt is None:
                try:
                    nxt = ctx.face_batch_submit(s, [capi.make_frame(f) for f in fr[i]])
                except capi.NvcaError:      # both tickets busy (another thread's): retry
                    time.sleep(0.0005)
            out.append(ctx.face_batch_collect(pend)); pend = nxt
        out.append(ctx.face_batch_collect(pend))
        return out
    return run


def seq_tracker(seed):
    fr = [np.concatenate([synth.make_bgr(320, 240, seed, "natural", [(40 + 12 * i, 50, 100)]), np.full((240, 320, 1), 255, np.uint8)], axis=2) for i in range(6)]
    def run():
        t = capi.Tracker(ctx)
        return [t.process(f, 100.0 + 33 * i) for i, f in enumerate(fr)]
    return run


def seq_parts(seed):
    fr = [synth.make_bgr(640, 480, seed + i, "natural", [(120 + 6 * i, 80, 240)]) for i in range(4)]
    def run():
        p = capi.PartStream(ctx, capi.PART_EYE, casc, eye_r, eye_l)
        return [p.process(f) for f in fr]
    return run


def seq_detect(seed):
    rng = np.random.RandomState(seed)
    imgs = [synth.make_gray(int(rng.randint(90, 260)), int(rng.randint(70, 200)), seed + i, "natural") for i in range(8)]
    def run():
        out = []
        for i, g in enumerate(imgs):
            out.append(ctx.detect_multiscale(casc, g, 1.1, 2, [0, capi.HAAR_SCALE_IMAGE, capi.HAAR_FIND_BIGGEST_OBJECT][i % 3], (3, 3)))
        return out
    return run


def seq_varbatch(seed):
    """batches of changing size and geometry: every buffer of the workspace grows at some point"""
    rng = np.random.RandomState(seed)
    sizes = [1, 3, 9, 2, 19, 5, 33, 4]
    geos = [(320, 240), (400, 300), (320, 240), (480, 360)]
    fr = [[synth.make_bgr(*geos[(k + j) % 4], seed + 50 * k + j, "natural", [(40 + 3 * j, 30, 120)] if j % 3 else []) for j in range(n)] for k, n in enumerate(sizes)]
    def run():
        out = []
        for k, n in enumerate(sizes):
            s = [capi.FaceStream(ctx, casc, width_to_process=geos[(k + j) % 4][0], multi_scale_factor=20, min_neighbors=2 + j % 2) for j in range(n)]
            out.append(ctx.face_batch_process(s, [capi.make_frame(f) for f in fr[k]]))
        return out
    return run


def same(a, b):
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    return np.array_equal(np.asarray(a), np.asarray(b))


jobs = [seq_face(1), seq_face(500), seq_async(900), seq_tracker(7), seq_parts(40), seq_detect(3), seq_detect(77), seq_varbatch(11)]
expect = [j() for j in jobs]
stop = time.time() + SECS
errors, rounds = [], [0] * len(jobs)


def worker(k):
    try:
        while time.time() < stop:
            got = jobs[k]()
            if not same(got, expect[k]):
                errors.append("job %d diverged in round %d" % (k, rounds[k])); return
            rounds[k] += 1
    except Exception as e:      # noqa
        errors.append("job %d: %r" % (k, e))


th = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
for t in th:
    t.start()
for t in th:
    t.join()
print("rounds per job", rounds, "errors", errors)
sys.exit(1 if errors else 0)
