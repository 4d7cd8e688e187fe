"""Several streaming threads (= several GStreamer elements) share one context: entry points serialise on the
context, results stay identical to single-threaded runs."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_streams_share_context(synth_xml, orc_cascade):
    import orc
    from nubovca import capi, synth
    ctx = capi.Context(0)
    casc = ctx.load_cascade_xml(synth_xml)
    W, H, N = 640, 480, 6
    seqs = [[synth.make_bgr(W, H, 10 * t + i, "natural", [(100 + 8 * i + 20 * t, 80, 200)]) for i in range(N)] for t in range(4)]
    bgra = [[np.concatenate([f, np.full((H, W, 1), 255, np.uint8)], axis=2) for f in s] for s in seqs]
    out, errs = {}, []

    def face_worker(t):
        try:
            fs = capi.FaceStream(ctx, casc)
            out[("f", t)] = [fs.process(f) for f in seqs[t]]
        except Exception as e:      # noqa
            errs.append(e)

    def trk_worker(t):
        try:
            tr = capi.Tracker(ctx)
            out[("t", t)] = [tr.process(f, 100.0 + 33 * i) for i, f in enumerate(bgra[t])]
        except Exception as e:      # noqa
            errs.append(e)

    th = [threading.Thread(target=face_worker, args=(t,)) for t in range(4)] + [threading.Thread(target=trk_worker, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for t in range(4):
        ofs = orc.FaceStream(orc_cascade)
        for i, f in enumerate(seqs[t]):
            eb, eid = ofs.process(f)
            assert np.array_equal(out[("f", t)][i][0], eb) and np.array_equal(out[("f", t)][i][1], eid)
    for t in range(2):
        otr = orc.Tracker()
        for i, f in enumerate(bgra[t]):
            assert np.array_equal(out[("t", t)][i], otr.process(f, 100.0 + 33 * i, cap=1 << 16))
    ctx.close()
