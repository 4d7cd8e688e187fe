"""Aggregate frame rate of B nubofacedetector elements in ONE process (one pipeline branch each, own streaming thread),
through the GStreamer shim, with and without the shim's frame combiner.  Start-up (gst_init, plugin load, context and
plan creation) is removed by differencing two run lengths.  Usage: python scripts/bench_gst_branches.py [B] [W H]"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd"))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd", "gst"))
import build_gst
from nubovca import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)
build_gst.build(required=True)
xml = synth.synthetic_cascade_xml()
with tempfile.TemporaryDirectory() as td:
    open(os.path.join(td, "haarcascade_frontalface_alt.xml"), "w").write(xml)
    base = [synth.make_bgr(W, H, 500 + i, "natural", [(40 + 5 * i, H // 6, H // 2)]) for i in range(10)]
    def run(nframes, extra):
        raw = os.path.join(td, "f%d.raw" % nframes)
        if not os.path.exists(raw):
            with open(raw, "wb") as f:
                for i in range(nframes):
                    f.write(base[i % len(base)].tobytes())
        env = build_gst.env(); env["NVCA_CASCADE_DIR"] = td; env["NVCA_GST_STATS"] = "1"; env.update(extra)
        t0 = time.time()
        r = subprocess.run([build_gst.HARNESS, "nubofacedetector", "BGR", str(W), str(H), ",".join([raw] * B), "process-x-every-4-frames=4"],
                           env=env, capture_output=True, text=True, timeout=600)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr[-1000:]
        stat = [l for l in r.stderr.splitlines() if "largest combined" in l]
        return dt, (stat[-1].split()[-1] if stat else "?")
    n1 = 40
    n2 = n1 + max(200, int(2.4e9 / (W * H * 3 * 8)))          # the long run adds a few seconds of work
    run(n1, {}); run(n2, {})                                  # page the files and the libraries in
    for label, extra in (("combined", {}), ("per-frame", {"NVCA_GST_NO_COMBINE": "1"})):
        t1 = min(run(n1, extra)[0] for _ in range(2))
        t2, mb = min(run(n2, extra) for _ in range(2))
        print("%-9s %d branches %dx%d: %.0f frames/s aggregate (largest round %s)" % (label, B, W, H, (n2 - n1) * B / (t2 - t1), mb), flush=True)
