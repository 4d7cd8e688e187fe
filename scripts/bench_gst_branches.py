"""Aggregate frame rate of B nubofacedetector elements in ONE process (one pipeline branch each, own streaming thread),
through the GStreamer shim, with and without the shim's frame combiner.  Start-up (gst_init, plugin load, context and
plan creation) is removed by differencing two run lengths.  Usage: python scripts/bench_gst_branches.py [B] [W H] [frames per branch]"""
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
    frame = synth.make_bgr(W, H, 500, "natural", [(40, H // 6, H // 2)])
    raw = os.path.join(td, "frame.raw")
    with open(raw, "wb") as f:
        f.write(frame.tobytes())

    def run(nframes, extra):
        env = build_gst.env(); env["NVCA_CASCADE_DIR"] = td; env["NVCA_GST_STATS"] = "1"; env["NVCA_HARNESS_LOOP"] = str(nframes); env.update(extra)
        t0 = time.time()
        r = subprocess.run([build_gst.HARNESS, "nubofacedetector", "BGR", str(W), str(H), ",".join([raw] * B), "process-x-every-4-frames=4"] + os.environ.get("BENCH_GST_PROPS", "").split(),
                           env=env, capture_output=True, text=True, timeout=600)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr[-1000:]
        stat = [l for l in r.stderr.splitlines() if "largest combined" in l]
        return dt, (stat[-1].split("batch")[1].split()[0] if stat else "?")

    # every branch pushes the same one-frame file n times (multifilesrc loop); start-up is removed by differencing
    n1 = 50
    n2 = n1 + (int(sys.argv[4]) if len(sys.argv) > 4 else 1000)
    run(n1, {})                                               # page the libraries in
    for label, extra in (("combined", {}), ("per-frame", {"NVCA_GST_NO_COMBINE": "1"})):
        t1, _ = run(n1, extra)
        t2, mb = run(n2, extra)
        print("%-9s %d branches %dx%d: %.0f frames/s aggregate over %.1f s (largest round %s)" % (label, B, W, H, (n2 - n1) * B / (t2 - t1), t2 - t1, mb), flush=True)
