"""Robustness run: B branches, each the full chain nubofacedetector ! nuboeyedetector ! nubonosedetector ! nubomouthdetector !
nuboeardetector ! nubotracker-free (BGR), in ONE process with two virtual GPU slots: every combiner of the shim is busy at once.
Checks that the process ends cleanly and every branch emitted one event per frame from the observed element."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd"))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd", "gst"))
import build_gst
from nubovca import synth
B, N, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 6, 24, 640, 480
build_gst.build(required=True)
names = {"righteye": "haarcascade_mcs_righteye.xml", "lefteye": "haarcascade_mcs_lefteye.xml", "nose": "haarcascade_mcs_nose.xml",
         "mouth": "haarcascade_mcs_mouth.xml", "leftear": "haarcascade_mcs_leftear.xml", "rightear": "haarcascade_mcs_rightear.xml"}
with tempfile.TemporaryDirectory() as td:
    open(os.path.join(td, "haarcascade_frontalface_alt.xml"), "w").write(synth.synthetic_cascade_xml())
    open(os.path.join(td, "haarcascade_profileface.xml"), "w").write(synth.synthetic_cascade_xml())
    for n, fn in names.items():
        open(os.path.join(td, fn), "w").write(synth.synthetic_part_cascade_xml(n))
    raws = []
    for b in range(B):
        p = os.path.join(td, "f%d.raw" % b)
        with open(p, "wb") as f:
            for i in range(N):
                f.write(synth.make_bgr(W, H, 70 * b + i, "natural", [] if (i + b) % 5 == 3 else [(100 + 10 * b + 4 * i, 80, 230)]).tobytes())
        raws.append(p)
    env = build_gst.env(); env["NVCA_CASCADE_DIR"] = td; env["NVCA_GST_STATS"] = "1"; env["NVCA_VIRTUAL_GPUS"] = "2"
    chain = "nubofacedetector view-faces=1 ! nuboeyedetector view-eyes=1 ! nubonosedetector view-noses=1 ! nubomouthdetector name=el view-mouths=1 ! nuboeardetector view-ears=1"
    r = subprocess.run([build_gst.HARNESS, chain, "BGR", str(W), str(H), ",".join(raws)], env=env, capture_output=True, text=True, timeout=600)
    ev = [l for l in r.stdout.splitlines() if l.startswith("event")]
    print("rc", r.returncode, "events", len(ev), "of", B * N, "|", "; ".join(l for l in r.stderr.splitlines() if "largest" in l)[:400])
    assert r.returncode == 0 and "done 0" in r.stdout, r.stderr[-1500:]
