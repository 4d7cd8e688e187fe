"""The GStreamer boundary: the shim registers the reference's factory names with their
caps / properties / signals, and -- on a GPU -- a pipeline `filesrc ! rawvideoparse ! nubofacedetector ! fakesink`
emits exactly the downstream "message" events the oracle's per-frame state machine predicts."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd", "gst"))
import build_gst  # noqa: E402

pytestmark = pytest.mark.skipif(not build_gst.available(), reason="GStreamer dev files not present")
GST_INSPECT = os.path.join(build_gst.CONDA, "bin", "gst-inspect-1.0")


@pytest.fixture(scope="module")
def shim():
    import __graft_entry__ as ge
    ge.build()
    return build_gst.build(required=True)


def _inspect(name):
    r = subprocess.run([GST_INSPECT, name], env=build_gst.env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_face_element_surface(shim):
    out = _inspect("nubofacedetector")
    for prop in ("view-faces", "detect-event", "send-meta-data", "width-to-process", "process-x-every-4-frames",
                 "euclidean-distance", "track-threshold", "area-threshold", "multi-scale-factor", "activate-events",
                 "events-ms", "image-to-overlay"):
        assert prop in out, prop
    assert '"face-event"' in out and "format: { (string)BGR }" in out.replace("  ", " ")
    assert "Range: 0 - 640 Default: 160" in out and "Range: 0 - 51 Default: 25" in out


def test_tracker_element_surface(shim):
    out = _inspect("nubotracker")
    for prop in ("set-threshold", "set-min-area", "set-max-area", "set-distance", "set-visual-mode", "activate-events", "events-ms"):  # GLib canonicalises _ to -
        assert prop in out, prop
    assert '"tracker-event"' in out and "BGRA" in out
    assert "Range: 0 - 255 Default: 20" in out and "Range: 0 - 300000 Default: 30000" in out


def _run_harness(element, fmt, W, H, frames, props=(), cascade_xml=None):
    with tempfile.TemporaryDirectory() as td:
        raw = os.path.join(td, "frames.raw")
        with open(raw, "wb") as f:
            for fr in frames:
                f.write(np.ascontiguousarray(fr).tobytes())
        env = build_gst.env()
        if cascade_xml is not None:
            with open(os.path.join(td, "haarcascade_frontalface_alt.xml"), "w") as f:
                f.write(cascade_xml)
            env["NVCA_CASCADE_DIR"] = td
        r = subprocess.run([build_gst.HARNESS, element, fmt, str(W), str(H), raw] + list(props), env=env,
                           capture_output=True, text=True, timeout=300)
        return r


def test_pipeline_without_gpu_passes_frames_through(shim):
    """reference behaviour: failures are logged, never propagated (FACE/kmsfacedetect.cpp:897)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    frames = [np.zeros((48, 64, 3), np.uint8)] * 3
    r = _run_harness("nubofacedetector", "BGR", 64, 48, frames)
    assert r.returncode == 0 and "done 0" in r.stdout, r.stderr[-1500:]


@pytest.mark.gpu
def test_face_pipeline_events_match_oracle(shim, synth_xml, orc_cascade):
    import orc
    from nubovca import synth
    W, H = 640, 480
    frames = []
    for i in range(8):
        faces = [] if i % 5 == 3 else [(40 + 6 * i, H // 6, H // 2)]
        frames.append(synth.make_bgr(W, H, 300 + i, "natural", faces))
    r = _run_harness("nubofacedetector", "BGR", W, H, frames, props=["activate-events=1", "events-ms=0"], cascade_xml=synth_xml)
    assert r.returncode == 0, r.stderr[-2000:]
    events = [l for l in r.stdout.splitlines() if l.startswith("event ")]
    assert len(events) == len(frames)
    ofs = orc.FaceStream(orc_cascade)
    n_boxes = 0
    for fr, line in zip(frames, events):
        boxes, _ = ofs.process(fr)
        exp = "".join("face:%d,%d,%d,%d;" % tuple(b) for b in boxes)
        got = line.split(" ", 2)[2] if len(line.split(" ", 2)) > 2 else ""
        assert got == exp, (line, exp)
        n_boxes += len(boxes)
    assert n_boxes > 0
    assert any(l.startswith("signal x:") for l in r.stdout.splitlines())


@pytest.mark.gpu
def test_tracker_pipeline_runs(shim):
    W, H = 320, 240
    frames = []
    for i in range(6):
        f = np.full((H, W, 4), 40, np.uint8)
        f[60:120, 30 + 10 * i:90 + 10 * i, :3] = 220
        frames.append(f)
    r = _run_harness("nubotracker", "BGRA", W, H, frames, props=["activate-events=1", "events-ms=0"])
    assert r.returncode == 0, r.stderr[-2000:]
    sig = [l for l in r.stdout.splitlines() if l.startswith("signal ")]
    assert len(sig) >= 3 and all("width:" in s for s in sig)
