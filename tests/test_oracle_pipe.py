"""Known-answer tests of the oracle's restatement of the reference's own glue:
Faces::track_faces (FACE/Faces.cpp:78-188), the per-frame gating of
kms_face_detect_process_frame (FACE/kmsfacedetect.cpp:794-830), and the
tracker (TRK/gstnubotracker.cpp:119-200,339-421; motempl semantics A.10-A.11)."""
import numpy as np
import orc
from nubovca import synth


# ----------------------------------------------------------- track_faces
def test_track_first_frame_assigns_ids():
    f, ids, nid = orc.track_faces([], [], 0, [[10, 10, 40, 40], [100, 50, 30, 30]])
    assert f.tolist() == [[10, 10, 40, 40], [100, 50, 30, 30]] and ids.tolist() == [0, 1] and nid == 2


def test_track_small_move_keeps_old_box():
    # area 1600 -> limit 3 ; centre moved by 2 px (<=3) and area equal -> old box kept
    f, ids, nid = orc.track_faces([[10, 10, 40, 40]], [5], 6, [[12, 10, 40, 40]])
    assert f.tolist() == [[10, 10, 40, 40]] and ids.tolist() == [5] and nid == 6


def test_track_big_move_takes_new_box_keeps_id():
    f, ids, nid = orc.track_faces([[10, 10, 40, 40]], [5], 6, [[20, 10, 40, 40]])
    assert f.tolist() == [[20, 10, 40, 40]] and ids.tolist() == [5]


def test_track_area_change_keeps_xy_takes_wh():
    # same centre (dist 0 <= limit), area 1600 vs 2500: |diff|*100/2500 = 36 > 15
    f, ids, _ = orc.track_faces([[10, 10, 40, 40]], [0], 1, [[5, 5, 50, 50]])
    assert f.tolist() == [[10, 10, 50, 50]] and ids.tolist() == [0]


def test_track_limit_depends_on_area():
    # area 90*90=8100 > 5000 -> limit 8 : a 7 px move keeps the old box, 9 px takes the new
    f, _, _ = orc.track_faces([[0, 0, 90, 90]], [0], 1, [[7, 0, 90, 90]])
    assert f.tolist() == [[0, 0, 90, 90]]
    f, _, _ = orc.track_faces([[0, 0, 90, 90]], [0], 1, [[9, 0, 90, 90]])
    assert f.tolist() == [[9, 0, 90, 90]]


def test_track_unmatched_old_dropped_new_appended():
    # old face far (>= track_threshold 40) from every new one is dropped; new faces get fresh ids
    f, ids, nid = orc.track_faces([[0, 0, 20, 20], [200, 200, 20, 20]], [3, 4], 7,
                                  [[201, 200, 20, 20], [100, 100, 20, 20]])
    assert f.tolist() == [[200, 200, 20, 20], [100, 100, 20, 20]]
    assert ids.tolist() == [4, 7] and nid == 8


def test_track_distance_truncates_and_threshold_is_strict():
    # centre distance sqrt(39^2+9^2)=40.02 -> 40, not < 40 -> unmatched
    f, ids, nid = orc.track_faces([[0, 0, 20, 20]], [0], 1, [[39, 9, 20, 20]])
    assert ids.tolist() == [1]
    f, ids, nid = orc.track_faces([[0, 0, 20, 20]], [0], 1, [[39, 8, 20, 20]])   # 39.8 -> 39
    assert ids.tolist() == [0]


# ----------------------------------------------------------- face stream
def test_face_stream_reference_mode_scales_boxes(orc_cascade):
    """640x480, width-to-process 160 -> scale 4 -> working image 160x120,
    emitted boxes are working-image boxes * norm_scale 4."""
    W, H = 640, 480
    bgr = synth.make_bgr(W, H, 11, "natural", [(160, 120, 240)])
    s = orc.FaceStream(orc_cascade)
    boxes, ids = s.process(bgr)
    small = orc.resize_linear(bgr, 160, 120)
    g = orc.equalize_hist(orc.bgr2gray(small))
    det = orc.detect_multiscale(orc_cascade, g, 1.25, 3, 0, (8, 6))
    assert len(det) >= 1
    assert np.array_equal(boxes, det * 4) and ids.tolist() == list(range(len(det)))


def test_face_stream_hysteresis_and_gating(orc_cascade):
    W, H = 320, 240
    face = synth.make_bgr(W, H, 1, "natural", [(80, 40, 120)])
    empty = synth.make_bgr(W, H, 2, "flat")
    s = orc.FaceStream(orc_cascade, width_to_process=320)
    b0, _ = s.process(face)
    assert len(b0) == 1
    b1, _ = s.process(empty)      # first empty frame: faces held (MAX_NUM_FPS_WITH_NO_DETECTION = 1)
    assert np.array_equal(b1, b0)
    b2, _ = s.process(empty)      # second empty frame: cleared
    assert len(b2) == 0
    # process-x-every-4-frames = 1: only frame 1 of each GOP of 4 is analysed
    s = orc.FaceStream(orc_cascade, width_to_process=320, process_x_every_4=1)
    outs = [len(s.process(f)[0]) for f in (empty, face, face, face, face)]
    assert outs == [0, 0, 0, 0, 1]


def test_face_stream_width_to_process_larger_than_frame_zeroes_boxes(orc_cascade):
    # W / width_to_process == 0 (integer): working image = frame, norm_scale = 0 (reference quirk)
    bgr = synth.make_bgr(128, 96, 1, "natural", [(24, 8, 80)])
    s = orc.FaceStream(orc_cascade, width_to_process=160)
    boxes, _ = s.process(bgr)
    assert len(boxes) >= 1 and not boxes.any()


# ----------------------------------------------------------- tracker pieces
def test_update_mhi():
    mhi = np.array([[0, 5.0, 9.9, 10.0]], np.float32)
    silh = np.array([[255, 0, 0, 0]], np.uint8)
    orc.update_mhi(silh, mhi, 10.0, 0.2)          # delbound = 9.8
    assert mhi.tolist() == [[10.0, 0.0, np.float32(9.9), 10.0]]


def test_segment_motion_components_and_order():
    ts = 100.0
    mhi = np.zeros((8, 10), np.float32)
    mhi[1:3, 6:9] = ts          # first seed in raster order (row 1)
    mhi[2:6, 1:3] = ts          # second
    mhi[6, 1] = ts - 20         # stale but within 32 of its neighbour above -> joins component 2
    mhi[7, 9] = ts - 20         # stale, isolated, no seed -> no component
    r = orc.segment_motion(mhi.copy(), ts)
    assert r.tolist() == [[6, 1, 3, 2], [1, 2, 2, 5]]


def test_segment_motion_threshold_chain():
    ts = 1000.0
    mhi = np.zeros((1, 5), np.float32)
    mhi[0] = [ts, ts - 30, ts - 60, ts - 100, 0]
    assert orc.segment_motion(mhi.copy(), ts).tolist() == [[0, 0, 3, 1]]


def test_join_objects():
    # area filter is strict on both sides; merge by centre distance < 35 into the lower index
    r = [[0, 0, 10, 10], [20, 0, 10, 10], [200, 200, 5, 5], [300, 0, 200, 200]]
    out = orc.join_objects(r)
    assert out.tolist() == [[0, 0, 30, 10]]
    # a box fully containing the other absorbs it
    out = orc.join_objects([[0, 0, 40, 40], [10, 10, 10, 10]])
    assert out.tolist() == [[0, 0, 40, 40]]


def test_tracker_moving_square():
    W, H = 160, 120
    trk = orc.Tracker()

    def frame(x):
        f = np.full((H, W, 4), 30, np.uint8)
        f[40:70, x:x + 30, :3] = 220
        return f

    assert len(trk.process(frame(20), 0.0)) == 0           # first frame only primes img_prev
    out = trk.process(frame(28), 33.0)
    # changed pixels: columns [20,28) and [50,58), rows [40,70): two 8x30 strips, centres 30 px apart
    # -> merged (distance 35) into one box spanning both
    assert out.tolist() == [[20, 40, 38, 30]]
    out = trk.process(frame(28), 66.0)                     # no motion
    assert len(out) == 0
