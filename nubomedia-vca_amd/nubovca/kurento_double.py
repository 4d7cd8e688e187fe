"""Stand-in for the Kurento server wrappers (modules/nubo_*/.../src/server/implementation/objects/Nubo*Impl.cpp),
which cannot be built here (kms-core and its code generator are absent).  The wrappers do two
things with an element: map remote methods onto GObject properties (NuboFaceDetectorImpl.cpp:158-237) and parse the
element's string signal into FaceInfo-like records (NuboFaceDetectorImpl.cpp:54-129).  This module restates both so
tests can check that what the shim emits is what the server layer expects to consume."""

# remote method -> element property (NuboFaceDetector.kmd.json / NuboFaceDetectorImpl.cpp:158-237)
FACE_METHODS = {
    "showFaces": "view-faces", "detectByEvent": "detect-event", "sendMetaData": "send-meta-data",
    "multiScaleFactor": "multi-scale-factor", "widthToProcess": "width-to-process",
    "processXevery4Frames": "process-x-every-4-frames", "euclideanDistance": "euclidean-distance",
    "trackThreshold": "track-threshold", "areaThreshold": "area-threshold", "activateServerEvents": "activate-events",
}
TRACKER_METHODS = {"setThreshold": "set_threshold", "setMinArea": "set_min_area", "setMaxArea": "set_max_area",
                   "setDistance": "set_distance", "setVisualMode": "set_visual_mode", "activateServerEvents": "activate-events"}


def parse_event_string(message, type_name="face"):
    """NuboFaceDetectorImpl::onFace: "x:..,y:..,width:..,height:..;..." -> [dict(name, x, y, width, height)].
    A record is completed by its `height` field; unknown keys are ignored; a trailing partial record is dropped."""
    out = []
    fields = []
    for face in message.split(";"):
        if face:
            fields += [f for f in face.split(",") if f]
    toks = []
    for f in fields:
        toks += [t for t in f.split(":") if t]
    cur = None
    for i in range(0, len(toks), 2):
        if cur is None:
            cur = dict(name=type_name, x=0, y=0, width=0, height=0)
        key = toks[i]
        if i + 1 >= len(toks):
            break
        if key in ("x", "y", "width"):
            cur[key] = int(toks[i + 1])
        elif key == "height":
            cur["height"] = int(toks[i + 1])
            out.append(cur)
            cur = None
    return out
