"""Stand-in for the Kurento server wrappers (modules/nubo_*/.../src/server/implementation/objects/Nubo*Impl.cpp),
which cannot be built here (kms-core and its code generator are absent).  A wrapper does two
things with its element: it maps remote methods onto GObject properties, and it parses the element's string signal
into <Kind>Info records that it re-emits as a server event.  This module restates both for all six elements so that the
tests can drive an element the way the server layer does (properties set by remote-method name) and check that what the
shim emits is what that layer consumes.

Reference, per element (path prefix modules/nubo_<kind>/.../src/server/implementation/objects/):
  face     NuboFaceDetectorImpl.cpp   onFace :55-129    methods :158-237   signal "face-event"    factory nubofacedetector
  eye      NuboEyeDetectorImpl.cpp    onEye :54-126     methods :154-219   signal "eye-event"     factory nuboeyedetector
  nose     NuboNoseDetectorImpl.cpp   onNose :50-123    methods :150-216   signal "nose-event"    factory nubonosedetector
  mouth    NuboMouthDetectorImpl.cpp  onMouth :52-125   methods :154-221   signal "mouth-event"   factory nubomouthdetector
  ear      NuboEarDetectorImpl.cpp    onEar :51-125     methods :155-221   signal "ear-event"     factory nuboeardetector
  tracker  NuboTrackerImpl.cpp        onTracker :50-126 methods :155-187   signal "tracker-event" factory nubotracker
"""

_COMMON = {"detectByEvent": ("detect-event",), "sendMetaData": ("send-meta-data",), "multiScaleFactor": ("multi-scale-factor",),
           "processXevery4Frames": ("process-x-every-4-frames",), "widthToProcess": ("width-to-process",),
           "activateServerEvents": ("activate-events", "events-ms")}        # (int activate, int ms): two properties, in this order


def _with(first, extra=None):
    m = dict(_COMMON)
    m.update(first)
    if extra:
        m.update(extra)
    return m


# kind -> what the wrapper knows about its element.  methods: remote method -> the properties its arguments are written to, in order.
# missing: the value a field keeps when the payload does not name it (EarInfo starts at -1, the others at 0);
# emit_empty: NuboEarDetectorImpl::onEar raises OnEar even with no record, the others only when at least one was completed.
ELEMENTS = {
    "face": dict(factory="nubofacedetector", signal="face-event", record="face", missing=0, emit_empty=False,
                 methods=_with({"showFaces": ("view-faces",)}, {"euclideanDistance": ("euclidean-distance",), "trackThreshold": ("track-threshold",),
                                                                 "areaThreshold": ("area-threshold",)})),
    "eye": dict(factory="nuboeyedetector", signal="eye-event", record="eye", missing=0, emit_empty=False, methods=_with({"showEyes": ("view-eyes",)})),
    "nose": dict(factory="nubonosedetector", signal="nose-event", record="nose", missing=0, emit_empty=False, methods=_with({"showNoses": ("view-noses",)})),
    "mouth": dict(factory="nubomouthdetector", signal="mouth-event", record="mouth", missing=0, emit_empty=False, methods=_with({"showMouths": ("view-mouths",)})),
    "ear": dict(factory="nuboeardetector", signal="ear-event", record="ear", missing=-1, emit_empty=True, methods=_with({"showEars": ("view-ears",)})),
    "tracker": dict(factory="nubotracker", signal="tracker-event", record="tracker", missing=0, emit_empty=False,
                    methods={"setThreshold": ("set_threshold",), "setMinArea": ("set_min_area",), "setMaxArea": ("set_max_area",),
                             "setDistance": ("set_distance",), "setVisualMode": ("set_visual_mode",),
                             "activateServerEvents": ("activate-events", "events-ms")}),
}

# kept for the callers of round 2 / 3
FACE_METHODS = {k: v[0] for k, v in ELEMENTS["face"]["methods"].items()}
TRACKER_METHODS = {k: v[0] for k, v in ELEMENTS["tracker"]["methods"].items()}


def split_message(fi, delimiter):
    """Nubo*Impl::split_message: every token up to each delimiter, then the remainder -- EMPTY tokens included (a payload that
    ends in ';' yields a trailing ''), which is why a stray separator shifts the key / value parity of everything behind it."""
    out = []
    while True:
        pos = fi.find(delimiter)
        if pos < 0:
            break
        out.append(fi[:pos])
        fi = fi[pos + len(delimiter):]
    out.append(fi)
    return out


def remote_call(kind, method, *args):
    """What the wrapper's method writes: [(property, value), ...] in the order of its g_object_set calls.
    NuboTrackerImpl::setMaxArea takes a float and stores it as a long (NuboTrackerImpl.cpp:166-171)."""
    props = ELEMENTS[kind]["methods"][method]
    if len(args) != len(props):
        raise TypeError("%s.%s takes %d argument(s)" % (kind, method, len(props)))
    vals = [int(a) for a in args]          # every remote argument lands in an integer property
    return list(zip(props, vals))


def harness_props(kind, calls):
    """[(method, args...), ...] -> the prop=value arguments gst_harness sets through GObject, in order"""
    out = []
    for c in calls:
        out += ["%s=%d" % pv for pv in remote_call(kind, c[0], *c[1:])]
    return out


def parse_event(kind, message):
    """Nubo<Kind>DetectorImpl::on<Kind> / NuboTrackerImpl::onTracker: "x:..,y:..,width:..,height:..;..." -> (records, raised).
    The payload is split at ';', every piece at ',', every piece of that at ':' into ONE flat token list that is walked two at a
    time: token i is a key, token i + 1 its value.  A record is completed (and appended) by its `height` key; keys other than
    x / y / width / height are skipped with their value; a trailing partial record is dropped; std::stoi parses the values
    (leading blanks and a sign allowed, anything else throws -- here ValueError).  raised: whether the server event goes out."""
    e = ELEMENTS[kind]
    all_ = []
    for piece in split_message(message, ";"):
        for field in split_message(piece, ","):
            all_ += split_message(field, ":")
    out, cur = [], None
    for i in range(0, len(all_), 2):
        if cur is None:
            cur = dict(name=e["record"], x=e["missing"], y=e["missing"], width=e["missing"], height=e["missing"])
        key = all_[i]
        if key in ("x", "y", "width", "height") and i + 1 < len(all_):
            cur[key] = _stoi(all_[i + 1])
            if key == "height":
                out.append(cur)
                cur = None
    return out, bool(out) or e["emit_empty"]


def _stoi(s):
    """std::stoi: optional leading whitespace, optional sign, decimal digits; stops at the first other character"""
    t = s.lstrip(" \t\n\v\f\r")
    j = 1 if t[:1] in "+-" else 0
    k = j
    while k < len(t) and t[k].isdigit():
        k += 1
    if k == j:
        raise ValueError("stoi: no conversion in %r" % s)
    return int(t[:k])


def parse_event_string(message, type_name="face"):
    """round 2 / 3 name: the records of parse_event for the element whose record type is `type_name`"""
    kind = next(k for k, v in ELEMENTS.items() if v["record"] == type_name)
    return parse_event(kind, message)[0]
