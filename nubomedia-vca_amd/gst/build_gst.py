"""Builds the GStreamer shim (libgstnubovca.so) and its test harness against the GStreamer 1.14
development files under /opt/conda.  Optional: skipped when they are absent."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
CONDA = os.environ.get("NVCA_GST_PREFIX", "/opt/conda")
INC = ["-I%s/include/gstreamer-1.0" % CONDA, "-I%s/include/glib-2.0" % CONDA, "-I%s/lib/glib-2.0/include" % CONDA,
       "-I%s/include" % ROOT]
LIBS = ["-L%s/lib" % CONDA, "-lgstvideo-1.0", "-lgstbase-1.0", "-lgstreamer-1.0", "-lgobject-2.0", "-lglib-2.0",
        # system libstdc++ must win over conda's older copy (libnubovca_hip needs GLIBCXX_3.4.29)
        "-Wl,--disable-new-dtags", "-Wl,-rpath,/usr/lib/x86_64-linux-gnu:%s/lib" % CONDA]
PLUGIN = os.path.join(HERE, "libgstnubovca.so")
HARNESS = os.path.join(HERE, "gst_harness")
# the reference's six plugins by their own library and plugin names (plugin_alias.cpp): (library, plugin name, element factory)
ALIAS_DIR = os.path.join(PKG, "gst_reference_names")        # NOT below gst/: GStreamer scans a plugin path recursively
ALIASES = [("nubofacedetector", "nubofacedetector", "nubofacedetector"), ("nuboeyedetector", "eyefilter", "nuboeyedetector"),
           ("nubonosedetector", "nubonosedetector", "nubonosedetector"), ("nubomouthdetector", "nubomouthdetector", "nubomouthdetector"),
           ("nuboeardetector", "earfilter", "nuboeardetector"), ("nubotracker", "nubotracker", "nubotracker")]


def available():
    return os.path.exists(os.path.join(CONDA, "include/gstreamer-1.0/gst/video/gstvideofilter.h"))


def _stale(out, srcs):
    return not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs)


def build(required=True):
    if not available():
        if required:
            raise RuntimeError("GStreamer development files not found under %s" % CONDA)
        return None
    src = os.path.join(HERE, "gstnubovca.cpp")
    hdr = os.path.join(ROOT, "include", "nubovca.h")
    if _stale(PLUGIN, [src, hdr, __file__]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-deprecated-declarations", src, "-o", PLUGIN]
                              + INC + LIBS + ["-L" + PKG, "-lnubovca_hip", "-Wl,-rpath,$ORIGIN/.."])
    alias_src = os.path.join(HERE, "plugin_alias.cpp")
    os.makedirs(ALIAS_DIR, exist_ok=True)
    for lib, plugin, factory in ALIASES:
        out = os.path.join(ALIAS_DIR, "lib%s.so" % lib)
        if _stale(out, [alias_src, PLUGIN, __file__]):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-DNVCA_ALIAS_PLUGIN=%s" % plugin,
                                   "-DNVCA_ALIAS_FACTORY=\"%s\"" % factory, "-DNVCA_ALIAS_LIB=%s" % lib]
                                  + (["-DNVCA_ALIAS_LIB_DIFFERS"] if lib != plugin else []) + [alias_src, "-o", out] + INC + LIBS
                                  + ["-L" + HERE, "-lgstnubovca", "-Wl,-rpath,$ORIGIN/../gst:$ORIGIN/.."])
    hs = os.path.join(HERE, "gst_harness.cpp")
    if _stale(HARNESS, [hs, __file__]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", hs, "-o", HARNESS] + INC + LIBS)
    return PLUGIN


def env(reference_names=False):
    """environment for running pipelines with the shim (reference_names: the six plugins under the reference's names instead of the one)"""
    e = dict(os.environ)
    e["GST_PLUGIN_PATH"] = ALIAS_DIR if reference_names else HERE
    e["GST_PLUGIN_SYSTEM_PATH"] = os.path.join(CONDA, "lib", "gstreamer-1.0")
    e["GST_REGISTRY"] = os.path.join("/tmp", "nubovca-gst-registry-%d%s.bin" % (os.getuid(), "-names" if reference_names else ""))
    sysstd = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"
    if os.path.exists(sysstd):          # conda's gst tools would otherwise pull conda's older libstdc++ first
        e["LD_PRELOAD"] = sysstd + (":" + e["LD_PRELOAD"] if e.get("LD_PRELOAD") else "")
    return e


if __name__ == "__main__":
    print(build(required=True))
