// plugin_alias.cpp -- one of the reference's six GStreamer plugins by its own name: a stub that registers its element out of the
// shim library (libgstnubovca.so, a link-time dependency: element types, batching state and contexts exist once per process).
// Built six times by build_gst.py with -DNVCA_ALIAS_PLUGIN=<plugin name> -DNVCA_ALIAS_FACTORY="<element>":
//   libnubofacedetector.so  nubofacedetector / nubofacedetector   (modules/nubo_face/nubo-face-detector/src/gst-plugins/nubofacedetector.c:39-43)
//   libnuboeyedetector.so   eyefilter        / nuboeyedetector    (modules/nubo_eye/.../nuboeyedetector.c:16-20)
//   libnubonosedetector.so  nubonosedetector / nubonosedetector   (modules/nubo_nose/.../nubonosedetector.c:16-20)
//   libnubomouthdetector.so nubomouthdetector / nubomouthdetector (modules/nubo_mouth/.../nubomouthdetector.c:38-42)
//   libnuboeardetector.so   earfilter        / nuboeardetector    (modules/nubo_ear/.../nuboeardetector.c:37-41)
//   libnubotracker.so       nubotracker      / nubotracker        (modules/nubo_tracker/.../nubotracker.c:15-19)
#include <gst/gst.h>

extern "C" gboolean nvca_gst_register_element(GstPlugin *plugin, const char *factory);

static gboolean alias_init(GstPlugin *plugin) { return nvca_gst_register_element(plugin, NVCA_ALIAS_FACTORY); }

#ifndef PACKAGE
#define PACKAGE "nubovca"
#endif
GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, NVCA_ALIAS_PLUGIN, "NUBOMEDIA-VCA filter on MI355X (HIP), under the reference's plugin name", alias_init,
                  "0.1", "LGPL", "nubovca-hip", "https://github.com/nubomedia/NUBOMEDIA-VCA")

// GStreamer >= 1.14 looks a plugin's description up as gst_plugin_<file name>_get_desc; two of the reference's plugins carry a
// name that is not their file's (libnuboeyedetector.so is plugin "eyefilter", libnuboeardetector.so is "earfilter"): the
// description under the file's name as well
#ifdef NVCA_ALIAS_LIB_DIFFERS
#define NVCA_CAT3_(a, b, c) a##b##c
#define NVCA_CAT3(a, b, c) NVCA_CAT3_(a, b, c)
extern "C" __attribute__((visibility("default"))) const GstPluginDesc *NVCA_CAT3(gst_plugin_, NVCA_ALIAS_LIB, _get_desc)(void)
{
    return NVCA_CAT3(gst_plugin_, NVCA_ALIAS_PLUGIN, _get_desc)();
}
#endif
