// gst_harness.cpp -- test driver for the GStreamer shim: pushes raw frames from a file through
//   filesrc ! rawvideoparse ! <element> ! fakesink
// and prints what an unmodified downstream would observe: every CUSTOM_DOWNSTREAM "message" event
// leaving the element (one line per frame: "event <pts> x,y,w,h;...") and every string signal
// ("signal <payload>").  Usage:
//   gst_harness <element> <format BGR|BGRA> <width> <height> <frames.raw>[,<more.raw>...] [prop=value ...]
// With several files the pipeline has one such branch per file (lines of branch k > 0 are tagged "event#k" / "signal#k").
#include <gst/gst.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>

static GstPadProbeReturn on_event(GstPad *, GstPadProbeInfo *info, gpointer user)
{
    GstEvent *ev = GST_PAD_PROBE_INFO_EVENT(info);
    if (GST_EVENT_TYPE(ev) != GST_EVENT_CUSTOM_DOWNSTREAM) return GST_PAD_PROBE_OK;
    const GstStructure *m = gst_event_get_structure(ev);
    if (!m || !(gst_structure_has_name(m, "message") || gst_structure_has_name(m, "noses"))) return GST_PAD_PROBE_OK;
    guint64 pts = 0;
    std::string line;
    const gint n = gst_structure_n_fields(m);
    for (gint i = 0; i < n; i++) {
        const gchar *name = gst_structure_nth_field_name(m, i);
        GstStructure *sub = NULL;
        if (!gst_structure_get(m, name, GST_TYPE_STRUCTURE, &sub, NULL) || !sub) continue;
        if (!strcmp(name, "timestamp")) gst_structure_get(sub, "pts", G_TYPE_UINT64, &pts, NULL);
        else {
            guint x = 0, y = 0, w = 0, h = 0;
            gst_structure_get(sub, "x", G_TYPE_UINT, &x, "y", G_TYPE_UINT, &y, "width", G_TYPE_UINT, &w, "height", G_TYPE_UINT, &h, NULL);
            const gchar *type = gst_structure_get_string(sub, "type");
            char b[160]; snprintf(b, sizeof(b), "%s/%s:%u,%u,%u,%u;", gst_structure_get_name(sub), type ? type : "?", x, y, w, h);
            line += b;
        }
        gst_structure_free(sub);
    }
    const int k = GPOINTER_TO_INT(user);
    char tag[24] = "event";
    if (k) snprintf(tag, sizeof(tag), "event#%d", k);
    printf("%s %llu %s\n", tag, (unsigned long long)pts, line.c_str());
    fflush(stdout);
    return GST_PAD_PROBE_OK;
}
// NVCA_HARNESS_DUMP=<file>: the frames leaving branch 0's element are appended to it (what a viewer would see)
static GstPadProbeReturn on_buffer(GstPad *, GstPadProbeInfo *info, gpointer user)
{
    GstBuffer *buf = GST_PAD_PROBE_INFO_BUFFER(info);
    GstMapInfo map;
    if (buf && gst_buffer_map(buf, &map, GST_MAP_READ)) {
        fwrite(map.data, 1, map.size, (FILE *)user);
        fflush((FILE *)user);
        gst_buffer_unmap(buf, &map);
    }
    return GST_PAD_PROBE_OK;
}
static void on_signal(GstElement *, const gchar *payload, gpointer user)
{
    const int k = GPOINTER_TO_INT(user);
    if (k) printf("signal#%d %s\n", k, payload); else printf("signal %s\n", payload);
    fflush(stdout);
}

struct FrameData { std::vector<unsigned char> bytes; size_t next = 0; };   // frames of the file, handed out in a cycle
static GstPadProbeReturn on_fill(GstPad *, GstPadProbeInfo *info, gpointer user)
{
    FrameData *fd = (FrameData *)user;
    GstBuffer *buf = gst_buffer_make_writable(GST_PAD_PROBE_INFO_BUFFER(info));
    GST_PAD_PROBE_INFO_DATA(info) = buf;
    GstMapInfo map;
    if (gst_buffer_map(buf, &map, GST_MAP_WRITE)) {
        const size_t nfr = std::max<size_t>(fd->bytes.size() / std::max<size_t>(map.size, 1), 1), at = (fd->next++ % nfr) * map.size;
        if (at < fd->bytes.size()) memcpy(map.data, fd->bytes.data() + at, std::min((size_t)map.size, fd->bytes.size() - at));
        gst_buffer_unmap(buf, &map);
    }
    return GST_PAD_PROBE_OK;
}

// one branch: filesrc ! rawvideoparse ! <element or bin> ! fakesink, observed on the element named "el"
static int add_branch(GstElement *pipe, int k, int argc, char **argv, const char *file)
{
    // NVCA_HARNESS_LOOP=<n>: the file holds one frame, which is pushed n times (throughput runs without gigabyte files):
    // videotestsrc provides the buffers, a probe copies the frame into each (the cost filesrc's read would have)
    const char *loop = getenv("NVCA_HARNESS_LOOP");
    GstElement *src = gst_element_factory_make(loop ? "videotestsrc" : "filesrc", NULL);
    GstElement *parse = gst_element_factory_make(loop ? "capsfilter" : "rawvideoparse", NULL);
    // argv[1]: a factory name, or a bin description ("a ! b name=el ...") whose element named "el" is observed
    GstElement *sink = gst_element_factory_make("fakesink", NULL);
    GstElement *el = NULL, *chain = NULL;
    if (strchr(argv[1], '!')) {
        GError *err = NULL;
        chain = gst_parse_bin_from_description(argv[1], TRUE, &err);
        if (!chain) { fprintf(stderr, "bad description: %s\n", err ? err->message : "?"); return 3; }
        el = gst_bin_get_by_name(GST_BIN(chain), "el");
    } else {
        el = gst_element_factory_make(argv[1], NULL);
        chain = el;
    }
    if (!src || !parse || !el || !sink) { fprintf(stderr, "missing element (%s?)\n", argv[1]); return 3; }
    const bool bgra = !strcmp(argv[2], "BGRA");
    if (loop) {
        FrameData *fd = new FrameData();
        FILE *fp = fopen(file, "rb");
        if (fp) { fseek(fp, 0, SEEK_END); fd->bytes.resize((size_t)ftell(fp)); fseek(fp, 0, SEEK_SET); if (fread(fd->bytes.data(), 1, fd->bytes.size(), fp) != fd->bytes.size()) fd->bytes.clear(); fclose(fp); }
        gst_util_set_object_arg(G_OBJECT(src), "pattern", "black");
        g_object_set(src, "num-buffers", atoi(loop), NULL);
        GstCaps *caps = gst_caps_new_simple("video/x-raw", "format", G_TYPE_STRING, bgra ? "BGRA" : "BGR", "width", G_TYPE_INT, atoi(argv[3]),
                                            "height", G_TYPE_INT, atoi(argv[4]), "framerate", GST_TYPE_FRACTION, 30, 1, NULL);
        g_object_set(parse, "caps", caps, NULL);
        gst_caps_unref(caps);
        GstPad *vp = gst_element_get_static_pad(src, "src");
        gst_pad_add_probe(vp, GST_PAD_PROBE_TYPE_BUFFER, on_fill, fd, NULL);
        gst_object_unref(vp);
    } else {
        g_object_set(src, "location", file, NULL);
        gst_util_set_object_arg(G_OBJECT(parse), "format", bgra ? "bgra" : "bgr");
        g_object_set(parse, "width", atoi(argv[3]), "height", atoi(argv[4]), NULL);
        gst_util_set_object_arg(G_OBJECT(parse), "framerate", "30/1");
    }
    for (int i = 6; i < argc; i++) {
        std::string kv(argv[i]);
        const size_t eq = kv.find('=');
        if (eq == std::string::npos) continue;
        gst_util_set_object_arg(G_OBJECT(el), kv.substr(0, eq).c_str(), kv.substr(eq + 1).c_str());
        // what the element holds afterwards, read back through GObject ("prop <name>=<value>"): a server wrapper's remote method is a
        // g_object_set on the property it names -- the test double (nubovca/kurento_double.py) checks the round trip
        if (k == 0) {
            GParamSpec *ps = g_object_class_find_property(G_OBJECT_GET_CLASS(el), kv.substr(0, eq).c_str());
            if (!ps) printf("prop %s=<no such property>\n", kv.substr(0, eq).c_str());
            else {
                GValue v = G_VALUE_INIT;
                g_value_init(&v, G_PARAM_SPEC_VALUE_TYPE(ps));
                g_object_get_property(G_OBJECT(el), ps->name, &v);
                gchar *txt = g_strdup_value_contents(&v);
                printf("prop %s=%s\n", kv.substr(0, eq).c_str(), txt);      // under the name the caller used (GObject holds it with '-' for '_')
                g_free(txt); g_value_unset(&v);
            }
            fflush(stdout);
        }
    }
    gst_bin_add_many(GST_BIN(pipe), src, parse, chain, sink, NULL);
    if (!gst_element_link_many(src, parse, chain, sink, NULL)) { fprintf(stderr, "link failed\n"); return 4; }
    GstPad *sp = gst_element_get_static_pad(el, "src");
    gst_pad_add_probe(sp, GST_PAD_PROBE_TYPE_EVENT_DOWNSTREAM, on_event, GINT_TO_POINTER(k), NULL);
    if (k == 0 && getenv("NVCA_HARNESS_DUMP")) {
        FILE *dump = fopen(getenv("NVCA_HARNESS_DUMP"), "wb");
        if (dump) gst_pad_add_probe(sp, GST_PAD_PROBE_TYPE_BUFFER, on_buffer, dump, NULL);
    }
    gst_object_unref(sp);
    const char *sigs[] = {"face-event", "tracker-event", "eye-event", "nose-event", "mouth-event", "ear-event"};
    for (const char *sig : sigs)
        if (g_signal_lookup(sig, G_OBJECT_TYPE(el))) g_signal_connect(el, sig, G_CALLBACK(on_signal), GINT_TO_POINTER(k));
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s element format width height file[,file...] [prop=value...]\n", argv[0]); return 2; }
    gst_init(&argc, &argv);
    GstElement *pipe = gst_pipeline_new("p");
    // several files => several branches in the one pipeline, each with its own streaming thread (one per "session")
    std::string files(argv[5]);
    int k = 0;
    for (size_t pos = 0; pos <= files.size();) {
        size_t c = files.find(',', pos);
        if (c == std::string::npos) c = files.size();
        const std::string file = files.substr(pos, c - pos);
        if (!file.empty()) { const int rc = add_branch(pipe, k++, argc, argv, file.c_str()); if (rc) return rc; }
        pos = c + 1;
    }
    gst_element_set_state(pipe, GST_STATE_PLAYING);
    GstBus *bus = gst_element_get_bus(pipe);
    GstMessage *msg = gst_bus_timed_pop_filtered(bus, 120 * GST_SECOND, (GstMessageType)(GST_MESSAGE_EOS | GST_MESSAGE_ERROR));
    int rc = 0;
    if (!msg) { fprintf(stderr, "timeout\n"); rc = 5; }
    else if (GST_MESSAGE_TYPE(msg) == GST_MESSAGE_ERROR) {
        GError *e = NULL; gst_message_parse_error(msg, &e, NULL);
        fprintf(stderr, "pipeline error: %s\n", e ? e->message : "?"); rc = 6;
    }
    if (msg) gst_message_unref(msg);
    gst_element_set_state(pipe, GST_STATE_NULL);
    gst_object_unref(bus); gst_object_unref(pipe);
    printf("done %d\n", rc);
    return rc;
}
