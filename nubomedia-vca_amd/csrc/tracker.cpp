// tracker.cpp -- NuboTracker stream (placeholder until the device path lands in this file)
#include "nvca_internal.h"

struct nvca_tracker { nvca_ctx *ctx; nvca_tracker_params p; };

extern "C" {
int nvca_tracker_create(nvca_ctx *ctx, const nvca_tracker_params *params, nvca_tracker **out)
{
    if (!ctx || !out) return NVCA_ERR_ARG;
    ctx->set_error("tracker: not implemented yet");
    return NVCA_ERR_UNSUPPORTED;
}
void nvca_tracker_destroy(nvca_tracker *t) { delete t; }
int nvca_tracker_set_params(nvca_tracker *t, const nvca_tracker_params *params) { return NVCA_ERR_UNSUPPORTED; }
int nvca_tracker_process(nvca_tracker *t, const nvca_frame *f, double ts, nvca_rect *out, int cap, int *n_out) { return NVCA_ERR_UNSUPPORTED; }
int nvca_tracker_batch_process(nvca_ctx *ctx, int n, nvca_tracker *const *trackers, const nvca_frame *frames,
                               const double *ts, nvca_rect *out, int cap, int *n_out) { return NVCA_ERR_UNSUPPORTED; }
}
