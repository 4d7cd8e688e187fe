// host_logic.h -- sequential host-side pieces (see host_logic.cpp)
#pragma once
#include "nvca_internal.h"

namespace nvca {

void group_rectangles(std::vector<nvca_rect> &rects, int groupThreshold, double eps,
                      std::vector<int> *weights = nullptr);

struct TrackedFace { nvca_rect box; int id; };
struct Faces {                 // FACE/Faces.hpp: the per-stream list with ids
    std::vector<TrackedFace> faces;
    int next_id = 0;
    void track(const std::vector<nvca_rect> &current, int track_threshold);
    void clear() { faces.clear(); }
};

void join_objects(std::vector<nvca_rect> &seg_bounds, int min_area, long max_area, int distance);

} // namespace nvca
