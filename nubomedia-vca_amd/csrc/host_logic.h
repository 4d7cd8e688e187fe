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

// part detectors' merging heuristics (EYE/kmseyedetect.cpp:778-913, NOSE/kmsnosedetect.cpp:745-790, MOUTH/kmsmouthdetect.cpp:750-796)
void merge_consecutive_nm(std::vector<nvca_rect> &cn, const std::vector<nvca_rect> &old, const nvca_rect &face, int scale, int dis, std::vector<nvca_rect> &res);
bool contain_bb(int px, int py, const nvca_rect &r);
void merge_eyes_current(const nvca_rect &face_bb, const std::vector<nvca_rect> &eye_r, std::vector<nvca_rect> &eyes, int scale, bool eye_left);
void merge_eyes_consecutive(std::vector<nvca_rect> &ce, const std::vector<nvca_rect> &old, std::vector<nvca_rect> &res);
void to_global(std::vector<nvca_rect> &v, const nvca_rect &face, int scale);

} // namespace nvca
