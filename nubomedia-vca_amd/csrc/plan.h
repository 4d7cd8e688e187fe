// plan.h -- per-(cascade, geometry) tables, host copy + device copy.
#pragma once
#include "nvca_internal.h"

namespace nvca {

void scale_grid(int ow, int oh, int cols, int rows, double scaleFactor, int minw, int minh,
                int maxw, int maxh, bool findBiggest, std::vector<double> &factors);
// cvSetImagesForHaarClassifierCascade for one factor, independent of the image geometry: corner columns / rows relative to
// the window, re-balanced weights, equRect.  Cached per (cascade, factor) in the context and shared by all plans.
struct ScaleTable {
    std::vector<TStumpRec> host;
    std::vector<LStumpRec> lhost;     // the same stumps, compact per-lane form (same order)
    DevBuf dev;                       // TStumpRec[n] followed by LStumpRec[n]
    const LStumpRec *d_lrecs = nullptr;
    // general cascades (tree weak classifiers / tilted features): every node at this factor, plus the cascade's leaf values and
    // first-node indices (cascade-level, kept with the table so that a plan finds everything in one place)
    std::vector<GNodeRec> ghost;
    std::vector<float> galpha;
    std::vector<int> gcls_first;
    DevBuf gdev;
    const GNodeRec *d_grecs = nullptr; const float *d_galpha = nullptr; const int *d_gcls_first = nullptr;
    int winw = 0, winh = 0, ex = 0, ey = 0, ew = 0, eh = 0;
    double inv_area = 0, factor = 0;
    int refs = 0; uint64_t last_use = 0;
    // distinct corner columns / rows of the stumps of stage ranges (cached per range: plans ask again and again)
    std::map<std::pair<int, int>, std::pair<std::vector<int>, std::vector<int>>> offsets;
    const std::pair<std::vector<int>, std::vector<int>> &corner_offsets(int k0, int k1, bool with_eq);
    ~ScaleTable() { dev.release(); gdev.release(); }
};
void build_scale_table(const Cascade &c, double factor, ScaleTable &t);
ScaleTable *get_scale_table(nvca_ctx *ctx, const Cascade &c, double factor);     // nullptr: allocation / copy failed (error set)
void free_scale_tables(nvca_ctx *ctx);
void build_stage_recs(const Cascade &c, std::vector<StageRec> &out);

// one scan grid handed to the evaluator
struct ScaleSpec {
    double table_factor;        // factor passed to cvSetImagesForHaarClassifierCascade (1 for pyramid levels)
    int plane_off, pitch;       // where this scale's integral planes live inside a slot
    int plane_rows;             // rows of those planes (image rows + 1)
    std::vector<int> xs, ys;    // window origins (plane coordinates) of the scan grid
    int adaptive;               // OpenCV's adaptive x step (scale-cascade branch) or plain grid (scale-image)
    // how a hit maps back to an image rectangle
    double out_factor;          // scale-image: Rect(cvRound(x*f), cvRound(y*f), out_w, out_h); 0: Rect(x, y, out_w, out_h)
    int out_w, out_h;
};

struct DetectPlan {
    // geometry of the working (gray) image the detector runs on
    int cols = 0, rows = 0, spitch = 0;
    int nstumps = 0;
    std::vector<ScaleRec> scales;
    std::vector<ScaleTable *> tabs;   // per scale (referenced: refs++ / refs-- in release_tables)
    void release_tables();
    std::vector<StageRec> stages;
    std::vector<float> stage_thr;     // StageRec::thr of every stage + padding (CascadeArgs::stage_thr)
    std::vector<int> stage_first;     // first stump of every stage, the stump count, INT_MAX padding (CascadeArgs::stage_first)
    std::vector<StripRec> strips;
    std::vector<int> pos;
    std::vector<unsigned> tasks; // stage-0 wave tasks
    bool device_group_ok = false;     // candidate rects are plain (x, y, winw, winh): k_group can rebuild them
    bool generic = false;             // tree weak classifiers / tilted features: evaluated by k_gen_stage0 + k_gen_rest
    bool needs_tilted = false;        // the cascade reads the tilted integral
    bool generic_stumps = false;      // general evaluator, but every weak classifier is a stump (the SSE2 pair policy applies)
    int deep_stage = 6;          // first stage run by k_deep; == stages.size(): the tile kernels walk the whole cascade (the default whenever the tiles can hold every stage's samples)
    std::vector<TileRec> tiles;  // LDS lattice tiles (k_tile); empty: row strips (k_strip)
    std::vector<int> tile_order; int tile_blocks_per_frame = 0;
    std::vector<unsigned short> tcoords;
    int tile_lds = 0;            // dynamic LDS bytes of the largest tile
    std::vector<DeepRec> deeprecs;  // per scale (k_deep LDS patches); empty: not used
    int deep_lds = 0;               // bytes of the largest patch
    std::vector<BandRec> bands;  // rows of tiles (k_band); usable when every scale is tiled
    std::vector<int> band_order; int band_blocks_per_frame = 0;
    std::vector<int> order;      // dispatch slot -> strip (-1 = padding); 8 equal-work chunks, one per XCD
    int blocks_per_frame = 0;
    // device copies
    DevBuf d_scales, d_stages, d_strips, d_pos, d_order, d_tasks, d_tiles, d_tile_order, d_tcoords, d_bands, d_band_order, d_deeprecs, d_stage_hint, d_stage_first, d_stage_thr, d_blob;   // the table buffers are views into d_blob (d_stage_hint: 8 words the tile kernels keep their stage statistics in, zero at upload)

    std::vector<ScaleSpec> specs;      // host copy (hit -> rectangle)
    int build_custom(nvca_ctx *ctx, const Cascade &c, std::vector<ScaleSpec> &&specs, bool allow_tiles, std::string &err);
    int build_tables(nvca_ctx *ctx, const Cascade &c, bool allow_tiles, std::string &err);      // from `specs` and `deep_stage`
    int key_sy = 13, key_ss = 26;                   // candidate key layout: scale << key_ss | iy << key_sy | ix (sized per plan in build_tables)
    int min_tile_side = 0, max_tile_side = 0;       // smallest tile side (windows) any scale got / the side asked for
    nvca_rect hit_rect(unsigned key) const;
    bool hit_valid(unsigned key) const;   // the key names a window of this plan's scan grids (a device result is checked before it indexes host tables)
    int build_scale_cascade(nvca_ctx *ctx, const Cascade &c, int cols, int rows, int pitch, double scaleFactor,
                            int minw, int minh, int maxw, int maxh, std::string &err);
    int upload(nvca_ctx *ctx);
    ~DetectPlan();
};

} // namespace nvca
