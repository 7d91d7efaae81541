// plade_amd/csrc/ransac.h -- GPU plane extraction (seam S1b): a device-driven Efficient-RANSAC that extracts the
// planes of up to sixteen clouds (the scans of one to eight pairs) in the same launch sequence, see ransac.hip.
#pragma once
#include "ctx.h"

namespace plade {

struct RansacParams {
    uint32_t min_support = 10000;
    float dist_rel = 0.005f, bitmap_rel = 0.02f, cos_thresh = 0.8f, overlook_p = 0.001f;  // plade.cpp:607
    int orient_normals = 0;     // 0 = reference behaviour (plane_extraction.cpp:43-58 never flips), see plade_hip.h
    uint64_t seed = 0;
    bool host_indices = true;   // also copy the inlier index lists to the host (PlaneSetOut::idx)
};

struct PlaneSetOut {
    std::vector<float> coef;       // P x 4 (unit n, d = -n.p)
    std::vector<int32_t> offsets;  // P + 1
    std::vector<int32_t> idx;      // original point indices
    const uint32_t *d_idx = nullptr;  // the same list on the device (valid until the next detect of this cloud slot)
    // ... and as positions in the extraction's Morton-ordered SoA copy of the cloud (x | y | z planes m_x, m_y, m_z): the lists
    // are ascending positions, so a stage that gathers the planes' points reads that copy almost sequentially
    const uint32_t *d_pos = nullptr;
    const float *m_x = nullptr, *m_y = nullptr, *m_z = nullptr;
    uint32_t n_score_passes = 0;   // full-array K1 launches that scanned this cloud (roofline bookkeeping)
    double score_bytes = 0;        // their algorithmic bytes: 28 B/point per launch + 1 mask byte per 4 points per chain
    uint32_t remaining = 0;
    uint32_t P() const { return (uint32_t)(coef.size() / 4); }
    // make `idx` valid when the detect call skipped the host copy
    void fetch_indices(hipStream_t stream) {
        const size_t m = offsets.empty() ? 0 : (size_t)offsets.back();
        if (idx.size() == m || !d_idx) return;
        idx.resize(m);
        if (m) {
            (void)hipMemcpyAsync(idx.data(), d_idx, 4 * m, hipMemcpyDeviceToHost, stream);
            (void)hipStreamSynchronize(stream);
        }
    }
};

struct RansacWork;
RansacWork *ransac_work_create();
void ransac_work_destroy(RansacWork *w);

// One slot of the acceptance chain on a caller-given score list (seam S1c): connected component of the
// list in the plane's bitmap, LS fit of the kept points, weighted score against the input plane.
struct ComponentOut {
    std::vector<int32_t> kept;
    float fit[7];      // unit normal, mean, dist
    double wscore;
    uint32_t err;      // 1: bitmap too large
};
void plane_component(plade_ctx *ctx, RansacWork &W, const CloudDev &cloud, const float normal[3], const float point[3],
                     const int32_t *idx, uint32_t m, float bitmap_eps, bool closing_filter, float w_eps, ComponentOut &out);

// Up to sixteen clouds are extracted together ("slots" 0-15 of the work area): the two scans of a registration, or the 2 x count
// scans of a group of registrations (plade_registration_pairs), in one launch sequence.
constexpr int RANSAC_SLOTS = 2 * PLADE_GROUP_MAX;

// Morton order + stratified subset of the clouds (once per set of clouds; every detect call on them reuses it).
void ransac_prepare(plade_ctx *ctx, RansacWork &W, const CloudDev *const clouds[RANSAC_SLOTS], int n_clouds);

// average_spacing (code/PLADE/util.cpp:1619-1648) of the clouds in `slots` from their Morton order: two kernels (for all of them) queued on the
// context's stream behind ransac_prepare; ransac_spacing_finish waits for them unless a later wait on that stream has completed (false: not available
// -- nothing queued, or a cloud too clumped for the octree cells -- use average_spacing_dev).
void ransac_spacing_enqueue(plade_ctx *ctx, RansacWork &W, const int *slots, int count, int k, uint32_t samples);
bool ransac_spacing_finish(plade_ctx *ctx, RansacWork &W, int slot, float *spacing_out);

// One PlaneExtraction::detect (code/PLADE/plane_extraction.cpp:173-200) per ACTIVE slot, all in one launch sequence:
// slots with active[s] == false keep the results of their previous detect call untouched.
struct RansacJob {
    bool active = false;
    RansacParams rp;
    PlaneSetOut *out = nullptr;
    Stats *stats = nullptr;     // where this cloud's counters go (nullptr: the calling context's)
};
void ransac_detect_prepared(plade_ctx *ctx, RansacWork &W, RansacJob jobs[RANSAC_SLOTS]);

// prepare + detect of a single cloud
void ransac_detect(plade_ctx *ctx, RansacWork &W, const CloudDev &cloud, const RansacParams &rp, PlaneSetOut &out);

// Seam S1a (plade_score_planes, plade_score_planes_subset): the caller's hypotheses through the loop's own K1 kernels --
// counts by k_r_rescore, ordered inlier lists by k_r_mark + k_r_compact_raster, subset counts by k_r_score_sub.
// The cloud is scanned in the caller's point order (no Morton pass), d_assigned = shapeIndex per point or nullptr.
void score_planes_seam(plade_ctx *ctx, RansacWork &W, const CloudDev &cloud, const int32_t *d_assigned, const float *planes,
                       uint32_t h, float eps, float cos_t, uint32_t *counts, uint32_t *idx_out, uint32_t cap);
void score_subset_seam(plade_ctx *ctx, RansacWork &W, const CloudDev &cloud, const int32_t *d_assigned, const uint32_t *sub_index,
                       uint32_t m, const float *planes, uint32_t h, float eps, float cos_t, uint32_t *counts, uint32_t *n_unassigned);

}  // namespace plade
