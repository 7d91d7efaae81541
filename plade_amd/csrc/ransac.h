// plade_amd/csrc/ransac.h -- GPU plane extraction (seam S1b).
#pragma once
#include "ctx.h"

namespace plade {

struct RansacParams {
    uint32_t min_support = 10000;
    float dist_rel = 0.005f, bitmap_rel = 0.02f, cos_thresh = 0.8f, overlook_p = 0.001f;  // plade.cpp:607
    int orient_normals = 0;
    uint64_t seed = 0;
    bool host_indices = true;   // also copy the inlier index lists to the host (PlaneSetOut::idx)
};

struct PlaneSetOut {
    std::vector<float> coef;       // P x 4 (unit n, d = -n.p)
    std::vector<int32_t> offsets;  // P + 1
    std::vector<int32_t> idx;      // original point indices
    const uint32_t *d_idx = nullptr;  // the same list on the device (valid until the next detect on this work area)
    uint32_t n_score_passes = 0;   // full-array K1 passes issued (roofline bookkeeping)
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

// Coordinator of the acceptance batches of two concurrent detect calls (the two scans of a pair): their batches are
// launched together, see ransac.hip.  `who` = 0 (target) / 1 (source).
struct PairAccept;
PairAccept *pair_accept_create();
void pair_accept_destroy(PairAccept *p);

// PlaneExtraction::detect (code/PLADE/plane_extraction.cpp:173-200)
void ransac_detect(plade_ctx *ctx, RansacWork &W, const CloudDev &cloud, const RansacParams &rp, PlaneSetOut &out,
                   PairAccept *pair = nullptr, int who = 0);

}  // namespace plade
