// plade_amd/csrc/stages.h -- device stages of registration(T, target, source, target_planes,
// source_planes) (code/PLADE/plade.cpp:31-580) between the seam kernels.
#pragma once
#include "ctx.h"
#include "geom.h"
#include <functional>

namespace plade {

// ---- K4: intersection-line pair tables + "22" descriptors ----------------------
// Host-prepared line table.  The reference re-normalises the stored direction vector in place
// every time a line takes part in ComputeNearstTwoPointsOfTwo3DLine (util.cpp:1171-1172 through
// non-const references), so a pair (i, j) sees the k-th normalisation iterate of each line with k
// depending on the call order.  `iter` holds, per line, the iterates number 3, 4, ... until the
// sequence repeats (fixed point: period 1); iterate k >= 3 is iter[off + idx(k)].
struct LineTableHost {
    std::vector<float> pt;        // L x 3
    std::vector<int32_t> sp;      // L x 2 support planes
    std::vector<float> iter;      // concatenated iterates (x,y,z)
    std::vector<int32_t> it_off, it_len, it_start, it_period;  // per line
    uint32_t L = 0;
};

struct PairTableDev {
    uint32_t count = 0;           // D: number of descriptors kept (valid after the sync() that follows build_pair_table)
    DBuf<float> desc;             // D x 8
    DBuf<float> lv1, lv2, p1;     // D x 3 each: PAIRLINE::lineVec1, lineVec2, linePoints1
    // scratch
    DBuf<float> all_desc, all_lv1, all_lv2, all_p1;
    DBuf<uint32_t> flags, pos;
    DBuf<float> cp;               // closest_point_mode = 1: (q1, q2) of every pair i < j from the 9 x 9 solver, slot i * L + j
    DBuf<uint32_t> d_blob;        // line table in one upload: pt | sp | iterates | it | normals (4-byte words)
};

// target = true: all ordered pairs i != j (ConstructPairLinesKdTree, util.cpp:774-826);
// target = false: pairs i < j (plade.cpp:454-482, 511-521; util.cpp:133-168).
void build_pair_table(plade_ctx *ctx, const LineTableHost &lt, const float *normals /*P x 3 host*/, uint32_t P,
                      float scale, bool target, PairTableDev &out);

// ---- K6 + clustering + K7 ---------------------------------------------------------------------
struct CandidateSet {
    uint32_t m = 0;               // initial transforms (one per match)
    DBuf<float4> rt;              // m x 4 float4: rows of [R | T] then (roll, pitch, yaw, 0)
    uint32_t n_clusters = 0;
    DBuf<uint32_t> parent, sizes_all, seeds, sizes;  // seeds/sizes: n_clusters, seed ascending
    DBuf<uint32_t> flags, pos;
    DBuf<uint64_t> ckeys, ckeys2;
    DBuf<uint32_t> cvals, cvals2;
    DBuf<float> t_minmax;         // per-workgroup bbox partials of the translations (6 each)
    DBuf<float4> st, se;          // nodes in cell order: (T, index bits), Euler angles
    DBuf<uint2> spans;            // per cell head: 9 row spans of its 27-neighbourhood
    DBuf<int32_t> plane_counts;   // per cluster (seed order): matched plane pairs, -1 = centre gate failed
};

// one (R, T) per (query, neighbour) in match order (util.cpp:303-327)
void build_transforms(plade_ctx *ctx, const PairTableDev &src, const PairTableDev &tgt, const uint32_t *d_q_idx,
                      const uint32_t *d_t_idx, uint32_t m, CandidateSet &cs);
// ClusterTransformation (util.cpp:1245-1277): connected components under
// |T_a - T_b|^2 < float(r^2)  &&  |euler_a - euler_b|^2 < angle_gate
void cluster_transforms(plade_ctx *ctx, CandidateSet &cs, float dist_threshold, float angle_gate);

struct PlaneGeomHost {            // per cloud side
    std::vector<float> coef;      // P x 4
    std::vector<float> center;    // P x 3
    std::vector<float> radius;    // P
    std::vector<float> four;      // P x 4 x 3
    uint32_t P = 0;
};
// plane-consistency count per cluster seed (util.cpp:359-401)
void plane_consistency(plade_ctx *ctx, CandidateSet &cs, const PlaneGeomHost &src, const PlaneGeomHost &tgt,
                       const float src_bcenter[3], const float tgt_bcenter[3], float max_radius, float cos_angle_th,
                       float length_threshold);

// ---- penetration filter (util.cpp:450-519, AreTwoPlanesPenetrable :1279-1458) -------------------
// in-plane frame + cell table of one plane cloud (penetration walk): cells of edge `cell` over the
// plane's bounding rectangle, (u, v) measured from its first corner along its two edges
struct PenFrame {
    float o[3], eu[3], ev[3];
    int nu, nv;
    uint32_t base;                // first cell of this plane in the side's cell table
};
struct PlaneCloudsDev {           // per-plane voxel-downsampled points, concatenated
    DBuf<float> xyz;              // total x 3
    std::vector<uint32_t> off;    // P + 1 (host)
    DBuf<uint32_t> d_off;
    // the same points binned into the in-plane grids (built by penetration_filter)
    DBuf<PenFrame> frames;        // P
    DBuf<float4> cell_pts;        // total, in cell order
    DBuf<uint32_t> cell_start;    // n_cells + 1
    DBuf<uint32_t> ckeys, ckeys2, cvals, cvals2;
    uint32_t n_cells = 0;
    float grid_cell = 0.f;        // cell edge the grid was built with (0 = not built)
};
// cell edge of the in-plane grids: twice the walk's search radius (= lengthThreshold, util.cpp:1279)
inline float pen_grid_cell(float length_threshold) { return 2.f * (float)(double)length_threshold; }
struct PlaneGeomHost;
void build_pen_grid(plade_ctx *ctx, PlaneCloudsDev &pc, const PlaneGeomHost &geom, float cell);
// (pipeline) the candidates' table is gathered on the device from `ids` (K words): the ids are uploaded together with the
// filter's own tables and `launch` queues the gather, given their device address, in front of the filter's kernels
struct PenGather { const uint32_t *ids; std::function<void(const uint32_t *d_ids)> launch; };
// flags_out[k] = 1 when candidate k has a penetrating plane pair
void penetration_filter(plade_ctx *ctx, const float *cand_rt_host /*K x 12: R row-major, T*/, uint32_t K,
                        const PlaneGeomHost &src, const PlaneGeomHost &tgt, PlaneCloudsDev &src_pts,
                        PlaneCloudsDev &tgt_pts, float length_threshold, float angle_threshold,
                        std::vector<int32_t> &flags_out, const float *cand_rt_dev = nullptr, const PenGather *gather = nullptr);
// cand_rt_dev: the same K x 12 table already on the device (cand_rt_host may then be null: nothing is uploaded)

// ---- oriented bounding boxes (k_obb.hip; ComputeBoundingBox, util.h:186-248) ---------------------------------
// Result block (floats): whole cloud = centre(3), pad, radius as a double (2 floats), pad, valid flag;
// per plane = the four projected corners (12), their centre (3), half diagonal (1).
constexpr uint32_t OBB_OUT_WHOLE = 8, OBB_OUT_PLANE = 16;
struct ObbWork {
    DBuf<float> d_coef, out;
    std::vector<float> host;     // the result block, valid after the sync that follows obb_units
};
// queues the boxes of the whole downsampled cloud and of every per-plane cloud (device arrays whose sizes are still on
// the device: *d_n_ds <= max_ds points, d_plane_off[P] <= max_plane_pts) + the small read-back
void obb_units(plade_ctx *ctx, ObbWork &W, const float *d_ds, const uint32_t *d_n_ds, uint32_t max_ds, const float *d_plane_ds,
               const uint32_t *d_plane_off, uint32_t max_plane_pts, uint32_t P, const float *coef_host, const float *d_coef = nullptr);
// the same for up to 16 clouds in ONE launch (the clouds of a group of pairs): every cloud's result block lands in ITS
// ObbWork::out; obb_adopt_batch(ctx, W, P) notes the read-back on the stream that goes on with that cloud
struct ObbBatchItem { const float *d_ds; const uint32_t *d_n_ds; const float *d_plane_ds; const uint32_t *d_plane_off; uint32_t P;
                      const float *coef_host; ObbWork *work; };
bool obb_units_batch(plade_ctx *ctx, int count, const ObbBatchItem *items, DBuf<float> &coef_scratch);
void obb_adopt_batch(plade_ctx *ctx, ObbWork &W, uint32_t P);

// generic: positions of set flags (ordered); returns count (sync)
uint32_t compact_flags(plade_ctx *ctx, const uint32_t *d_flags, uint32_t n, DBuf<uint32_t> &pos_scratch,
                       DBuf<uint32_t> &out_idx);

}  // namespace plade
