// plade_amd/csrc/score.h -- K1: point-to-plane inlier scoring on gfx950.
#pragma once
#include "ctx.h"

namespace plade {

// Counts for H hypotheses over one SoA cloud.  planes_dev: H x float4 (n, dist) on device.
// assigned may be nullptr.  sub_index (optional): the cloud is a gathered subset and
// assigned must be looked up through sub_index[i].
void score_multi(plade_ctx *ctx, const float *x, const float *y, const float *z, const float *nx,
                 const float *ny, const float *nz, const int32_t *assigned, const uint32_t *sub_index,
                 uint32_t n, const float4 *planes_dev, uint32_t h, float eps, float cos_thresh,
                 uint32_t *counts_dev /* zeroed by callee unless counts_are_zero */, bool counts_are_zero = false);

// Ordered compaction for ONE hypothesis read from device memory (plane_dev[0]):
// idx_out_dev receives ascending point indices, *count_dev the total.  eps_scale multiplies eps
// (3 for Schnabel's global scoring).  masks/blockcounts are scratch owned by the ctx.
struct CompactScratch {
    DBuf<uint8_t> masks;
    DBuf<uint32_t> block_counts;
};
void score_compact(plade_ctx *ctx, CompactScratch &s, const float *x, const float *y, const float *z,
                   const float *nx, const float *ny, const float *nz, const int32_t *assigned, uint32_t n,
                   const float4 *plane_dev, float eps, float cos_thresh, uint32_t *idx_out_dev,
                   uint32_t *count_dev, const uint32_t *skip_flag = nullptr);

// Generic ordered compaction of a precomputed mask array (1 bit per point in groups of 4:
// masks[i/4] bit (i%4)); used by the connected-component filter too.
void compact_masks(plade_ctx *ctx, CompactScratch &s, uint32_t n, const uint32_t *values_or_null,
                   uint32_t *idx_out_dev, uint32_t *count_dev, const uint32_t *skip_flag = nullptr);
// skip_flag (device, nullable): when *skip_flag != 0 the kernels return immediately and leave their
// outputs untouched (used by the refit chain once it has converged).

// Batched forms (one launch for up to 16 hypotheses of up to two clouds; each cloud is read once).  The job
// tables are host arrays copied into the kernel arguments; masks hold cdiv(n,1024)*256 bytes, block_counts
// cdiv(n,1024) words per job.
struct MarkJob {
    const float4 *plane;
    uint8_t *masks;
    uint32_t *block_counts;
    const uint32_t *skip;      // nullable
};
struct CompactJob {
    const uint8_t *masks;
    const uint32_t *block_counts;
    const uint32_t *values;    // nullable: emit the point index itself
    uint32_t *out;
    uint32_t *total;
    const uint32_t *skip;      // nullable
    // optional fused plane parametrisation of the emitted points (PlanePrimitiveShape::Parameters,
    // ransac/PlanePrimitiveShape.h:97-109): uv[j] of the j-th emitted index and the (u, v) bounding box
    const float *frame;        // nullable: pos(3), unused(1), axis0(3), axis1(3)
    float2 *uv;
    float *bbox_part;          // per tile: min u, min v, max u, max v of its emitted points (tiles with a zero
                               // block count are left untouched): reduced by the consumer, no atomics
    uint32_t nb;               // tiles of the cloud the masks cover (filled by compact_batch / the caller)
    const float *px, *py, *pz; // the cloud the indices refer to (only read by jobs with a frame)
};
// One cloud of a batched scoring pass: a launch may serve the jobs of two clouds (the two scans of a pair are
// extracted in lock-step), workgroups [tile0, tile0 + tiles) scan this cloud for jobs [job0, job0 + nj).
struct ScanGroup {
    const float *x, *y, *z, *nx, *ny, *nz;
    const int32_t *assigned;
    uint32_t n, tile0, job0, nj;
    float eps, cos_t;
};
// Job tables travel BY VALUE in the kernel arguments (scalar loads from the kernarg segment): a table in device
// memory costs every workgroup a dependent global load (~1 us) before it can fetch what the entries point to.
constexpr int BATCH_MAXJ = 16;     // 8 per cloud
struct MarkJobs { MarkJob j[BATCH_MAXJ]; ScanGroup g[2]; uint32_t ng; };
struct CompactJobs { CompactJob j[BATCH_MAXJ]; };
// groups: 1 or 2 clouds; tile0 / job0 are filled here (jobs_host holds group 0's jobs, then group 1's)
void score_mark_batch(plade_ctx *ctx, hipStream_t stream, const MarkJob *jobs_host, ScanGroup *groups, uint32_t ng);
// every job carries its cloud (nb, px, py, pz)
void compact_batch(plade_ctx *ctx, hipStream_t stream, const CompactJob *jobs_host, uint32_t nj);

}  // namespace plade
