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

}  // namespace plade
