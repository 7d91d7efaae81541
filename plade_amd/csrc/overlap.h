// plade_amd/csrc/overlap.h -- K8 overlap counting + the uniform target grid it probes.
#pragma once
#include "ctx.h"

namespace plade {

struct TargetGrid {
    struct { float mnx, mny, mnz, inv; int dx, dy, dz; } gp;
    uint32_t n = 0;
    size_t ncells = 0;
    DBuf<float> bbox;
    DBuf<uint32_t> keys, keys2, vals, vals2, cell_start, cell_end;
    DBuf<float4> sorted;  // target points in cell order: (x, y, z, bitcast original index)
    // compact occupancy index (build(..., compact = true)): surfaces fill a few percent of the cells, so the
    // dense start/end tables (tens of MB, L2 misses on every probe) are replaced by one bit per cell, a
    // rank per 64-cell word and the start offsets of the occupied cells only.  Cells are numbered in blocks
    // of 4 x 4 x 4: cell id = block id * 64 + (x&3 | (y&3)<<2 | (z&3)<<4), so one 64-bit word is one block,
    // a 27-cell neighbourhood touches at most 8 words, and the points of a block are contiguous
    DBuf<unsigned long long> occ_bits;   // ncells / 64 words
    DBuf<unsigned long long> occ_blk;    // one bit per block (word of occ_bits): non-empty; staged in LDS by the verification kernel
    DBuf<uint32_t> occ_pop, occ_rank;    // per word: popcount, exclusive prefix (+ total)
    DBuf<uint32_t> occ_start;            // per occupied cell (+ 1): first sorted position
    DBuf<float4> cell_first;             // per occupied cell: its first point, w = that position | 1 << 31 when it is the cell's only point
    bool compact = false;
    // (r5) dense ROW index, the default form of the compact request: target points sorted by their linear cell id in a grid padded by
    // two cells on every side (x fastest), and row_start[L] = number of points in cells < L.  The 27-cell neighbourhood of a probe is
    // nine rows of three consecutive cells = nine contiguous runs [row_start[a], row_start[a + 3]) of `sorted`: no bit masks, no ranks,
    // no per-cell look-ups (the verification kernel was bound by its ~1 900 vector instructions per wavefront and candidate, not by
    // memory).  near_mask: one bit per block of (1 << mask_shift)^3 padded cells, set when an occupied cell lies within one cell of
    // the block -- a probe whose own block is clear ends without a global load.
    bool dense = false;
    int DX = 0, DY = 0, DZ = 0, mask_shift = 2;
    uint32_t mask_words = 0;
    DBuf<uint32_t> row_start, near_mask;
    DBuf<uint32_t> row_occ;       // (r6) bit a: the run of three cells starting at cell a is not empty (k_row_occ)
    bool row_occ_on = false;
    // d_xyz: device pointer, `stride` floats between points; min_cell = largest probe radius used.
    void build(plade_ctx *ctx, const float *d_xyz, uint32_t n_pts, uint32_t stride, float min_cell,
               const float *bbox_min = nullptr, const float *bbox_max = nullptr,   // known bbox skips a device round trip
               bool compact_index = false);
};

// counts[k] (device, int32) and any[k] (device, 1 when the coarse sphere is non-empty)
// The source points are visited in a spatially blocked order (sorted copy made inside, `work`) so that the
// lanes of a wavefront probe neighbouring cells.
struct OverlapWork {
    DBuf<uint32_t> keys, keys2, vals, vals2;
    DBuf<float> sorted;   // 3 x n_s SoA in blocked order
    DBuf<int> bbox;
};
// blocked copy of the source (x | y | z in work.sorted); cell = the target grid's cell edge
void overlap_sort_source(plade_ctx *ctx, OverlapWork &work, const float *d_sx, const float *d_sy, const float *d_sz, uint32_t n_s,
                         float cell);
// d_sx/d_sy/d_sz: the source in any order -- pass work.sorted's planes when overlap_sort_source ran before
void overlap_counts(plade_ctx *ctx, OverlapWork &work, const float *d_sx, const float *d_sy, const float *d_sz, uint32_t n_s,
                    const TargetGrid &grid, const float *d_T, const float *d_centers, uint32_t K, float src_radius,
                    float inlier_dist, int32_t *d_counts, uint32_t *d_any, bool counts_are_zero = false);
// counts_are_zero: d_counts / d_any arrived zeroed with the caller's upload (no memset commands)

void deinterleave3(plade_ctx *ctx, const float *d_xyz, uint32_t n, float *d_x, float *d_y, float *d_z);

}  // namespace plade
