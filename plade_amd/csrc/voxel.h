// plade_amd/csrc/voxel.h -- K9 voxel-grid downsampling and average spacing.
#pragma once
#include "ctx.h"

namespace plade {

struct VoxelWork {
    uint32_t n_out = 0;
    DBuf<uint64_t> keys, keys2;
    DBuf<uint32_t> vals, vals2, heads, seg_group, group_offsets, count;
    DBuf<float> sorted_xyz;   // the items' coordinates in voxel order (x | y | z)
    DBuf<float> out_xyz;  // n_out x 3, ordered by (group, k, j, i)
    // items: item i refers to point item_point[i] (or i when null) of the strided xyz array and
    // belongs to group item_group[i] (or 0).  Items of a voxel are summed in ascending item order.
    uint32_t run(plade_ctx *ctx, const float *d_xyz, uint32_t stride, const uint32_t *d_item_point,
                 const uint32_t *d_item_group, uint32_t n_items, uint32_t n_groups, float leaf,
                 const float bbox_min[3], const float bbox_max[3]);
    // The same in two steps: enqueue() queues keys, sort, runs and centroids (8 launches, no host round trip; the
    // optional d_soa_* arrays are the cloud's SoA copy for the whole-cloud call), finish() waits for the stream and
    // returns the number of voxels.  Several runs can be queued before the first finish().
    void enqueue(plade_ctx *ctx, const float *d_xyz, uint32_t stride, const float *d_soa_x, const float *d_soa_y,
                 const float *d_soa_z, const uint32_t *d_item_point, const uint32_t *d_item_group, uint32_t n_items,
                 uint32_t n_groups, float leaf, const float bbox_min[3], const float bbox_max[3], bool soa_indexed = false,
                 bool groups_are_offsets = false, float *d_out_soa = nullptr);
    // d_out_soa (>= 3 * n_items floats): the centroids also as planes x | y | z with pitch = number of voxels
    // groups_are_offsets: d_item_group holds n_groups + 1 ascending item offsets (the groups are contiguous item ranges)
    // instead of one group id per item
    // soa_indexed: the items are POSITIONS in the SoA planes d_soa_* (e.g. the extraction's Morton-ordered copy, whose
    // plane lists are ascending positions: the gathers then walk memory almost in order) instead of rows of d_xyz
    uint32_t finish(plade_ctx *ctx);
    // The grid has been queued by voxel_whole_batch (below) on another stream that `ctx`'s stream already waits for: note the
    // read-back of its size here; finish(ctx) delivers it.
    void adopt_batch(plade_ctx *ctx);
    uint32_t n_pending = 0, n_pending_host = 0;
};

// The whole-cloud grids (n_groups = 1, items = all points in input order) of up to 16 clouds in ONE launch sequence: keys of all
// clouds, one segmented sort, runs and centroids of all clouds -- 7 launches instead of 7 per cloud (the clouds of a group of
// pairs, registration.hip).  Every cloud's result lands in ITS VoxelWork (out_xyz, count, group_offsets) and out_soa exactly as
// VoxelWork::enqueue would leave it: same keys, same stable order inside a voxel, same sums.
struct VoxBatchItem {
    const float *aos, *sx, *sy, *sz;   // the cloud: N x 6 rows and its SoA planes (item lists: the SoA planes the items index; aos unused)
    const uint32_t *items = nullptr;   // device: item i = position items[i] of sx / sy / sz; nullptr: the whole cloud, item i = point i
    const int32_t *offsets_host = nullptr;   // with items: P + 1 ascending item offsets, group g = items [off[g], off[g + 1]) (the planes)
    uint32_t P = 0;
    uint32_t n;                        // points (whole cloud) or items
    float leaf, bbmin[3], bbmax[3];
    VoxelWork *work;
    float *out_soa;                    // >= 3 n floats (whole cloud; may be nullptr for item lists)
};
// (VoxBatchWork, the batch's scratch arrays, lives in ctx.h: every context owns one)
// false (nothing queued): the batch does not fit this path (an empty cloud, keys wider than 31 bits, a grid PCL would refuse)
// -- the caller lets every pair build its own grid, which reports what is wrong where it belongs
bool voxel_whole_batch(plade_ctx *ctx, VoxBatchWork &B, int count, const VoxBatchItem *items);

struct TargetGrid;
// average_spacing (code/PLADE/util.cpp:1619-1648) of a strided device xyz array with known bbox
float average_spacing_dev(plade_ctx *ctx, const float *d_aos, uint32_t stride_f, uint32_t n, const float *bbmin,
                          const float *bbmax, int k, uint32_t samples, TargetGrid &grid);
// min/max of a strided device xyz array, returned on the host
void bbox_host(plade_ctx *ctx, const float *d_xyz, uint32_t n, uint32_t stride, float mn[3], float mx[3]);
// the same in two halves for uploads that run ahead on their own stream (cloud.hip): see k_voxel.hip
void bbox_init_pattern(int init[8]);
void bbox_async(hipStream_t st, const float *d_xyz, uint32_t n, uint32_t stride, int *d_slot, const int *h_init, int *h_out);
void bbox_decode(const int out[8], float mn[3], float mx[3]);

}  // namespace plade
