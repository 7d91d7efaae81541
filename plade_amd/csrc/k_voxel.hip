// plade_amd/csrc/k_voxel.hip -- K9: voxel-grid downsampling and average point spacing.
//
// (1) pcl::VoxelGrid<PointT>::applyFilter as reached through DownSamplePointCloud
//     (code/PLADE/util.h:161-184; pcl-1.8.1/filters/include/pcl/filters/impl/voxel_grid.hpp:214-262,
//     310-345, 416-426; centroid = fp32 sum / fp32 count, accumulators.hpp:65-84).  Output order =
//     ascending voxel index i + j*dx + k*dx*dy, i.e. lexicographic in (k, j, i).  PCL orders the
//     points of one voxel with an UNSTABLE std::sort, so its fp32 centroid sum order is
//     implementation-defined; this kernel sums in ascending input position (the order a stable sort
//     gives) -- the oracle's `sort_mode = 1`; DESIGN.md quantifies the <= few-ulp difference.
//     Several point groups (per-plane clouds, plade.cpp:93-105) are voxelised in one pass by putting
//     the group id in the top bits of the sort key.
// (2) average_spacing (code/PLADE/util.cpp:1619-1648): k = 6 nearest neighbours (FLANN fp32
//     distances) of <= 10000 strided sample points, (sum_{i=1..k-1} sqrt(d_i)) / k averaged in
//     double.  Exact brute-force kNN on the GPU: the k smallest fp32 distances are the same numbers
//     whatever search structure finds them.
#include "voxel.h"
#include "overlap.h"
#include "prims.h"

namespace plade {

// ------------------------------------------------------------------------------------------------
// (a) keys: item i -> packed (group | k | j | i-voxel) key.  The whole-cloud call reads the SoA copy of the cloud
//     (coalesced 4 B/lane streams); item lists (per-plane clouds) gather 12 of the 24 B of an AoS record.
// (K = uint32_t when the packed key fits 32 bits -- the usual case: a 10 m room at a 5 cm leaf takes 22 bits + 5 group bits --
//  so that the sort moves 8 instead of 12 bytes per item and pass; uint64_t otherwise)
template <class K>
__global__ void k_voxel_keys(const float *__restrict__ xyz, uint32_t stride, const float *__restrict__ sx,
                             const float *__restrict__ sy, const float *__restrict__ sz, int soa_indexed,
                             const uint32_t *__restrict__ item_point, const uint32_t *__restrict__ item_group,
                             uint32_t n_items, float inv, int lminx, int lminy, int lminz, int bx, int by, int bz,
                             K *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t n_group_offsets) {
    // n_group_offsets != 0: item_group holds that many ascending item OFFSETS (group g = items [off[g], off[g + 1])) instead
    // of one group id per item; they are staged in LDS and searched (<= 1025 entries)
    __shared__ uint32_t s_off[1026];
    if (n_group_offsets) {
        for (uint32_t q = threadIdx.x; q < n_group_offsets; q += blockDim.x) s_off[q] = item_group[q];
        __syncthreads();
    }
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    float x, y, z;
    if (!item_point && sx) { x = sx[i]; y = sy[i]; z = sz[i]; }
    else if (soa_indexed) { const uint32_t p = item_point[i]; x = sx[p]; y = sy[p]; z = sz[p]; }   // items = positions in an SoA copy
    else {
        const uint32_t p = item_point ? item_point[i] : i;
        x = xyz[(size_t)p * stride]; y = xyz[(size_t)p * stride + 1]; z = xyz[(size_t)p * stride + 2];
    }
    // voxel_grid.hpp:330-332: floor(x * inverse_leaf_size) (fp32), then the integer offset
    const uint64_t lx = (uint64_t)((int)floorf(x * inv) - lminx);
    const uint64_t ly = (uint64_t)((int)floorf(y * inv) - lminy);
    const uint64_t lz = (uint64_t)((int)floorf(z * inv) - lminz);
    uint64_t g = 0u;
    if (n_group_offsets) {
        uint32_t lo = 0, hi = n_group_offsets - 1;   // last g with off[g] <= i
        while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (s_off[mid] <= i) lo = mid; else hi = mid - 1; }
        g = lo;
    } else if (item_group) g = item_group[i];
    keys[i] = (K)((g << (bx + by + bz)) | (lz << (bx + by)) | (ly << bx) | lx);
    vals[i] = i;
}

// (b) after the sort, ONE launch: the runs of equal keys (= voxels) are numbered by a decoupled look-back scan of the
//     run-head flags (heads[s] = first sorted position of voxel s, seg_group[s] = its group, *n_seg = their number), and
//     the items' coordinates are gathered into sorted order (coalesced stores), so that the summation kernel streams
//     contiguous memory.  Tiles of 4096 sorted positions.
constexpr int VR_T = 256, VR_I = 16, VR_TILE = VR_T * VR_I;
constexpr uint64_t VR_AGG = 1ull << 32, VR_PREFIX = 2ull << 32, VR_STATUS = 3ull << 32;
template <class K>
__global__ __launch_bounds__(VR_T) void k_voxel_runs(const K *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                      uint32_t n, const float *__restrict__ xyz, uint32_t stride,
                                                      const float *__restrict__ ix, const float *__restrict__ iy,
                                                      const float *__restrict__ iz /* SoA planes indexed by item_point, or null */,
                                                      const uint32_t *__restrict__ item_point, int group_shift,
                                                      uint64_t *__restrict__ state, uint32_t *__restrict__ ticket, uint32_t base,
                                                      uint32_t gen, uint32_t *__restrict__ heads, uint32_t *__restrict__ seg_group,
                                                      uint32_t *__restrict__ n_seg, float *__restrict__ ox, float *__restrict__ oy,
                                                      float *__restrict__ oz) {
    __shared__ uint32_t s_tile, s_w[VR_T / 64], s_excl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u) - base;
    __syncthreads();
    const uint32_t tile = s_tile;
    // the gather first: its loads are in flight while the scan part runs
    float gx[VR_I], gy[VR_I], gz[VR_I];
#pragma unroll
    for (int q = 0; q < VR_I; ++q) {
        const uint32_t j = tile * VR_TILE + q * VR_T + tid;
        if (j < n) {
            const uint32_t it = vals[j];
            const uint32_t p = item_point ? item_point[it] : it;
            if (ix) { gx[q] = ix[p]; gy[q] = iy[p]; gz[q] = iz[p]; }
            else {
                const float *r = xyz + (size_t)p * stride;
                gx[q] = r[0]; gy[q] = r[1]; gz[q] = r[2];
            }
        }
    }
    const uint32_t first = tile * VR_TILE + tid * VR_I;
    K k[VR_I + 1];
    k[0] = (first > 0 && first <= n) ? keys[first - 1] : (K)~(K)0;
#pragma unroll
    for (int q = 0; q < VR_I; ++q) k[q + 1] = first + q < n ? keys[first + q] : (K)~(K)0;
    uint32_t fl = 0, sum = 0;
#pragma unroll
    for (int q = 0; q < VR_I; ++q) {
        const bool head = first + q < n && (first + q == 0 || k[q + 1] != k[q]);
        fl |= (head ? 1u : 0u) << q;
        sum += head;
    }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t agg = 0, woff = 0;
    for (int w = 0; w < VR_T / 64; ++w) { if (w < wave) woff += s_w[w]; agg += s_w[w]; }
    const uint64_t tag = (uint64_t)gen << 34;
    if (wave == 0) {
        if (lane == 0)
            __hip_atomic_store(state + tile, tag | (tile == 0 ? VR_PREFIX : VR_AGG) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        if (tile > 0) {
            int t = (int)tile - 1;
            for (;;) {
                const int idx = t - lane;
                uint64_t st = tag | VR_PREFIX;
                if (idx >= 0) st = __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ready = (st >> 34) == gen && (st & VR_STATUS) != 0ull;
                const unsigned long long not_ready = __ballot(!ready), is_prefix = __ballot(ready && (st & VR_PREFIX));
                const int first_bad = not_ready ? __ffsll((long long)not_ready) - 1 : 64;
                const int first_pre = is_prefix ? __ffsll((long long)is_prefix) - 1 : 64;
                const int take = first_pre < first_bad ? first_pre + 1 : first_bad;
                uint32_t part = lane < take ? (uint32_t)st : 0u;
                for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
                excl += part;
                if (first_pre < first_bad) break;
                t -= take;
            }
            if (lane == 0)
                __hip_atomic_store(state + tile, tag | VR_PREFIX | (uint64_t)(excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_excl = excl;
    }
    __syncthreads();
    uint32_t rank = s_excl + woff + incl - sum;
#pragma unroll
    for (int q = 0; q < VR_I; ++q)
        if (fl & (1u << q)) {
            heads[rank] = first + q;
            seg_group[rank] = (uint32_t)(k[q + 1] >> group_shift);
            ++rank;
        }
    if ((uint64_t)tile * VR_TILE + VR_TILE >= n && tid == VR_T - 1) *n_seg = s_excl + agg;   // the last tile
#pragma unroll
    for (int q = 0; q < VR_I; ++q) {
        const uint32_t j = tile * VR_TILE + q * VR_T + tid;
        if (j < n) { ox[j] = gx[q]; oy[j] = gy[q]; oz[j] = gz[q]; }
    }
}

// (c) centroids: one lane per voxel adds its points in ascending sorted position (= ascending item order inside the
//     voxel: the sort is stable) -- the fp32 sum order of the oracle's sort_mode 1 -- from contiguous memory;
//     centroid = fp32 sum / fp32 count (accumulators.hpp:65-84).  Workgroup 0 also derives the groups' offsets
//     (first voxel of every group, empty groups included) by binary search in the voxels' group ids.
__global__ __launch_bounds__(128) void k_voxel_centroids(const float *__restrict__ sx, const float *__restrict__ sy,
                                                         const float *__restrict__ sz, const uint32_t *__restrict__ heads,
                                                         const uint32_t *__restrict__ seg_group, const uint32_t *__restrict__ n_seg_p,
                                                         uint32_t n_items, uint32_t n_groups, float *__restrict__ out_xyz,
                                                         uint32_t *__restrict__ group_offsets, float *__restrict__ out_soa) {
    const uint32_t n_seg = *n_seg_p;
    if (blockIdx.x == 0)
        for (uint32_t g = threadIdx.x; g <= n_groups; g += blockDim.x) {
            uint32_t lo = 0, hi = n_seg;   // first voxel whose group is >= g
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (seg_group[mid] < g) lo = mid + 1; else hi = mid; }
            group_offsets[g] = lo;
        }
    // The 128 voxels of a workgroup own one contiguous span of the sorted points.  A lane that walked its run in global memory
    // touched a different line than its neighbours with every load (64 lines per wavefront load for 64 x 4 useful bytes); the
    // span is staged through LDS in chunks instead -- coalesced loads -- and every lane adds, in ascending position as before,
    // the part of its run that lies in the chunk.
    constexpr uint32_t CH = 2048;
    __shared__ float s_x[CH], s_y[CH], s_z[CH];
    __shared__ uint32_t s_span[2];
    const uint32_t s0 = blockIdx.x * blockDim.x;
    if (s0 >= n_seg) return;
    const uint32_t s = s0 + threadIdx.x;
    const bool live = s < n_seg;
    const uint32_t b = live ? heads[s] : 0u, e = live ? ((s + 1 < n_seg) ? heads[s + 1] : n_items) : 0u;
    if (threadIdx.x == 0) s_span[0] = b;
    if (s == min(n_seg, s0 + (uint32_t)blockDim.x) - 1) s_span[1] = e;
    __syncthreads();
    const uint32_t span_b = s_span[0], span_e = s_span[1];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (uint32_t c0 = span_b; c0 < span_e; c0 += CH) {
        const uint32_t c1 = min(span_e, c0 + CH);
        for (uint32_t j = c0 + threadIdx.x; j < c1; j += blockDim.x) { s_x[j - c0] = sx[j]; s_y[j - c0] = sy[j]; s_z[j - c0] = sz[j]; }
        __syncthreads();
        const uint32_t jb = max(b, c0), je = min(e, c1);
        for (uint32_t j = jb; j < je; ++j) { ax += s_x[j - c0]; ay += s_y[j - c0]; az += s_z[j - c0]; }
        __syncthreads();
    }
    if (!live) return;
    const float cnt = (float)(e - b);
    const float cx = ax / cnt, cy = ay / cnt, cz = az / cnt;
    out_xyz[3 * (size_t)s] = cx;
    out_xyz[3 * (size_t)s + 1] = cy;
    out_xyz[3 * (size_t)s + 2] = cz;
    if (out_soa) { out_soa[s] = cx; out_soa[(size_t)n_seg + s] = cy; out_soa[2 * (size_t)n_seg + s] = cz; }   // x | y | z, pitch = voxel count
}

void VoxelWork::enqueue(plade_ctx *ctx, const float *d_xyz, uint32_t stride, const float *d_soa_x, const float *d_soa_y,
                        const float *d_soa_z, const uint32_t *d_item_point, const uint32_t *d_item_group, uint32_t n_items,
                        uint32_t n_groups, float leaf, const float bbox_min[3], const float bbox_max[3], bool soa_indexed,
                        bool groups_are_offsets, float *d_out_soa) {
    PLADE_REQUIRE(!soa_indexed || (d_item_point && d_soa_x), PLADE_EINVAL, "voxel: an indexed SoA source needs items and planes");
    n_out = 0;
    n_pending = 0;
    PLADE_REQUIRE(leaf > 0.f, PLADE_EINVAL, "voxel: leaf must be positive");
    PLADE_REQUIRE(n_groups >= 1 && n_groups <= 1024, PLADE_ELIMIT, "voxel: at most 1024 point groups");
    if (n_items == 0) {   // no items: zero voxels, every group empty (the consumers still read the device arrays)
        group_offsets.ensure((size_t)n_groups + 2);
        count.ensure(4);
        out_xyz.ensure(4);
        ctx->fill_async(group_offsets.p, 0, ((size_t)n_groups + 2) * 4);
        ctx->fill_async(count.p, 0, 4);
        return;
    }
    const float inv = 1.f / leaf;
    const int lmin[3] = {(int)floorf(bbox_min[0] * inv), (int)floorf(bbox_min[1] * inv), (int)floorf(bbox_min[2] * inv)};
    const int lmax[3] = {(int)floorf(bbox_max[0] * inv), (int)floorf(bbox_max[1] * inv), (int)floorf(bbox_max[2] * inv)};
    for (int k = 0; k < 3; ++k)
        PLADE_REQUIRE((int64_t)lmax[k] - lmin[k] < (1 << 18), PLADE_ELIMIT,
                      "voxel: more than 2^18 leaves along one axis (PCL would refuse this leaf size too)");
    // voxel_grid.hpp:236-245: dx*dy*dz must fit int32 (per cloud; checked on the union bbox, which is
    // the tighter of the two only for the whole-cloud call -- per-plane clouds are subsets)
    {
        int64_t dx = (int64_t)((bbox_max[0] - bbox_min[0]) * inv) + 1, dy = (int64_t)((bbox_max[1] - bbox_min[1]) * inv) + 1,
                dz = (int64_t)((bbox_max[2] - bbox_min[2]) * inv) + 1;
        PLADE_REQUIRE(dx * dy * dz <= (int64_t)INT32_MAX, PLADE_ELIMIT, "voxel: leaf size too small for the data extent");
    }
    // key = (group | k | j | i) packed with just enough bits per field: the radix sort cost is
    // proportional to the key width
    auto bits_for = [](int64_t range) { int b = 1; while (((int64_t)1 << b) <= range) ++b; return b; };
    const int bx = bits_for((int64_t)lmax[0] - lmin[0]), by = bits_for((int64_t)lmax[1] - lmin[1]), bz = bits_for((int64_t)lmax[2] - lmin[2]);
    keys.ensure(n_items); keys2.ensure(n_items); vals.ensure(n_items); vals2.ensure(n_items);
    sorted_xyz.ensure(3 * (size_t)n_items + 4);
    heads.ensure((size_t)n_items + 1);
    seg_group.ensure((size_t)n_items + 1);
    out_xyz.ensure((size_t)n_items * 3 + 4);     // upper bound: one voxel per item
    group_offsets.ensure((size_t)n_groups + 2);
    count.ensure(4);
    const unsigned nb = cdiv(n_items, 256);
    int gbits = 0;
    while ((1u << gbits) < n_groups) ++gbits;
    const int key_bits = bx + by + bz + gbits;
    const bool narrow = key_bits <= 31;   // (31: the all-ones padding key of k_voxel_runs must not be a real key)
    uint32_t *k32 = reinterpret_cast<uint32_t *>(keys.p), *k32b = reinterpret_cast<uint32_t *>(keys2.p);
    if (narrow)
        launch_raw(ctx, k_voxel_keys<uint32_t>, dim3(nb), dim3(256), 0, d_xyz, stride, d_soa_x, d_soa_y, d_soa_z, soa_indexed ? 1 : 0,
                           d_item_point, d_item_group, n_items, inv, lmin[0], lmin[1], lmin[2], bx, by, bz, k32, vals.p,
                           groups_are_offsets ? n_groups + 1 : 0u);
    else
        launch_raw(ctx, k_voxel_keys<uint64_t>, dim3(nb), dim3(256), 0, d_xyz, stride, d_soa_x, d_soa_y, d_soa_z, soa_indexed ? 1 : 0,
                           d_item_point, d_item_group, n_items, inv, lmin[0], lmin[1], lmin[2], bx, by, bz, keys.p, vals.p,
                           groups_are_offsets ? n_groups + 1 : 0u);
    if (narrow) sort_pairs_u32(ctx, k32, k32b, vals.p, vals2.p, n_items, key_bits);
    else sort_pairs_u64(ctx, keys.p, keys2.p, vals.p, vals2.p, n_items, key_bits);
    const ScanTicket t = scan_ticket(ctx, n_items, VR_TILE);
    float *ox = sorted_xyz.p, *oy = ox + n_items, *oz = oy + n_items;
    if (narrow)
        launch_raw(ctx, k_voxel_runs<uint32_t>, dim3(t.tiles), dim3(VR_T), 0, k32b, vals2.p, n_items, d_xyz, stride,
                           soa_indexed ? d_soa_x : nullptr, soa_indexed ? d_soa_y : nullptr, soa_indexed ? d_soa_z : nullptr, d_item_point,
                           bx + by + bz, t.state, t.ticket, t.base, t.gen, heads.p, seg_group.p, count.p, ox, oy, oz);
    else
        launch_raw(ctx, k_voxel_runs<uint64_t>, dim3(t.tiles), dim3(VR_T), 0, keys2.p, vals2.p, n_items, d_xyz, stride,
                           soa_indexed ? d_soa_x : nullptr, soa_indexed ? d_soa_y : nullptr, soa_indexed ? d_soa_z : nullptr, d_item_point,
                           bx + by + bz, t.state, t.ticket, t.base, t.gen, heads.p, seg_group.p, count.p, ox, oy, oz);
    launch_raw(ctx, k_voxel_centroids, dim3(cdiv(n_items, 128)), dim3(128), 0, ox, oy, oz, heads.p, seg_group.p,
                       count.p, n_items, n_groups, out_xyz.p, group_offsets.p, d_out_soa);
    HIP_TRY(hipGetLastError());
    ctx->d2h(&n_pending_host, count.p, 4);   // valid after the next sync of this stream
    n_pending = 1;
}

void VoxelWork::adopt_batch(plade_ctx *ctx) {
    n_out = 0;
    ctx->d2h(&n_pending_host, count.p, 4);   // valid after the next sync of this stream
    n_pending = 1;
}

// ---- the whole-cloud grids of several clouds in one launch sequence (voxel.h) ------------------------------------------------
constexpr int VB_MAX = 16;
struct VBCloud {
    const float *sx, *sy, *sz, *aos;   // whole cloud: its SoA planes (keys) and N x 6 rows (gather); item lists: the SoA planes the items index
    const uint32_t *items;             // item i = position items[i] in sx / sy / sz (nullptr: item i = point i)
    const uint32_t *offs;              // P + 1 ascending item offsets: group g = items [offs[g], offs[g + 1]) (nullptr: one group)
    uint32_t n, P;
    float inv;
    int lminx, lminy, lminz, bx, by, gshift;   // gshift = bx + by + bz: the group id sits above the voxel index
    float *out_xyz, *out_soa;          // out_soa may be nullptr
    uint32_t *count, *group_offsets;
};
struct VBArgs {
    VBCloud c[VB_MAX];
    uint32_t ncl;
    uint32_t start[VB_MAX + 1];        // first item of cloud g in the concatenated arrays
    uint32_t tile_start[VB_MAX + 1];   // first tile of VR_TILE sorted positions
    uint32_t cblk_start[VB_MAX + 1];   // first workgroup of the centroid kernel
    uint32_t total;
};
__device__ __forceinline__ uint32_t vb_find(const uint32_t *start, uint32_t ncl, uint32_t v) {
    uint32_t g = 0;
#pragma unroll
    for (int q = 1; q < VB_MAX; ++q) g += (q < (int)ncl && v >= start[q]) ? 1u : 0u;
    return g;
}

__global__ void k_vb_keys(const VBArgs A, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= A.total) return;
    const uint32_t g = vb_find(A.start, A.ncl, t);
    const VBCloud &C = A.c[g];
    const uint32_t i = t - A.start[g];
    const uint32_t p = C.items ? C.items[i] : i;
    const float x = C.sx[p], y = C.sy[p], z = C.sz[p];
    // voxel_grid.hpp:330-332, as k_voxel_keys
    const uint32_t lx = (uint32_t)((int)floorf(x * C.inv) - C.lminx);
    const uint32_t ly = (uint32_t)((int)floorf(y * C.inv) - C.lminy);
    const uint32_t lz = (uint32_t)((int)floorf(z * C.inv) - C.lminz);
    uint32_t grp = 0;
    if (C.offs) {   // last group whose first item is <= i
        uint32_t lo = 0, hi = C.P;
        while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (C.offs[mid] <= i) lo = mid; else hi = mid - 1; }
        grp = lo;
    }
    keys[t] = (grp << C.gshift) | (lz << (C.bx + C.by)) | (ly << C.bx) | lx;
    vals[t] = i;
}

// k_voxel_runs for the concatenated clouds: a tile belongs to one cloud, the look-back stops at the cloud's first tile
__global__ __launch_bounds__(VR_T) void k_vb_runs(const VBArgs A, const uint32_t *__restrict__ keys_all, const uint32_t *__restrict__ vals_all,
                                                  uint64_t *__restrict__ state_all, uint32_t *__restrict__ ticket, uint32_t base, uint32_t gen,
                                                  uint32_t *__restrict__ heads_all, uint32_t *__restrict__ seg_group_all,
                                                  float *__restrict__ sorted_all) {
    __shared__ uint32_t s_tile, s_w[VR_T / 64], s_excl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u) - base;
    __syncthreads();
    const uint32_t gtile = s_tile;
    const uint32_t g = vb_find(A.tile_start, A.ncl, gtile);
    const VBCloud &C = A.c[g];
    const uint32_t tile = gtile - A.tile_start[g], n = C.n;
    const uint32_t *__restrict__ keys = keys_all + A.start[g];
    const uint32_t *__restrict__ vals = vals_all + A.start[g];
    uint64_t *__restrict__ state = state_all + A.tile_start[g];
    uint32_t *__restrict__ heads = heads_all + A.start[g] + g;   // n + 1 slots per cloud
    uint32_t *__restrict__ seg_group = seg_group_all + A.start[g] + g;
    float *__restrict__ ox = sorted_all + A.start[g], *__restrict__ oy = ox + A.total, *__restrict__ oz = oy + A.total;
    float gx[VR_I], gy[VR_I], gz[VR_I];
#pragma unroll
    for (int q = 0; q < VR_I; ++q) {
        const uint32_t j = tile * VR_TILE + q * VR_T + tid;
        if (j < n) {
            if (C.items) { const uint32_t p = C.items[vals[j]]; gx[q] = C.sx[p]; gy[q] = C.sy[p]; gz[q] = C.sz[p]; }
            else { const float *r = C.aos + (size_t)vals[j] * 6; gx[q] = r[0]; gy[q] = r[1]; gz[q] = r[2]; }
        }
    }
    const uint32_t first = tile * VR_TILE + tid * VR_I;
    uint32_t k[VR_I + 1];
    k[0] = (first > 0 && first <= n) ? keys[first - 1] : 0xffffffffu;
#pragma unroll
    for (int q = 0; q < VR_I; ++q) k[q + 1] = first + q < n ? keys[first + q] : 0xffffffffu;
    uint32_t fl = 0, sum = 0;
#pragma unroll
    for (int q = 0; q < VR_I; ++q) {
        const bool head = first + q < n && (first + q == 0 || k[q + 1] != k[q]);
        fl |= (head ? 1u : 0u) << q;
        sum += head;
    }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t agg = 0, woff = 0;
    for (int w = 0; w < VR_T / 64; ++w) { if (w < wave) woff += s_w[w]; agg += s_w[w]; }
    const uint64_t tag = (uint64_t)gen << 34;
    if (wave == 0) {
        if (lane == 0)
            __hip_atomic_store(state + tile, tag | (tile == 0 ? VR_PREFIX : VR_AGG) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        if (tile > 0) {
            int t = (int)tile - 1;
            for (;;) {
                const int idx = t - lane;
                uint64_t st = tag | VR_PREFIX;
                if (idx >= 0) st = __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ready = (st >> 34) == gen && (st & VR_STATUS) != 0ull;
                const unsigned long long not_ready = __ballot(!ready), is_prefix = __ballot(ready && (st & VR_PREFIX));
                const int first_bad = not_ready ? __ffsll((long long)not_ready) - 1 : 64;
                const int first_pre = is_prefix ? __ffsll((long long)is_prefix) - 1 : 64;
                const int take = first_pre < first_bad ? first_pre + 1 : first_bad;
                uint32_t part = lane < take ? (uint32_t)st : 0u;
                for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
                excl += part;
                if (first_pre < first_bad) break;
                t -= take;
            }
            if (lane == 0)
                __hip_atomic_store(state + tile, tag | VR_PREFIX | (uint64_t)(excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_excl = excl;
    }
    __syncthreads();
    uint32_t rank = s_excl + woff + incl - sum;
#pragma unroll
    for (int q = 0; q < VR_I; ++q)
        if (fl & (1u << q)) { heads[rank] = first + q; seg_group[rank] = k[q + 1] >> C.gshift; ++rank; }
    if ((uint64_t)tile * VR_TILE + VR_TILE >= n && tid == VR_T - 1) *C.count = s_excl + agg;   // the cloud's last tile
#pragma unroll
    for (int q = 0; q < VR_I; ++q) {
        const uint32_t j = tile * VR_TILE + q * VR_T + tid;
        if (j < n) { ox[j] = gx[q]; oy[j] = gy[q]; oz[j] = gz[q]; }
    }
}

// k_voxel_centroids for the concatenated clouds (one group per cloud: group_offsets = {0, voxels})
__global__ __launch_bounds__(128) void k_vb_centroids(const VBArgs A, const uint32_t *__restrict__ heads_all,
                                                      const uint32_t *__restrict__ seg_group_all, const float *__restrict__ sorted_all) {
    constexpr uint32_t CH = 2048;
    __shared__ float s_x[CH], s_y[CH], s_z[CH];
    __shared__ uint32_t s_span[2];
    const uint32_t g = vb_find(A.cblk_start, A.ncl, blockIdx.x);
    const VBCloud &C = A.c[g];
    const uint32_t blk = blockIdx.x - A.cblk_start[g];
    const uint32_t n_seg = *C.count, n_items = C.n;
    const uint32_t *__restrict__ heads = heads_all + A.start[g] + g;
    if (blk == 0) {   // first voxel of every group, empty groups included (as k_voxel_centroids)
        const uint32_t *__restrict__ seg_group = seg_group_all + A.start[g] + g;
        for (uint32_t q = threadIdx.x; q <= C.P; q += blockDim.x) {
            uint32_t lo = 0, hi = n_seg;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (seg_group[mid] < q) lo = mid + 1; else hi = mid; }
            C.group_offsets[q] = lo;
        }
    }
    const float *__restrict__ sx = sorted_all + A.start[g], *__restrict__ sy = sx + A.total, *__restrict__ sz = sy + A.total;
    const uint32_t s0 = blk * blockDim.x;
    if (s0 >= n_seg) return;
    const uint32_t s = s0 + threadIdx.x;
    const bool live = s < n_seg;
    const uint32_t b = live ? heads[s] : 0u, e = live ? ((s + 1 < n_seg) ? heads[s + 1] : n_items) : 0u;
    if (threadIdx.x == 0) s_span[0] = b;
    if (s == min(n_seg, s0 + (uint32_t)blockDim.x) - 1) s_span[1] = e;
    __syncthreads();
    const uint32_t span_b = s_span[0], span_e = s_span[1];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (uint32_t c0 = span_b; c0 < span_e; c0 += CH) {
        const uint32_t c1 = min(span_e, c0 + CH);
        for (uint32_t j = c0 + threadIdx.x; j < c1; j += blockDim.x) { s_x[j - c0] = sx[j]; s_y[j - c0] = sy[j]; s_z[j - c0] = sz[j]; }
        __syncthreads();
        const uint32_t jb = max(b, c0), je = min(e, c1);
        for (uint32_t j = jb; j < je; ++j) { ax += s_x[j - c0]; ay += s_y[j - c0]; az += s_z[j - c0]; }
        __syncthreads();
    }
    if (!live) return;
    const float cnt = (float)(e - b);
    const float cx = ax / cnt, cy = ay / cnt, cz = az / cnt;
    C.out_xyz[3 * (size_t)s] = cx; C.out_xyz[3 * (size_t)s + 1] = cy; C.out_xyz[3 * (size_t)s + 2] = cz;
    if (C.out_soa) { C.out_soa[s] = cx; C.out_soa[(size_t)n_seg + s] = cy; C.out_soa[2 * (size_t)n_seg + s] = cz; }
}

bool voxel_whole_batch(plade_ctx *ctx, VoxBatchWork &B, int count, const VoxBatchItem *items) {
    if (count < 1 || count > VB_MAX) return false;
    VBArgs A;
    memset(&A, 0, sizeof(A));
    A.ncl = (uint32_t)count;
    auto bits_for = [](int64_t range) { int b = 1; while (((int64_t)1 << b) <= range) ++b; return b; };
    int max_bits = 1;
    uint64_t total = 0;
    size_t n_offs = 0;
    for (int g = 0; g < count; ++g) {
        const VoxBatchItem &it = items[g];
        if (it.n == 0 || !(it.leaf > 0.f) || !it.work || (!it.items && !it.out_soa)) return false;
        if (it.items && (!it.offsets_host || it.P < 1 || it.P > 1024)) return false;
        const float inv = 1.f / it.leaf;
        const int lmin[3] = {(int)floorf(it.bbmin[0] * inv), (int)floorf(it.bbmin[1] * inv), (int)floorf(it.bbmin[2] * inv)};
        const int lmax[3] = {(int)floorf(it.bbmax[0] * inv), (int)floorf(it.bbmax[1] * inv), (int)floorf(it.bbmax[2] * inv)};
        for (int k = 0; k < 3; ++k)
            if ((int64_t)lmax[k] - lmin[k] >= (1 << 18)) return false;
        const int64_t dx = (int64_t)((it.bbmax[0] - it.bbmin[0]) * inv) + 1, dy = (int64_t)((it.bbmax[1] - it.bbmin[1]) * inv) + 1,
                      dz = (int64_t)((it.bbmax[2] - it.bbmin[2]) * inv) + 1;
        if (dx * dy * dz > (int64_t)INT32_MAX) return false;
        const int bx = bits_for((int64_t)lmax[0] - lmin[0]), by = bits_for((int64_t)lmax[1] - lmin[1]), bz = bits_for((int64_t)lmax[2] - lmin[2]);
        int gbits = 0;
        if (it.items) while ((1u << gbits) < it.P) ++gbits;
        if (bx + by + bz + gbits > 31) return false;
        max_bits = std::max(max_bits, bx + by + bz + gbits);
        n_offs += it.items ? it.P + 1 : 0;
        VBCloud &C = A.c[g];
        C.sx = it.sx; C.sy = it.sy; C.sz = it.sz; C.aos = it.aos; C.n = it.n; C.inv = inv;
        C.items = it.items; C.P = it.items ? it.P : 1u; C.gshift = bx + by + bz;
        C.lminx = lmin[0]; C.lminy = lmin[1]; C.lminz = lmin[2]; C.bx = bx; C.by = by;
        VoxelWork &w = *it.work;
        w.out_xyz.ensure((size_t)it.n * 3 + 4);
        w.group_offsets.ensure((size_t)C.P + 2);
        w.count.ensure(4);
        w.n_out = 0; w.n_pending = 0;
        C.out_xyz = w.out_xyz.p; C.out_soa = it.out_soa; C.count = w.count.p; C.group_offsets = w.group_offsets.p;
        A.start[g + 1] = A.start[g] + it.n;
        A.tile_start[g + 1] = A.tile_start[g] + cdiv(it.n, VR_TILE);
        A.cblk_start[g + 1] = A.cblk_start[g] + cdiv(it.n, 128);
        total += it.n;
    }
    if (total >= (1ull << 30)) return false;
    for (int g = count; g < VB_MAX; ++g) { A.c[g] = A.c[0]; A.start[g + 1] = A.start[g]; A.tile_start[g + 1] = A.tile_start[g]; A.cblk_start[g + 1] = A.cblk_start[g]; }
    A.total = (uint32_t)total;
    B.keys.ensure(total); B.keys2.ensure(total); B.vals.ensure(total); B.vals2.ensure(total);
    B.heads.ensure(total + VB_MAX + 1);
    B.seg_group.ensure(total + VB_MAX + 1);
    B.sorted_xyz.ensure(3 * total + 4);
    hipStream_t st = ctx->stream;
    if (n_offs) {   // the item offsets of all item-list clouds: one upload
        B.offs.ensure(n_offs + 4);
        std::vector<uint32_t> h(n_offs);
        size_t o = 0;
        for (int g = 0; g < count; ++g) {
            if (!items[g].items) continue;
            for (uint32_t q = 0; q <= items[g].P; ++q) h[o + q] = (uint32_t)items[g].offsets_host[q];
            A.c[g].offs = B.offs.p + o;
            o += items[g].P + 1;
        }
        const bool staged = ctx->h2d(B.offs.p, h.data(), 4 * n_offs);
        if (!staged) ctx->sync();
    }
    hipLaunchKernelGGL(k_vb_keys, dim3(cdiv(total, 256)), dim3(256), 0, st, A, B.keys.p, B.vals.p);
    radix_sort_segments_u32(ctx, B.keys.p, B.keys2.p, B.vals.p, B.vals2.p, A.start, count, max_bits);
    const uint32_t tiles = A.tile_start[count];
    const ScanTicket t = scan_ticket(ctx, (size_t)tiles * VR_TILE, VR_TILE);
    hipLaunchKernelGGL(k_vb_runs, dim3(tiles), dim3(VR_T), 0, st, A, B.keys2.p, B.vals2.p, t.state, t.ticket, t.base, t.gen, B.heads.p,
                       B.seg_group.p, B.sorted_xyz.p);
    hipLaunchKernelGGL(k_vb_centroids, dim3(A.cblk_start[count]), dim3(128), 0, st, A, B.heads.p, B.seg_group.p, B.sorted_xyz.p);
    HIP_TRY(hipGetLastError());
    return true;
}

uint32_t VoxelWork::finish(plade_ctx *ctx) {
    if (!n_pending) return n_out;
    ctx->sync();
    n_pending = 0;
    n_out = n_pending_host;
    return n_out;
}

uint32_t VoxelWork::run(plade_ctx *ctx, const float *d_xyz, uint32_t stride, const uint32_t *d_item_point,
                        const uint32_t *d_item_group, uint32_t n_items, uint32_t n_groups, float leaf,
                        const float bbox_min[3], const float bbox_max[3]) {
    enqueue(ctx, d_xyz, stride, nullptr, nullptr, nullptr, d_item_point, d_item_group, n_items, n_groups, leaf, bbox_min, bbox_max);
    return finish(ctx);
}

// ------------------------------------------------------------------------------------------------
// average spacing: exact k nearest neighbours through a uniform grid, one wavefront per query.
// Lanes sweep the cells of the current Chebyshev ring in parallel and keep private ascending top-K
// lists; after each ring the wave extracts its K best by repeated arg-min and stops once the K-th
// distance is closer than anything an outer ring could hold.
constexpr int SP_K = 8;     // max k supported

struct SpGrid { float mnx, mny, mnz, inv, cell; int dx, dy, dz; };

__global__ __launch_bounds__(256) void k_knn_grid(const float4 *__restrict__ pts, const uint32_t *__restrict__ cstart,
                                                  const uint32_t *__restrict__ cend, SpGrid g,
                                                  const float *__restrict__ aos, uint32_t stride_f, uint32_t n,
                                                  uint32_t step, uint32_t nq, int k, double *__restrict__ avg_out,
                                                  uint32_t *__restrict__ nbs_out) {
    const int lane = threadIdx.x & 63;
    const uint32_t qi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const uint32_t pi = min(n - 1, qi * step);
    const f3 q(aos[(size_t)pi * stride_f], aos[(size_t)pi * stride_f + 1], aos[(size_t)pi * stride_f + 2]);
    const int cx = min(max((int)floorf((q.x - g.mnx) * g.inv), 0), g.dx - 1);
    const int cy = min(max((int)floorf((q.y - g.mny) * g.inv), 0), g.dy - 1);
    const int cz = min(max((int)floorf((q.z - g.mnz) * g.inv), 0), g.dz - 1);
    float best[SP_K];
#pragma unroll
    for (int b = 0; b < SP_K; ++b) best[b] = INFINITY;
    float kth[SP_K];
    int found = 0;
    const int max_ring = max(g.dx, max(g.dy, g.dz));
    for (int ring = 0; ring <= max_ring; ++ring) {
        const int w = 2 * ring + 1;
        for (int t = lane; t < w * w * w; t += 64) {
            const int ddx = t % w - ring, ddy = (t / w) % w - ring, ddz = t / (w * w) - ring;
            if (max(abs(ddx), max(abs(ddy), abs(ddz))) != ring) continue;
            const int x = cx + ddx, y = cy + ddy, z = cz + ddz;
            if (x < 0 || y < 0 || z < 0 || x >= g.dx || y >= g.dy || z >= g.dz) continue;
            const int c = x + g.dx * (y + g.dy * z);
            for (uint32_t j = cstart[c]; j < cend[c]; ++j) {
                const float4 p4 = pts[j];
                float d = flann_d2(q, f3(p4.x, p4.y, p4.z));
                if (d < best[SP_K - 1]) {
#pragma unroll
                    for (int b = 0; b < SP_K; ++b)
                        if (d < best[b]) { float tt = best[b]; best[b] = d; d = tt; }
                }
            }
        }
        // wave-wide K smallest (lists are ascending: every lane offers its current head)
        int head = 0;
        found = 0;
        for (int r = 0; r < k; ++r) {
            float v = INFINITY;
#pragma unroll
            for (int b = 0; b < SP_K; ++b) if (b == head) v = best[b];
            float mv = v;
            int ml = lane;
            for (int d = 32; d >= 1; d >>= 1) {
                const float ov = __shfl_xor(mv, d, 64);
                const int ol = __shfl_xor(ml, d, 64);
                if (ov < mv || (ov == mv && ol < ml)) { mv = ov; ml = ol; }
            }
            if (mv == INFINITY) break;
            if (lane == ml) ++head;
            kth[r] = mv;
            ++found;
        }
        // everything in ring+1 and beyond is at least ring*cell away from q
        const float bound = (float)ring * g.cell * 0.999f;
        if (found == k && kth[k - 1] < bound * bound) break;
    }
    if (lane == 0) {
        double avg = 0.0;
        for (int r = 1; r < found; ++r) avg += (double)sqrtf(kth[r]);  // util.cpp:1640-1642: starts from 1 to exclude itself
        avg_out[qi] = avg;
        nbs_out[qi] = (uint32_t)found;
    }
}

// occupancy of a grid from its sorted cell keys: flag[0] = some cell holds more than `run` points,
// flag[1] = number of occupied cells
__global__ __launch_bounds__(256) void k_grid_occupancy(const uint32_t *__restrict__ keys, uint32_t n, uint32_t run,
                                                        uint32_t *__restrict__ flag) {
    // grid-stride with per-thread counts and ONE atomic per workgroup: thousands of atomics on one address
    // serialise at the memory side (this kernel took 180 us at 1M keys with one atomic per wavefront)
    __shared__ uint32_t s_c[4];
    uint32_t c = 0;
    bool dense = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t k = keys[i];
        if (i + run < n && k == keys[i + run]) dense = true;
        c += (i == 0 || k != keys[i - 1]) ? 1u : 0u;
    }
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if (__ballot(dense) && (threadIdx.x & 63) == 0) flag[0] = 1u;
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        if (t) atomicAdd(&flag[1], t);
    }
}

float average_spacing_dev(plade_ctx *ctx, const float *d_aos, uint32_t stride_f, uint32_t n, const float *bbmin,
                          const float *bbmax, int k, uint32_t samples, TargetGrid &grid) {
    PLADE_REQUIRE(k >= 1 && k <= SP_K, PLADE_ELIMIT, "average_spacing: k must be in [1, 8]");
    if (n == 0) return 0.f;
    size_t step = 1;
    if (n > samples) step = n / samples;
    const uint32_t nq = (uint32_t)((n + step - 1) / step);
    // cell ~ 2 x the spacing of a surface-like cloud filling its bounding box faces
    const double ex = std::max(1e-9, (double)bbmax[0] - bbmin[0]), ey = std::max(1e-9, (double)bbmax[1] - bbmin[1]),
                 ez = std::max(1e-9, (double)bbmax[2] - bbmin[2]);
    const double area = 2 * (ex * ey + ey * ez + ex * ez);
    float cell = (float)(2.0 * std::sqrt(area / (double)n));
    if (!(cell > 0.f)) cell = 1.f;
    // The estimate assumes a surface-like cloud.  Points on a line, in a thin slab or in tight clumps make it far
    // too coarse (thousands of points per cell: the exact ring search turns quadratic) or far too fine (every
    // point alone, hundreds of empty rings per query): adapt the cell until no cell holds more than 256 points
    // and the occupied cells hold two or more points on average (or the cell budget / duplicates forbid it).
    int last_dir = 0;
    for (int attempt = 0; attempt < 12; ++attempt) {
        grid.build(ctx, d_aos, n, stride_f, cell, bbmin, bbmax);
        if (n <= 256) break;
        uint32_t *d_flag = reinterpret_cast<uint32_t *>(ctx->scratch[2].ensure(64));
        ctx->fill_async(d_flag, 0, 8);
        launch_raw(ctx, k_grid_occupancy, dim3(std::min(cdiv(n, 1024), 512u)), dim3(256), 0, grid.keys2.p, n, 256u, d_flag);
        uint32_t occ[2] = {0, 0};
        ctx->d2h(occ, d_flag, 8);
        ctx->sync();
        const float built = 1.f / grid.gp.inv;          // build() enlarges the cell when the cell budget is hit
        int dir = 0;
        if (occ[0] && built <= cell * 1.01f) dir = -1;              // too coarse
        else if (!occ[0] && (double)occ[1] > 0.5 * (double)n) dir = +1;   // too fine
        if (dir == 0 || dir == -last_dir) break;
        cell = dir < 0 ? built * 0.25f : built * 4.f;
        last_dir = dir;
    }
    SpGrid g{grid.gp.mnx, grid.gp.mny, grid.gp.mnz, grid.gp.inv, 1.f / grid.gp.inv, grid.gp.dx, grid.gp.dy, grid.gp.dz};
    double *d_avg = reinterpret_cast<double *>(ctx->scratch[1].ensure((size_t)nq * 8 + 8));
    uint32_t *d_nbs = reinterpret_cast<uint32_t *>(ctx->scratch[2].ensure((size_t)nq * 4 + 8));
    ctx->ev_begin("knn_spacing", 12.0 * n);
    launch_raw(ctx, k_knn_grid, dim3(cdiv(nq, 4)), dim3(256), 0, grid.sorted.p, grid.cell_start.p,
                       grid.cell_end.p, g, d_aos, stride_f, n, (uint32_t)step, nq, k, d_avg, d_nbs);
    ctx->ev_end();
    HIP_TRY(hipGetLastError());
    std::vector<double> avg(nq);
    std::vector<uint32_t> nbs(nq);
    ctx->d2h(avg.data(), d_avg, (size_t)nq * 8);
    ctx->d2h(nbs.data(), d_nbs, (size_t)nq * 4);
    ctx->sync();
    // util.cpp:1630-1647: sequential double accumulation in sample order
    double total = 0.0;
    size_t total_count = 0;
    for (uint32_t i = 0; i < nq; ++i) {
        int nb = (int)nbs[i];
        if (nb <= 1) continue;
        total += (avg[i] / nb);
        ++total_count;
    }
    return static_cast<float>(total / total_count);
}

__global__ void k_strided_to_soa(const float *__restrict__ in, uint32_t n, uint32_t stride, float *__restrict__ x,
                                 float *__restrict__ y, float *__restrict__ z) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    x[i] = in[(size_t)i * stride]; y[i] = in[(size_t)i * stride + 1]; z[i] = in[(size_t)i * stride + 2];
}


__global__ __launch_bounds__(256) void k_minmax3_v(const float *__restrict__ xyz, uint32_t n, uint32_t stride, int *__restrict__ out6) {
    __shared__ float s_lds[6][8];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int k = 0; k < 3; ++k) {
            float v = xyz[(size_t)i * stride + k];
            if (!(fabsf(v) <= FLT_MAX)) out6[6] = 1;   // NaN or infinity (fminf / fmaxf would hide a NaN)
            mn[k] = fminf(mn[k], v);
            mx[k] = fmaxf(mx[k], v);
        }
    block_minmax_commit<3>(mn, mx, out6, s_lds);
}

// Queues the bounding-box reduction of a strided xyz array on `st`: d_slot (8 ints on the device) is initialised from
// h_init (8 ints of page-locked memory holding bbox_init_pattern()), reduced into, and copied to h_out (8 ints of
// page-locked memory), which bbox_decode() reads once `st` has got there.
void bbox_init_pattern(int init[8]) {
    float pinf = INFINITY, ninf = -INFINITY;
    int a, b;
    memcpy(&a, &pinf, 4); memcpy(&b, &ninf, 4);
    for (int k = 0; k < 3; ++k) { init[k] = a; init[3 + k] = b ^ 0x7fffffff; }
    init[6] = init[7] = 0;
}
void bbox_async(hipStream_t st, const float *d_xyz, uint32_t n, uint32_t stride, int *d_slot, const int *h_init, int *h_out) {
    HIP_TRY(hipMemcpyAsync(d_slot, h_init, 32, hipMemcpyHostToDevice, st));
    if (n) hipLaunchKernelGGL(k_minmax3_v, dim3(std::min(cdiv(n, 1024), 1024u)), dim3(256), 0, st, d_xyz, n, stride, d_slot);
    HIP_TRY(hipMemcpyAsync(h_out, d_slot, 32, hipMemcpyDeviceToHost, st));
}
void bbox_decode(const int out[8], float mn[3], float mx[3]) {
    // every grid, key and threshold downstream is derived from the coordinates: refuse what the reference's
    // kd-trees and voxel grids could not digest either, instead of looping on a NaN extent
    PLADE_REQUIRE(out[6] == 0, PLADE_EINVAL, "the point cloud contains non-finite coordinates (NaN or infinity)");
    for (int k = 0; k < 6; ++k) {
        int v = out[k] >= 0 ? out[k] : out[k] ^ 0x7fffffff;
        float f;
        memcpy(&f, &v, 4);
        if (k < 3) mn[k] = f; else mx[k - 3] = f;
    }
}

void bbox_host(plade_ctx *ctx, const float *d_xyz, uint32_t n, uint32_t stride, float mn[3], float mx[3]) {
    int init[8];
    bbox_init_pattern(init);
    int *d = reinterpret_cast<int *>(ctx->scratch[3].ensure(64));
    ctx->h2d(d, init, 32);
    if (n) launch_raw(ctx, k_minmax3_v, dim3(std::min(cdiv(n, 1024), 1024u)), dim3(256), 0, d_xyz, n, stride, d);
    int out[8];
    ctx->d2h(out, d, 32);
    ctx->sync();
    bbox_decode(out, mn, mx);
}
}  // namespace plade

using namespace plade;

// ---- C ABI ---------------------------------------------------------------------------------
extern "C" int plade_average_spacing(plade_ctx *ctx, const float *xyz, uint32_t n, uint32_t stride, uint32_t k,
                                     uint32_t samples, float *spacing_out) {
    return guarded(ctx, [&]() -> int {
        PLADE_REQUIRE(xyz && spacing_out && stride >= 3 && samples >= 1, PLADE_EINVAL, "plade_average_spacing: bad argument");
        *spacing_out = 0.f;
        if (n == 0) return PLADE_OK;
        DBuf<float> d_in;
        d_in.ensure((size_t)n * stride + 4);
        ctx->h2d(d_in.p, xyz, (size_t)n * stride * 4);
        float mn[3], mx[3];
        bbox_host(ctx, d_in.p, n, stride, mn, mx);
        TargetGrid grid;
        *spacing_out = average_spacing_dev(ctx, d_in.p, stride, n, mn, mx, (int)k, samples, grid);
        return PLADE_OK;
    });
}

extern "C" int plade_voxel_downsample(plade_ctx *ctx, const float *xyz, uint32_t n, uint32_t stride, float leaf,
                                      float *out_xyz, uint32_t *n_out) {
    return guarded(ctx, [&]() -> int {
        PLADE_REQUIRE(xyz && out_xyz && n_out && stride >= 3, PLADE_EINVAL, "plade_voxel_downsample: bad argument");
        *n_out = 0;
        if (n == 0) return PLADE_OK;
        DBuf<float> d_in;
        d_in.ensure((size_t)n * stride + 4);
        ctx->h2d(d_in.p, xyz, (size_t)n * stride * 4);
        float mn[3], mx[3];
        bbox_host(ctx, d_in.p, n, stride, mn, mx);
        VoxelWork w;
        uint32_t m = w.run(ctx, d_in.p, stride, nullptr, nullptr, n, 1, leaf, mn, mx);
        ctx->d2h(out_xyz, w.out_xyz.p, (size_t)m * 12);
        ctx->sync();
        *n_out = m;
        return PLADE_OK;
    });
}
