// plade_amd/csrc/ransac.hip -- plane extraction on one MI355X (seam S1b).
//
// Replaces PlaneExtraction::detect (code/PLADE/plane_extraction.cpp:61-200) and the
// RansacShapeDetector::Detect loop it drives (code/3rd_party/ransac/RansacShapeDetector.cpp:455-907).
// The reference is a sequential, time-seeded Efficient-RANSAC over octrees; it cannot reproduce its
// own output run to run, so parity is pinned per kernel (same hypothesis => identical inliers,
// identical connected component, LS plane within the reference's own accumulation noise) and at the
// plane-set level.  This is a GPU-native driver with the same semantics per accepted plane:
//
//   sampling   : DrawSamplesStratified (RansacShapeDetector.cpp:909-954) -- first point anywhere,
//                the other two from the same octree cell at a random level.  Here the cloud is
//                Morton-sorted once and "the cell at level l" is a contiguous key range.
//   hypothesis : Plane::Init(p1,p2,p3) (ransac/Plane.cpp:29-38) + the 3-sample verification
//                (RansacShapeDetector.cpp:143-153).
//   scoring    : K1.  Schnabel scores lazily on nested random subsets to save CPU time; on the GPU a
//                round of 4096 hypotheses is scored on a stratified subset in one launch and the
//                leaders are re-scored on ALL unassigned points in one HBM pass.
//   acceptance : exactly the reference's sequence (RansacShapeDetector.cpp:618-675):
//                GlobalScore(3 eps) -> ConnectedComponent(bitmap eps) -> up to 3 x
//                { LSFit -> GlobalWeightedScore(3 eps) } keeping a refit only if its weighted score
//                and size improve -> points marked assigned, drawnCandidates scaled by (1-|S|/n)^3.
//   stopping   : CandidateFailureProbability(minSupport, n_remaining, drawn, levels) <= p
//                (RansacShapeDetector.h:61-67, .cpp:856-858).
//
// Control is on the DEVICE.  The whole loop state of a cloud (remaining points, drawn candidates, candidate pool,
// accepted planes) lives in HBM (RState); one iteration is a fixed sequence of 27 launches
//     sample -> score on the subset -> leaders -> re-score the pool -> select a conflict-free batch ->
//     4 x { mark, compact + rasterise, label, select + moments, fit } -> decide -> remove points
// whose kernels read what to do from that state (a cloud that has an empty batch or has finished makes its workgroups
// return at once).  Like the reference's loop, which generates new candidates in every pass before it takes the best
// one (RansacShapeDetector.cpp:548-617), every iteration draws a round of hypotheses; what the previous batch left of
// the pool competes with them (RState::topup; plade_params.ransac_topup = 0 brings back round 2's first scheme -- draw only
// when the pool is empty, two more launches per iteration -- which needed 7 iterations instead of 5 for a 1M-point pair).
// The sequence is captured once as a hipGraph and replayed; the host never reads anything back in between -- the
// `decide` kernel reports the iteration count and the final results through a block of host-mapped memory that the
// host polls while the next (speculative) iteration is already queued.
// Every kernel serves the clouds of up to RANSAC_SLOTS "slots": the two scans of a registration -- or all scans of a group of
// registrations (plade_registration_pairs) -- are extracted in lock step by one launch sequence (blockIdx selects the cloud):
// most kernels of the loop are latency chains whose duration hardly grows with more workgroups, so every cloud added to a
// sequence divides their cost per registration.
#include "ransac.h"
#include "k1_point_test.h"
#include "prims.h"
#include "voxel.h"
#include <algorithm>
#include <map>
#include <memory>

namespace plade {

namespace {

constexpr uint32_t R_H = 4096;          // hypotheses per sampling round
constexpr uint32_t R_TOP = 48;          // candidate pool
constexpr int R_B = 8;                  // acceptance chains per cloud and iteration (16 gave the same batches: the pool
                                        // rarely holds more than 8 mutually conflict-free planes)
constexpr int R_G = RANSAC_SLOTS;
constexpr uint32_t R_MAXP = 4096;       // accepted shapes per detect call
constexpr uint32_t R_MAX_ROUNDS = 4000;
constexpr int R_MIN_LEVEL = 1, R_MAX_LEVEL = 8;
constexpr int TPB = K1_TPB, PPT = K1_PPT, TILE = K1_TILE;   // scan kernels: 1024 points per workgroup, 16 B per lane and array (k1_point_test.h)
constexpr int HCHUNK = 64;              // hypotheses per workgroup of the counting kernels (>= R_TOP)
constexpr int FIT_COLS = 14;            // 12 LS-fit moments, weighted score, kept-point count
constexpr uint32_t CC_MAXPIX = 1u << 20;
constexpr int CC_LDS_PIX = 4096;   // (r5: 8192 cost the labelling workgroup 80 KB of LDS -- half a CU; with 40 KB the batch pipeline runs 2 % faster, and every plane of a scene at bitmap eps = 2 % of its extent still fits: 50 x 40 pixels)

// ------------------------------------------------------------------------------------------------
// device-resident state
struct PlaneState {
    float n[3], pos[3], dist;        // Plane (ransac/Plane.h: m_normal, m_pos, m_dist)
    float a0[3], a1[3];              // HyperplaneCoordinateSystem axes (GfxTL/HyperplaneCoordinateSystem.h:81-93)
    float bb[4];                     // (min u, min v, max u, max v) of the score list
    uint32_t ue, ve;                 // bitmap extent
    uint32_t best_root, n_fg;        // largest component: its root pixel, its pixel count
    uint32_t n_list, n_kept;         // |score list|, |points of the largest component|
    double wscore;                   // Candidate::WeightedScore of the kept points
    float nsum[3];                   // sum of the kept points' normals (orientation)
    uint32_t err;                    // 1: bitmap too large, 2: LS fit impossible
    uint32_t converged;              // refit chain: this slot's plane is bitwise the previous slot's
};

struct ChainHdr {
    PlaneState st[4];                // slot 0 = the candidate, slot k = k-th LS refit
    float4 cand[2];                  // hypothesis (n, dist), position
    uint32_t pool_index, pad[3];
};

// Every chain owns two slabs.  The FIXED one (header, bitmap, labels) has the same layout whatever the cloud, so the
// bitmap -- which every labelling pass leaves all-zero for the next rasterisation -- stays where it is when a context
// goes from one cloud size to the next; the VARIABLE one holds the arrays whose size follows the cloud, at the byte
// offsets below (a function of the cloud size only).
constexpr uint64_t F_BMP = 4096, F_TMP = F_BMP + CC_MAXPIX, F_LABEL = F_TMP + CC_MAXPIX, F_SIZES = F_LABEL + 4ull * CC_MAXPIX,
                   F_AGG = F_SIZES + 4ull * CC_MAXPIX;
// Aggregates of a chain's score list, accumulated by the mark pass with integer / order-independent atomics (so that the
// result does not depend on the order the tiles finish in) and consumed by the compaction pass that follows: list length,
// bounding box of the (u, v) parameters, and the list's entries per SUPERTILE of 32 tiles.  With them a compaction workgroup
// finds the offset of a tile from <= nb / 32 + 31 numbers instead of re-reading the counts and boxes of all nb tiles (r3:
// every one of ~2000 workgroups per launch read ~1000 counts and up to ~300 boxes: more traffic than the compaction itself).
// All-zero = empty; the labelling kernel of the slot zeroes what the mark pass added (as it does for the bitmap).
constexpr int SUP_SHIFT = 5;
constexpr uint32_t AGG_MAX_SUP = 1u << 16;       // supertiles (2^16 x 32 tiles x 1024 points: any cloud below 2^31 points)
struct ChainAgg {
    uint32_t tot;                    // list length
    uint32_t bb[4];                  // ~enc(min u), ~enc(min v), enc(max u), enc(max v): enc = order-preserving float -> u32, 0 = none
    uint32_t pad[3];
    uint32_t sup[AGG_MAX_SUP];
};
constexpr uint64_t F_BYTES = F_AGG + ((sizeof(ChainAgg) + 255) & ~255ull);
__device__ __forceinline__ uint32_t enc_f(float x) { const uint32_t u = __float_as_uint(x); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float dec_f(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e); }
__device__ __forceinline__ void agg_add(ChainAgg *agg, uint32_t tile, uint32_t tot, float mnu, float mnv, float mxu, float mxv) {
    atomicAdd(&agg->tot, tot);
    atomicAdd(&agg->sup[tile >> SUP_SHIFT], tot);
    atomicMax(&agg->bb[0], ~enc_f(mnu));
    atomicMax(&agg->bb[1], ~enc_f(mnv));
    atomicMax(&agg->bb[2], enc_f(mxu));
    atomicMax(&agg->bb[3], enc_f(mxv));
}
__device__ __forceinline__ void agg_clear(ChainAgg *agg, uint32_t nb, uint32_t tid, uint32_t nthreads) {
    const uint32_t nsup = (nb >> SUP_SHIFT) + 1;
    for (uint32_t q = tid; q < nsup; q += nthreads) agg->sup[q] = 0u;
    if (tid < 8) (&agg->tot)[tid] = 0u;
}
struct ChainLayout {
    uint32_t nb, pad;                // tiles of the cloud
    uint64_t masks1, bc1, part, idxA, idxA_stride, masks2, masks2_stride, bc2, bc2_stride, bytes;
};

struct RResult;

struct RState {
    // ---- parameters of the running detect call
    uint32_t n, min_support, orient, active, gen;   // gen: tag of the call in the result flag
    uint32_t topup;                  // every iteration draws a round; what is left of the pool competes with the new hypotheses
    float eps, eps3, bitmap_eps, cos_t, overlook_p, bbmin[3], bbmax[3];
    uint64_t seed;
    // ---- loop state
    uint32_t done, sampling, fresh, round, it;   // fresh: the pool holds this iteration's new leaders (not yet re-scored)
    uint32_t n_remaining, sub_unassigned;
    // the SCAN VIEW: what the full passes (mark, re-score) read.  0: the Morton-ordered cloud itself (nothing has been taken yet);
    // 1 / 2: compacted copy A / B = exactly the points no shape has taken, in Morton order, with their Morton positions
    // (rebuilt behind every iteration that took points, k_view_*)
    uint32_t view_sel, view_n, view_dirty, view_next_n;
    float drawn;
    uint32_t npool, nc;
    uint32_t batch_idx[R_B];
    uint32_t pool_cnt[R_TOP];
    float4 pool_pl[R_TOP], pool_pos[R_TOP];
    uint32_t n_acc, out_off, err;
    // ---- removal jobs of the current iteration
    uint32_t aj_n, aj_chain[R_B], aj_slot[R_B], aj_out[R_B];   // aj_out: offset into out_idx, 0xffffffff = none
    int32_t aj_id[R_B];
    // ---- counters
    uint32_t n_rounds, n_rescores[2], n_batches, n_accepts, n_mark_launches, n_mark_chains, n_deferred;
    uint32_t n_final[4], n_stop[5];   // chains by the slot they ended with / by the refit at which the reference loop stops
    // ---- accepted shapes (copied to the host-mapped result block when the call ends)
    float acc_coef[R_MAXP][4];
    uint32_t acc_support[R_MAXP], acc_offset[R_MAXP];
    uint32_t acc_dbg[R_MAXP];        // (iteration << 16) | (chain << 8) | final slot: PLADE_TRACE_RANSAC
};

// Host-mapped, written by single device lanes, read by the host once `flag` says so.
struct RResult {
    uint32_t flag;                   // bits 0-23 iterations completed, 24-30 tag of the detect call (a queued speculative
                                     // iteration of the PREVIOUS call may still report), 31 = the call has finished
    uint32_t n_acc, out_off, err, remaining;
    uint32_t n_rounds, n_rescores, n_batches, n_accepts, n_mark_launches, n_mark_chains, pad;
    uint32_t n_final[4], n_stop[5], n_deferred, pad2[2];
    float coef[R_MAXP][4];
    uint32_t support[R_MAXP];        // 0: below min_support (points removed, no plane reported)
    uint32_t dbg[R_MAXP];
    uint32_t offset[R_MAXP];
};

struct CloudView {
    const float *x, *y, *z, *nx, *ny, *nz;
    uint32_t n;
};

// Everything static about one cloud slot, passed BY VALUE in the kernel arguments (scalar loads from the kernarg
// segment; a table in device memory costs every workgroup a dependent global load before it can fetch anything).
struct RCloudArgs {
    CloudView cv;                    // the Morton-ordered cloud
    const uint32_t *codes;           // its Morton codes
    const uint32_t *cells6;          // start of every level-6 octree cell in the sorted cloud (8^6 + 1 entries), or nullptr
    const uint32_t *orig;            // Morton position -> original point index (nullptr: identity, seam S1c)
    int32_t *assigned;               // seams only: the caller's shapeIndex per point (nullptr: all unassigned)
    uint8_t *taken;                  // the loop's own form: one BYTE per Morton position, 1 = taken by a shape (nullptr in the seams).
                                     // Plain byte stores from the assign kernel (a bit per point needed an atomic per taken point:
                                     // 76 -> 122 us per launch); read by the sampler's random probes and by the rebuild of the scan
                                     // view -- the full passes never look at it (they read the view)
    const float *tile_box;           // per tile of 1024 Morton-neighbouring points: min x, y, z, max x, y, z, 2 pad (nullptr: no culling)
    float *view[2];                  // scan views A / B: SoA x | y | z | nx | ny | nz with the cloud's pitch (nullptr: seams)
    uint32_t *view_map[2];           // view position -> Morton position
    uint32_t *view_cnt, *view_sup;   // rebuild: surviving points per view tile / per supertile of 32 tiles (all-zero between rebuilds)
    const float *sub;                // stratified subset, SoA with pitch sub_pitch
    const uint32_t *sub_index;
    uint32_t sub_pitch, n_sub;
    // shapeIndex of the subset's points, position for position (the subset is every sub_stride-th point of the Morton order):
    // kept up to date by k_r_assign, so that the subset scorer reads it with the same coalesced loads as the coordinates
    // instead of gathering assigned[sub_index[i]] -- a 64-byte line per point, per hypothesis chunk (r3: 184 MB of L2
    // fetches per registration for a 16 k-point subset).  nullptr (seam S1a: an arbitrary subset): gather.
    int32_t *sub_assigned;
    uint32_t sub_stride, pad0_;
    RState *st;
    RResult *res;                    // device address of the host-mapped result block
    float4 *hyp, *hyp_pos;           // R_H hypotheses of the current round
    uint32_t *hyp_counts;
    int32_t *out_idx;                // original indices of the accepted planes' supports, plane after plane
    uint32_t *out_pos;               // the same entries as positions in the Morton-ordered copy (nullptr: not wanted)
    char *fixed, *var;               // R_B slabs of F_BYTES / L.bytes
    const uint32_t *list_values;     // seam S1c: the score list is given (list position -> point), else nullptr
    ChainLayout L;
};
// What the kernels take by value: the per-cloud table lives in device memory (16 clouds x 360 bytes do not fit the 4 KB of
// kernel arguments; the kernels index it exactly as they indexed the array), the rest is small.
struct RKArgs {
    const RCloudArgs *c;             // R_G entries
    uint32_t ng, topup;              // clouds in this sequence; topup (host side): the sequence has no separate pass over the
                                     // old pool (see RState::topup)
    uint32_t tile_start[R_G + 1];    // scan grids are the concatenated tiles of the clouds: first workgroup of every cloud
    uint32_t pad_;
};
// The kernels read the table through the CONSTANT address space: its loads are then known to be invariant and, the index being
// uniform, become scalar loads wherever they stand in the kernel -- as the loads of by-value kernel arguments were.  (Through a
// plain global pointer every field read after the kernel's first store was a per-lane vector load: -2.5 % on the whole step.)
typedef const RCloudArgs __attribute__((address_space(4))) RCloudArgsK;
__device__ __forceinline__ RCloudArgsK &cloud_args(const RKArgs &A, uint32_t g) { return *(RCloudArgsK *)(A.c + g); }
// The host's form: the table as the host fills it.  args_commit() uploads it and sets `dev`; a launch converts to RKArgs.
struct RArgs {
    RCloudArgs c[R_G];
    uint32_t ng, topup;
    uint32_t tile_start[R_G + 1];
    uint32_t pad_;
    const RCloudArgs *dev;           // the committed device copy of c (nullptr: not committed)
    operator RKArgs() const {
        RKArgs k;
        k.c = dev; k.ng = ng; k.topup = topup; k.pad_ = 0;
        for (int g = 0; g <= R_G; ++g) k.tile_start[g] = tile_start[g];
        return k;
    }
};

// profiled runs: the launch's own (first wavefront in, last wavefront out) times on the device's wall clock.  A launch owns
// CLK_WAYS (start, end) pairs, workgroup b uses pair b % CLK_WAYS (thousands of atomics on ONE address would take longer
// than the kernel); the host takes the minimum / maximum over the pairs.
struct ClockScope {
    unsigned long long *p;
    __device__ explicit ClockScope(unsigned long long *q) : p(q ? q + 2 * (blockIdx.x % plade_ctx::CLK_WAYS) : nullptr) {
        if (p && threadIdx.x == 0) atomicMin(p, (unsigned long long)wall_clock64());
    }
    __device__ ~ClockScope() { if (p && (threadIdx.x & 63) == 0) atomicMax(p + 1, (unsigned long long)wall_clock64()); }
};

struct ChainPtr {
    ChainHdr *hdr;
    uint8_t *masks1; uint32_t *bc1; double *part; ChainAgg *agg;
    uint8_t *bmp, *tmp; uint32_t *label, *sizes;
    char *base; const ChainLayout __attribute__((address_space(4))) *L;   // inside the constant-space argument table
    __device__ __forceinline__ uint32_t *idxA(int k) const { return reinterpret_cast<uint32_t *>(base + L->idxA + k * L->idxA_stride); }
    __device__ __forceinline__ uint8_t *masks2(int k) const { return reinterpret_cast<uint8_t *>(base + L->masks2 + k * L->masks2_stride); }
    __device__ __forceinline__ uint32_t *bc2(int k) const { return reinterpret_cast<uint32_t *>(base + L->bc2 + k * L->bc2_stride); }
};
template <class CA>
__device__ __forceinline__ ChainPtr chain_of(const CA &C, uint32_t b) {
    ChainPtr p;
    char *base = C.var + (size_t)b * C.L.bytes, *fx = C.fixed + (size_t)b * F_BYTES;
    p.base = base; p.L = &C.L;
    p.hdr = reinterpret_cast<ChainHdr *>(fx);
    p.masks1 = reinterpret_cast<uint8_t *>(base + C.L.masks1);
    p.bc1 = reinterpret_cast<uint32_t *>(base + C.L.bc1);
    p.agg = reinterpret_cast<ChainAgg *>(fx + F_AGG);
    p.part = reinterpret_cast<double *>(base + C.L.part);
    p.bmp = reinterpret_cast<uint8_t *>(fx + F_BMP);
    p.tmp = reinterpret_cast<uint8_t *>(fx + F_TMP);
    p.label = reinterpret_cast<uint32_t *>(fx + F_LABEL);
    p.sizes = reinterpret_cast<uint32_t *>(fx + F_SIZES);
    return p;
}

ChainLayout make_layout(uint32_t n) {
    ChainLayout L;
    memset(&L, 0, sizeof(L));
    const uint64_t nb = cdiv(n, TILE), n4 = ((uint64_t)n + 7) & ~(uint64_t)3;
    L.nb = (uint32_t)nb;
    static_assert(sizeof(ChainHdr) <= F_BMP, "chain header must fit in front of the bitmap");
    uint64_t o = 0;
    auto take = [&](uint64_t bytes) { const uint64_t at = o; o = (o + bytes + 255) & ~(uint64_t)255; return at; };
    L.masks1 = take(nb * TPB);
    L.bc1 = take((nb + 4) * 4);
    L.part = take((nb + 1) * FIT_COLS * 8);
    L.idxA_stride = (n4 * 4 + 255) & ~(uint64_t)255;
    L.idxA = take(4 * L.idxA_stride);
    L.masks2_stride = (nb * TPB + 255) & ~(uint64_t)255;
    L.masks2 = take(4 * L.masks2_stride);
    L.bc2_stride = ((nb + 4) * 4 + 255) & ~(uint64_t)255;
    L.bc2 = take(4 * L.bc2_stride);
    L.bytes = o;
    return L;
}

// ------------------------------------------------------------------------------------------------
// Morton order: 8 bits per axis = the 8 octree levels the sampler draws from; the cloud slot takes the bits above the
// code (24-26) so that the clouds of a launch sequence are ordered by ONE sort
__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // up to 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FF;
    v = (v | (v << 8)) & 0x0300F00F;
    v = (v | (v << 4)) & 0x030C30C3;
    v = (v | (v << 2)) & 0x09249249;
    return v;
}

struct MortonIn {
    const float *x, *y, *z;
    uint32_t n;
    float mnx, mny, mnz, inv_cube;
};
struct MortonArgs { MortonIn c[R_G]; uint32_t ng; uint32_t start[R_G + 1]; };   // start: first item of every cloud in the joint arrays

__global__ void k_morton(const MortonArgs A, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    int g = 0;
#pragma unroll
    for (int q = 1; q < R_G; ++q) g += (q < (int)A.ng && t >= A.start[q]) ? 1 : 0;
    const MortonIn &C = A.c[g];
    const uint32_t i = t - A.start[g];
    if (i >= C.n) return;
    const uint32_t qx = min(255u, (uint32_t)max(0.f, (C.x[i] - C.mnx) * C.inv_cube * 256.f));
    const uint32_t qy = min(255u, (uint32_t)max(0.f, (C.y[i] - C.mny) * C.inv_cube * 256.f));
    const uint32_t qz = min(255u, (uint32_t)max(0.f, (C.z[i] - C.mnz) * C.inv_cube * 256.f));
    keys[t] = ((uint32_t)g << 24) | (spread3(qz) << 2) | (spread3(qy) << 1) | spread3(qx);
    vals[t] = t;
}

// Morton-order gather from the AoS copy (24 contiguous bytes per point: one or two 64 B sectors per
// point instead of six scattered 4 B reads from the SoA planes) + the stratified subset (every stride-th
// point of the Morton order)
struct GatherOut {
    const float *aos;                // the cloud's N x 6 input layout
    float *dst; uint32_t pitch;      // Morton-ordered SoA
    uint32_t *codes, *orig;
    int32_t *assigned;
    float *sub; uint32_t sub_pitch, n_sub, stride; uint32_t *sub_index;
    uint32_t n;
};
struct GatherArgs { GatherOut c[R_G]; uint32_t ng; uint32_t start[R_G + 1]; };

__global__ void k_gather_cloud(const GatherArgs A, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ perm) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    int g = 0;
#pragma unroll
    for (int q = 1; q < R_G; ++q) g += (q < (int)A.ng && t >= A.start[q]) ? 1 : 0;
    const GatherOut &C = A.c[g];
    const uint32_t i = t - A.start[g];
    if (i >= C.n) return;
    const uint32_t src = perm[t] - A.start[g];
    const float2 *p = reinterpret_cast<const float2 *>(C.aos + 6 * (size_t)src);
    const float2 a = p[0], b = p[1], c = p[2];
    const size_t pitch = C.pitch;
    C.dst[i] = a.x; C.dst[pitch + i] = a.y; C.dst[2 * pitch + i] = b.x;
    C.dst[3 * pitch + i] = b.y; C.dst[4 * pitch + i] = c.x; C.dst[5 * pitch + i] = c.y;
    C.codes[i] = keys[t] & 0xffffffu;
    C.orig[i] = src;
    if (i % C.stride == 0) {
        const uint32_t s = i / C.stride;
        if (s < C.n_sub) {
            const size_t sp = C.sub_pitch;
            C.sub_index[s] = i;
            C.sub[s] = a.x; C.sub[sp + s] = a.y; C.sub[2 * sp + s] = b.x;
            C.sub[3 * sp + s] = b.y; C.sub[4 * sp + s] = c.x; C.sub[5 * sp + s] = c.y;
        }
    }
}

// Bounding box of every tile (1024 consecutive points of the Morton order = a patch of ~0.5 m): the mark pass skips the tiles
// none of its planes' slabs can reach without loading them.  grid: the concatenated tiles of the clouds.
struct TileBoxArgs { const float *x[R_G], *y[R_G], *z[R_G]; float *box[R_G]; uint32_t n[R_G]; uint32_t ng; uint32_t tile_start[R_G + 1]; };
__global__ __launch_bounds__(K1_TPB) void k_tile_boxes(const TileBoxArgs A) {
    __shared__ float s_v[6][K1_TPB / 64];
    int g = 0;
#pragma unroll
    for (int q = 1; q < R_G; ++q) g += (q < (int)A.ng && blockIdx.x >= A.tile_start[q]) ? 1 : 0;
    const uint32_t tile = blockIdx.x - A.tile_start[g], n = A.n[g];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    const uint32_t base = tile * K1_TILE + threadIdx.x * K1_PPT;
    for (int q = 0; q < K1_PPT; ++q)
        if (base + q < n) {
            const float v[3] = {A.x[g][base + q], A.y[g][base + q], A.z[g][base + q]};
            for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], v[k]); mx[k] = fmaxf(mx[k], v[k]); }
        }
    for (int k = 0; k < 3; ++k)
        for (int d = 32; d >= 1; d >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], d, 64)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], d, 64)); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) for (int k = 0; k < 3; ++k) { s_v[k][wave] = mn[k]; s_v[3 + k][wave] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s_v[threadIdx.x][0];
        for (int w = 1; w < K1_TPB / 64; ++w) v = threadIdx.x < 3 ? fminf(v, s_v[threadIdx.x][w]) : fmaxf(v, s_v[threadIdx.x][w]);
        A.box[g][8 * (size_t)tile + threadIdx.x] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// shared device helpers
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Rng {
    uint64_t s;
    __device__ uint32_t next() { s = mix64(s); return (uint32_t)(s >> 32); }
};

__device__ __forceinline__ uint32_t lb_u32(const uint32_t *a, uint32_t n, uint32_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

// compatible(), Tile, load_tile(): k1_point_test.h (the one definition of the K1 point test)

// what a full pass reads: the cloud itself or its current compacted view (RState::view_sel)
struct ScanSrc { const float *x, *y, *z, *nx, *ny, *nz; const uint32_t *map; uint32_t n; };
template <class CA>
__device__ __forceinline__ ScanSrc scan_src(const CA &C, const RState *S) {
    ScanSrc v;
    const uint32_t sel = C.view[0] ? S->view_sel : 0u;
    if (sel == 0u) { v.x = C.cv.x; v.y = C.cv.y; v.z = C.cv.z; v.nx = C.cv.nx; v.ny = C.cv.ny; v.nz = C.cv.nz; v.map = nullptr; v.n = C.cv.n; }
    else {
        const float *b = C.view[sel - 1];
        const size_t pitch = ((size_t)C.cv.n + 3) & ~(size_t)3;
        v.x = b; v.y = b + pitch; v.z = b + 2 * pitch; v.nx = b + 3 * pitch; v.ny = b + 4 * pitch; v.nz = b + 5 * pitch;
        v.map = C.view_map[sel - 1]; v.n = S->view_n;
    }
    return v;
}

// which cloud a workgroup of a "concatenated tiles" grid scans
__device__ __forceinline__ int scan_group(const RKArgs &A, uint32_t &tile) {
    int g = 0;
#pragma unroll
    for (int q = 1; q < R_G; ++q) g += (q < (int)A.ng && blockIdx.x >= A.tile_start[q]) ? 1 : 0;
    tile = blockIdx.x - A.tile_start[g];
    return g;
}

// number of points a full pass scans (= scan_src(C, S).n)
template <class CA>
__device__ __forceinline__ uint32_t scan_n(const CA &C, const RState *S) { return (C.view[0] && S->view_sel) ? S->view_n : C.cv.n; }

// ------------------------------------------------------------------------------------------------
// Virtual grids (r5).  The per-chain kernels of the acceptance chain used to be launched over (workgroups per chain) x (R_B
// chains x R_G clouds) -- ~15 600 workgroups for a group of sixteen 1M-point clouds, of which a few dozen have anything to do
// behind the first iteration; the rest read two words of the loop state and return.  Alone on the GPU that costs little (4.6 us
// for 16 384 empty workgroups, tools/empty_grid.hip), but beside the long kernels of three other groups the empty workgroups
// queue for the few free wave slots like everybody else (k_r_compact_raster: 27 us alone, 69-87 us in the pipeline).  Now such a
// launch is a fixed, small number of workgroups: each builds the table of what every chain needs (one lane per chain, a few
// loads) and strides over the concatenation of the chains' own grids.  No result depends on the shape of the grid: every offset
// comes from counts.
constexpr int CH_E = R_G * R_B;
__device__ __forceinline__ uint32_t cdiv_d(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
constexpr uint32_t VGRID_WGS = 2048;     // workgroups of a virtual-grid launch (the host launches min(this, what the old grid was))
constexpr int OWN_MAX = 8;               // tiles a compaction workgroup owns per round
__host__ __device__ inline uint32_t loop_grid(uint32_t nb) {
    const uint32_t a = (nb + OWN_MAX - 1) / OWN_MAX;
    return a > 1024u ? 1024u : (a < 32u ? 32u : a);
}
template <class F>
__device__ __forceinline__ uint32_t chain_grid_build(const RKArgs &A, uint32_t *s_start /* CH_E + 1 */, uint32_t *s_carry, F need) {
    static_assert(CH_E == 128 && TPB >= 128, "one lane per chain, two wavefronts");
    const uint32_t e = threadIdx.x;
    uint32_t n = 0;
    if (e < (uint32_t)CH_E && e / R_B < A.ng) n = need(e / R_B, e % R_B);
    uint32_t incl = n;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (e == 63) *s_carry = incl;
    __syncthreads();
    if (e >= 64 && e < (uint32_t)CH_E) incl += *s_carry;
    if (e < (uint32_t)CH_E) s_start[e + 1] = incl;
    if (e == 0) s_start[0] = 0;
    __syncthreads();
    return s_start[CH_E];
}
__device__ __forceinline__ uint32_t chain_grid_find(const uint32_t *s_start, uint32_t vt) {   // e with start[e] <= vt < start[e + 1]
    uint32_t lo = 0, hi = CH_E;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_start[mid] <= vt) lo = mid; else hi = mid; }
    return lo;
}

// The scan kernels' virtual grid: the concatenation of what the clouds that take part in THIS launch really have to be scanned
// over -- the tiles of their current views (a third of the cloud in the second iteration, a sixth in the third, ...) instead of
// the tiles of all clouds of the sequence (the launch used to be that: workgroups of finished clouds and of the tiles beyond a
// view returned at once, but they had to be dispatched).  One lane per cloud fills the table.
template <class F>
__device__ __forceinline__ uint32_t scan_grid_build(const RKArgs &A, uint32_t *s_start /* R_G + 1 */, F units) {
    static_assert(R_G <= 64, "one lane of the first wavefront per cloud");
    const uint32_t g = threadIdx.x;
    if (g < 64) {
        const uint32_t n = (g < (uint32_t)R_G && g < A.ng) ? units(g) : 0u;
        uint32_t incl = n;
#pragma unroll
        for (int d = 1; d < R_G; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (g >= (uint32_t)d) incl += o; }
        if (g < (uint32_t)R_G) s_start[g + 1] = incl;
        if (g == 0) s_start[0] = 0;
    }
    __syncthreads();
    return s_start[R_G];
}
__device__ __forceinline__ int scan_grid_find(const uint32_t *s_start, uint32_t vt) {   // g with start[g] <= vt < start[g + 1]
    int g = 0;
#pragma unroll
    for (int q = 1; q < R_G; ++q) g += vt >= s_start[q] ? 1 : 0;
    return g;
}

__device__ __forceinline__ void hcs_axes(const float *n, float *a0, float *a1) {
    float t[3];
    if (fabsf(n[0]) < 0.015625f && fabsf(n[1]) < 0.015625f) {  // (0,1,0) x n
        t[0] = 1.f * n[2] - 0.f * n[1]; t[1] = 0.f * n[0] - 0.f * n[2]; t[2] = 0.f * n[1] - 1.f * n[0];
    } else {                                                      // (0,0,1) x n
        t[0] = 0.f * n[2] - 1.f * n[1]; t[1] = 1.f * n[0] - 0.f * n[2]; t[2] = 0.f * n[1] - 0.f * n[0];
    }
    float l = t[0] * t[0];
    l += t[1] * t[1];
    l += t[2] * t[2];
    l = sqrtf(l);
    a0[0] = t[0] / l; a0[1] = t[1] / l; a0[2] = t[2] / l;
    float u[3] = {n[1] * a0[2] - n[2] * a0[1], n[2] * a0[0] - n[0] * a0[2], n[0] * a0[1] - n[1] * a0[0]};
    l = u[0] * u[0];
    l += u[1] * u[1];
    l += u[2] * u[2];
    l = sqrtf(l);
    a1[0] = u[0] / l; a1[1] = u[1] / l; a1[2] = u[2] / l;
}

// plane frame of a slot as the scan kernels keep it in LDS: pos(3), a0(3), a1(3)
__device__ __forceinline__ void plane_uv(const float *fr, float x, float y, float z, float &u, float &v) {
    // PlanePrimitiveShape::Parameters (ransac/PlanePrimitiveShape.h:97-109)
    const float pp[3] = {x - fr[0], y - fr[1], z - fr[2]};
    u = pp[0] * fr[3] + pp[1] * fr[4] + pp[2] * fr[5];
    v = pp[0] * fr[6] + pp[1] * fr[7] + pp[2] * fr[8];
}

// state of slot 0 from a hypothesis (n, dist) + position
__device__ void state_from_hyp(PlaneState *st, float4 hyp, float4 pos) {
    st->n[0] = hyp.x; st->n[1] = hyp.y; st->n[2] = hyp.z; st->dist = hyp.w;
    st->pos[0] = pos.x; st->pos[1] = pos.y; st->pos[2] = pos.z;
    hcs_axes(st->n, st->a0, st->a1);
    st->err = 0;
    st->converged = 0;
    st->n_list = st->n_kept = 0;
    st->wscore = 0.0;
}

// ------------------------------------------------------------------------------------------------
// init: shapeIndex = -1 for the clouds that take part, loop state from the call's parameters
struct RInitCloud {
    uint32_t active, min_support, orient, gen, topup;
    float eps, eps3, bitmap_eps, cos_t, overlook_p, bbmin[3], bbmax[3];
    uint64_t seed;
};
struct RInit { RInitCloud c[R_G]; };

__global__ __launch_bounds__(TPB) void k_r_init(const RKArgs A, const RInit I) {
    uint32_t tile;
    const int g = scan_group(A, tile);
    const RCloudArgsK &C = cloud_args(A, g);
    const RInitCloud &P = I.c[g];
    if (!P.active) {
        if (tile == 0 && threadIdx.x == 0) { C.st->active = 0; C.st->done = 1; C.st->nc = 0; C.st->aj_n = 0; C.st->npool = 0; C.st->sampling = 0; C.st->fresh = 0; }
        return;
    }
    const uint32_t base = tile * TILE + threadIdx.x * PPT;
    if (C.taken && base < ((C.cv.n + 3) & ~3u)) *reinterpret_cast<uint32_t *>(C.taken + base) = 0u;   // 4 points per lane
    if (C.sub_assigned && base + PPT <= ((C.n_sub + 3) & ~3u)) *reinterpret_cast<int4 *>(C.sub_assigned + base) = make_int4(-1, -1, -1, -1);
    if (C.view_sup && tile == 0) for (uint32_t q = threadIdx.x; q < (C.L.nb >> SUP_SHIFT) + 2; q += blockDim.x) C.view_sup[q] = 0u;
    if (tile < (uint32_t)R_B) agg_clear(chain_of(C, tile).agg, C.L.nb, threadIdx.x, blockDim.x);   // (a call that died half-way left some)
    else if (C.L.nb < (uint32_t)R_B && tile == 0) for (uint32_t b2 = C.L.nb; b2 < (uint32_t)R_B; ++b2) agg_clear(chain_of(C, b2).agg, C.L.nb, threadIdx.x, blockDim.x);
    if (tile == 0 && threadIdx.x == 0) {
        RState *S = C.st;
        S->n = C.cv.n; S->min_support = P.min_support; S->orient = P.orient; S->active = 1; S->gen = P.gen; S->topup = P.topup;
        S->eps = P.eps; S->eps3 = P.eps3; S->bitmap_eps = P.bitmap_eps; S->cos_t = P.cos_t; S->overlook_p = P.overlook_p;
        for (int k = 0; k < 3; ++k) { S->bbmin[k] = P.bbmin[k]; S->bbmax[k] = P.bbmax[k]; }
        S->seed = P.seed;
        const bool nothing = C.cv.n < 3 || C.cv.n < P.min_support;   // RansacShapeDetector.cpp: no shape can reach minSupport
        S->done = nothing ? 1u : 0u; S->sampling = nothing ? 0u : 1u; S->fresh = 0; S->round = 0; S->it = 0;
        S->n_remaining = C.cv.n; S->sub_unassigned = 0; S->drawn = 0.f;
        S->view_sel = 0; S->view_n = C.cv.n; S->view_dirty = 0; S->view_next_n = 0;
        S->npool = 0; S->nc = 0; S->n_acc = 0; S->out_off = 0; S->err = 0; S->aj_n = 0;
        S->n_rounds = S->n_rescores[0] = S->n_rescores[1] = S->n_batches = S->n_accepts = S->n_mark_launches = S->n_mark_chains = 0;
        S->n_deferred = 0;
        for (int q = 0; q < 4; ++q) S->n_final[q] = 0;
        for (int q = 0; q < 5; ++q) S->n_stop[q] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// hypothesis sampling: one lane per hypothesis
__global__ __launch_bounds__(256) void k_r_sample(const RKArgs A) {
    const RCloudArgsK &C = cloud_args(A, blockIdx.y);
    RState *S = C.st;
    if (S->done || !S->sampling) return;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R_H) return;
    const auto &c = C.cv;
    const uint8_t *__restrict__ taken = C.taken;
    const uint32_t *__restrict__ codes = C.codes;
    const float eps = S->eps, cos_t = S->cos_t;
    Rng rng{mix64(S->seed ^ ((uint64_t)S->round << 32) ^ t)};
    const float nanv = __int_as_float(0x7fc00000);
    C.hyp[t] = make_float4(0.f, 0.f, 0.f, nanv);
    C.hyp_pos[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    C.hyp_counts[t] = 0;            // the scoring pass that follows accumulates with atomics
    if (t == 0) S->sub_unassigned = 0;   // and so does the unassigned-point count of the subset
    if (S->topup && t < S->npool) {
        // what the last batch left of the pool takes the first hypothesis slots and competes with the new draws on the
        // subset (w = 4: a candidate, but not a draw of this round); versions of planes that have been accepted since
        // score next to nothing there and drop out
        const float4 pp = S->pool_pos[t];
        C.hyp[t] = S->pool_pl[t];
        C.hyp_pos[t] = make_float4(pp.x, pp.y, pp.z, 4.f);
        return;
    }
    // draws are made eight at a time so that their shapeIndex look-ups are in flight together (late rounds
    // have few unassigned points left and most draws miss)
    uint32_t i0 = 0;
    bool ok = false;
    for (int tr = 0; tr < 64 && !ok; tr += 8) {
        uint32_t cand[8];
        int32_t av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cand[u] = rng.next() % c.n;
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = taken[cand[u]] ? 0 : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (!ok && av[u] == -1) { i0 = cand[u]; ok = true; }
    }
    if (!ok) return;
    const int level = R_MIN_LEVEL + (int)(rng.next() % (uint32_t)(R_MAX_LEVEL - R_MIN_LEVEL + 1));
    const uint32_t low_bits = 24 - 3 * level;
    const uint32_t mask = low_bits >= 32 ? 0u : ~((1u << low_bits) - 1u);
    const uint32_t lo_key = codes[i0] & mask, hi_key = lo_key | ~mask;
    uint32_t lo, hi;
    if (C.cells6) {
        // the cell's range from the level-6 table: two look-ups down to level 6, below that a search inside one level-6 cell
        // (instead of two 20-step binary searches over the whole cloud: the kernel is a chain of dependent loads)
        if (level <= 6) {
            const uint32_t c0 = lo_key >> 6, span = 1u << (3 * (6 - level));
            lo = C.cells6[c0];
            hi = C.cells6[c0 + span];
        } else {
            const uint32_t c0 = lo_key >> 6;
            const uint32_t b6 = C.cells6[c0], e6 = C.cells6[c0 + 1];
            lo = b6 + lb_u32(codes + b6, e6 - b6, lo_key);
            hi = (hi_key >= 0xffffffu) ? e6 : b6 + lb_u32(codes + b6, e6 - b6, hi_key + 1u);
        }
    } else {
        lo = lb_u32(codes, c.n, lo_key);
        hi = (hi_key == 0xffffffffu || hi_key >= 0xffffffu) ? c.n : lb_u32(codes, c.n, hi_key + 1u);
    }
    if (hi - lo < 3) return;
    uint32_t s[3] = {i0, 0, 0};
    int got = 1;
    for (int tr = 0; tr < 40 && got < 3; tr += 8) {
        uint32_t cand[8];
        int32_t av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cand[u] = lo + rng.next() % (hi - lo);
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = taken[cand[u]] ? 0 : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (got >= 3 || av[u] != -1) continue;
            bool dup = false;
            for (int q = 0; q < got; ++q) dup = dup || (s[q] == cand[u]);
            if (!dup) s[got++] = cand[u];
        }
    }
    if (got < 3) return;
    // three samples drawn: this counts as a generated candidate (genCands, RansacShapeDetector.cpp:122-125)
    // whether or not the plane survives construction / verification below
    C.hyp_pos[t] = make_float4(0.f, 0.f, 0.f, 2.f);
    // Plane::Init (ransac/Plane.cpp:29-38)
    const float p1[3] = {c.x[s[0]], c.y[s[0]], c.z[s[0]]}, p2[3] = {c.x[s[1]], c.y[s[1]], c.z[s[1]]},
                p3[3] = {c.x[s[2]], c.y[s[2]], c.z[s[2]]};
    const float a[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, b[3] = {p3[0] - p2[0], p3[1] - p2[1], p3[2] - p2[2]};
    float nr[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    float sq = nr[0] * nr[0];
    sq += nr[1] * nr[1];
    sq += nr[2] * nr[2];
    if (sq < 1E-6f) return;
    const float len = sqrtf(sq);
    nr[0] /= len; nr[1] /= len; nr[2] /= len;
    float dist = p1[0] * nr[0];
    dist += p1[1] * nr[1];
    dist += p1[2] * nr[2];
    // verify the three samples (RansacShapeDetector.cpp:143-153)
    for (int k = 0; k < 3; ++k) {
        float d = nr[0] * c.x[s[k]];
        d += nr[1] * c.y[s[k]];
        d += nr[2] * c.z[s[k]];
        float nd = nr[0] * c.nx[s[k]];
        nd += nr[1] * c.ny[s[k]];
        nd += nr[2] * c.nz[s[k]];
        if (!(fabsf(dist - d) < eps && fabsf(nd) >= cos_t)) return;
    }
    C.hyp[t] = make_float4(nr[0], nr[1], nr[2], dist);
    C.hyp_pos[t] = make_float4(p1[0], p1[1], p1[2], 1.f);
}

// Which of the first `np` (<= 64) planes of `s_pl` can have an inlier among the unassigned points this wavefront holds?
// Bit h is CLEAR only when plane h's eps slab provably misses the bounding box of those points (the interval of n.p over the
// box, widened far beyond the rounding of the three-term sums); planes with a NaN distance are cleared too (nothing is
// compatible with them).  The points of a wavefront are neighbours on the Morton curve, i.e. a small box, and most planes
// of a pool or of a chunk of hypotheses pass nowhere near it.
__device__ __forceinline__ unsigned long long slab_mask(const Tile &t, const float4 *s_pl, uint32_t np, float eps, int lane) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int k = 0; k < K1_PPT; ++k)
        if (t.valid[k]) {
            mn[0] = fminf(mn[0], t.px[k]); mn[1] = fminf(mn[1], t.py[k]); mn[2] = fminf(mn[2], t.pz[k]);
            mx[0] = fmaxf(mx[0], t.px[k]); mx[1] = fmaxf(mx[1], t.py[k]); mx[2] = fmaxf(mx[2], t.pz[k]);
        }
#pragma unroll
    for (int q = 0; q < 3; ++q)
        for (int d = 32; d >= 1; d >>= 1) { mn[q] = fminf(mn[q], __shfl_xor(mn[q], d, 64)); mx[q] = fmaxf(mx[q], __shfl_xor(mx[q], d, 64)); }
    bool near = false;
    if ((uint32_t)lane < np && mn[0] <= mx[0]) {
        const float4 pl = s_pl[lane];
        const float lo = fminf(pl.x * mn[0], pl.x * mx[0]) + fminf(pl.y * mn[1], pl.y * mx[1]) + fminf(pl.z * mn[2], pl.z * mx[2]);
        const float hi = fmaxf(pl.x * mn[0], pl.x * mx[0]) + fmaxf(pl.y * mn[1], pl.y * mx[1]) + fmaxf(pl.z * mn[2], pl.z * mx[2]);
        const float slack = 1.001f * eps + 1e-5f * (fabsf(lo) + fabsf(hi) + fabsf(pl.w));
        near = pl.w == pl.w && !(pl.w - hi > slack || lo - pl.w > slack);   // a NaN normal keeps the plane in: the exact test decides
    }
    return __ballot(near);
}

// K1 on the stratified subset: grid (subset tiles, hypothesis chunks, clouds)
// (virtual grid: subset tiles x hypothesis chunks of the clouds that are sampling)
__global__ __launch_bounds__(TPB) void k_r_score_sub(const RKArgs A) {
    __shared__ float4 s_pl[HCHUNK];
    __shared__ uint32_t s_cnt[TPB / 64][HCHUNK];
    __shared__ uint32_t s_start[R_G + 1];
    const uint32_t total = scan_grid_build(A, s_start, [&](uint32_t g) -> uint32_t {
        const RCloudArgsK &C = cloud_args(A, g);
        const RState *S = C.st;
        return (S->done || !S->sampling) ? 0u : cdiv_d(C.n_sub, TILE) * (R_H / HCHUNK);
    });
    for (uint32_t vt = blockIdx.x; vt < total; vt += gridDim.x) {
    const int g = __builtin_amdgcn_readfirstlane(scan_grid_find(s_start, vt));
    __syncthreads();   // the previous round's readers of the LDS arrays are done
    const RCloudArgsK &C = cloud_args(A, g);
    RState *S = C.st;
    const uint32_t sub_tiles = cdiv_d(C.n_sub, TILE), v = vt - s_start[g];
    const uint32_t bx = v % sub_tiles, by = v / sub_tiles;
    const uint32_t h0 = by * HCHUNK;
    if (threadIdx.x < HCHUNK) s_pl[threadIdx.x] = C.hyp[h0 + threadIdx.x];
    const size_t sp = C.sub_pitch;
    const float eps = S->eps, cos_t = S->cos_t;
    Tile t;
    load_tile(t, C.sub, C.sub + sp, C.sub + 2 * sp, C.sub + 3 * sp, C.sub + 4 * sp, C.sub + 5 * sp,
              C.sub_assigned ? C.sub_assigned : C.assigned, C.sub_assigned ? nullptr : C.sub_index, C.n_sub,
              bx * TILE + threadIdx.x * PPT);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (by == 0) {   // unassigned points of the subset (estimates the support on the whole cloud)
        uint32_t un = 0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) un += (uint32_t)__popcll(__ballot(t.valid[k]));
        if (lane == 0 && un) atomicAdd(&S->sub_unassigned, un);
    }
    // (a draw that gave no verified plane carries a NaN distance: nothing is compatible with it, slab_mask drops it)
    unsigned long long live = slab_mask(t, s_pl, HCHUNK, eps, lane);
    s_cnt[wave][lane] = 0u;
    while (live) {                       // wave-uniform
        const uint32_t hh = (uint32_t)__ffsll((long long)live) - 1u;
        live &= live - 1ull;
        const float4 pl = s_pl[hh];
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const bool in = t.valid[k] && compatible(pl, t.px[k], t.py[k], t.pz[k], t.qx[k], t.qy[k], t.qz[k], eps, cos_t);
            c += (uint32_t)__popcll(__ballot(in));
        }
        if (lane == 0) s_cnt[wave][hh] = c;
    }
    __syncthreads();
    if (threadIdx.x < HCHUNK) {
        const uint32_t tot = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
        if (tot) atomicAdd(&C.hyp_counts[h0 + threadIdx.x], tot);
    }
    }
}

__device__ __forceinline__ bool same_plane(const float4 &a, const float4 &b, float eps) {
    const float c = a.x * b.x + a.y * b.y + a.z * b.z;
    if (fabsf(c) < 0.995f) return false;
    const float db = c >= 0 ? b.w : -b.w;
    return fabsf(a.w - db) < 2 * eps;
}

// Leaders of a round: the hypotheses by estimated support, one representative per distinct plane -- a greedy pass in
// descending order that drops what duplicates an earlier pick (= repeatedly take the best candidate alive and strike
// out its duplicates).  One workgroup of 1024 lanes per cloud, four hypotheses per lane; a pick costs one wave-level
// arg-max of 32-bit keys (count << 12 | 4095 - index: the subset holds fewer than 2^20 points), ONE barrier (the wave
// winners and their planes go through double-buffered LDS) and four duplicate tests per lane.
__global__ __launch_bounds__(1024) void k_r_leaders(const RKArgs A) {
    const RCloudArgsK &C = cloud_args(A, blockIdx.x);
    RState *S = C.st;
    if (S->done || !S->sampling) return;
    __shared__ uint32_t s_key[2][16];
    __shared__ float4 s_pl[2][16];
    __shared__ uint32_t s_val[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t sub_un = S->sub_unassigned, n_rem = S->n_remaining, ms = S->min_support;
    const float eps = S->eps;
    const double ratio = sub_un ? (double)n_rem / sub_un : 0.0;
    uint32_t key[4];
    float4 pl[4];
    uint32_t valid = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t i = tid + q * 1024;
        const float4 pos = C.hyp_pos[i];
        const uint32_t c = C.hyp_counts[i];
        pl[q] = C.hyp[i];
        valid += pos.w == 1.f || pos.w == 2.f;   // w: 0 no samples, 1 verified plane, 2 drawn but rejected, 4 kept from the pool
        const bool ok = (pos.w == 1.f || pos.w == 4.f) && c * ratio >= 0.5 * ms;
        key[q] = ok ? ((min(c, 0xfffffu) << 12) | (0xfffu - i)) : 0u;   // count descending, index ascending; 0 = not a candidate
    }
    for (int d = 32; d >= 1; d >>= 1) valid += __shfl_xor(valid, d, 64);
    if (lane == 0) s_val[wave] = valid;
    uint32_t npool = 0;
    for (; npool < R_TOP; ++npool) {
        const int buf = npool & 1;
        // this lane's best, then the wave's
        uint32_t m = key[0];
        int mq = 0;
        if (key[1] > m) { m = key[1]; mq = 1; }
        if (key[2] > m) { m = key[2]; mq = 2; }
        if (key[3] > m) { m = key[3]; mq = 3; }
        uint32_t wm = m;
        for (int d = 32; d >= 1; d >>= 1) wm = max(wm, (uint32_t)__shfl_xor((int)wm, d, 64));
        if (m == wm && wm != 0u) {   // keys are unique: exactly one lane of the wave
            s_key[buf][wave] = wm;
            float4 v = pl[0];
            if (mq == 1) v = pl[1]; else if (mq == 2) v = pl[2]; else if (mq == 3) v = pl[3];
            s_pl[buf][wave] = v;
        } else if (lane == 0 && wm == 0u) s_key[buf][wave] = 0u;
        __syncthreads();
        uint32_t best = 0;
        int bw = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const uint32_t k = s_key[buf][w]; if (k > best) { best = k; bw = w; } }
        if (best == 0u) break;     // uniform
        const float4 nw = s_pl[buf][bw];
        if (tid == 0) {
            const uint32_t idx = 0xfffu - (best & 0xfffu);
            S->pool_pl[npool] = nw;
            S->pool_pos[npool] = C.hyp_pos[idx];
        }
        // Only near-identical planes are struck out.  (Striking out every tilted version of the pick as well -- candidates
        // within 25 degrees whose sample point lies in the pick's 3 eps band -- fills the pool with distinct surfaces and
        // saves a quarter of the iterations, but small planes then enter the early batches; on 1 of 16 synthetic pairs the
        // registration that followed chose a symmetric alignment of the room.  Measured on the MI355X, DESIGN.md.)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (key[q] && same_plane(nw, pl[q], eps)) key[q] = 0u;   // strikes the pick itself too
    }
    __syncthreads();
    if (tid < (int)R_TOP) S->pool_cnt[tid] = 0;
    if (tid == 0) {
        uint32_t v = 0;
        for (int w = 0; w < 16; ++w) v += s_val[w];
        S->drawn += (float)v;
        S->npool = npool;
        S->round += 1;
        S->n_rounds += 1;
        S->sampling = 0;
        S->fresh = 1;
    }
}

// K1: the pool re-scored on ALL unassigned points of its cloud: one HBM pass per cloud, the pool's planes in LDS
// phase 0: what the previous iteration left in the pool; phase 1: the leaders of a round drawn in this iteration
__global__ __launch_bounds__(TPB) void k_r_rescore(const RKArgs A, int phase, unsigned long long *clk) {
    const ClockScope clock_scope(clk);
    __shared__ float4 s_pl[HCHUNK];
    __shared__ uint32_t s_cnt[TPB / 64][HCHUNK];
    __shared__ uint32_t s_start[R_G + 1];
    // virtual grid: the view tiles of the clouds whose pool is re-scored in this phase (at least one each: tile 0 keeps the launch counter)
    const uint32_t total = scan_grid_build(A, s_start, [&](uint32_t g) -> uint32_t {
        const RCloudArgsK &C = cloud_args(A, g);
        const RState *S = C.st;
        if (S->done || S->npool == 0 || (S->fresh != 0u) != (phase != 0)) return 0u;
        return max(1u, cdiv_d(scan_n(C, S), TILE));
    });
    for (uint32_t vt = blockIdx.x; vt < total; vt += gridDim.x) {
    const int g = __builtin_amdgcn_readfirstlane(scan_grid_find(s_start, vt));
    const uint32_t tile = vt - s_start[g];
    __syncthreads();   // the previous round's readers of the LDS arrays are done
    const RCloudArgsK &C = cloud_args(A, g);
    RState *S = C.st;
    const uint32_t np = S->npool;
    const ScanSrc V = scan_src(C, S);
    if (threadIdx.x < np) s_pl[threadIdx.x] = S->pool_pl[threadIdx.x];
    const float eps = S->eps, cos_t = S->cos_t;
    Tile t;
    // A view holds exactly the points no shape has taken (the cloud itself while nothing has been taken: the view is rebuilt behind
    // every iteration that takes points): no shapeIndex to look at.  (The seams scan the caller's cloud with its shapeIndex array.)
    load_tile(t, V.x, V.y, V.z, V.nx, V.ny, V.nz, V.map ? nullptr : C.assigned, nullptr, V.n, tile * TILE + threadIdx.x * PPT);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // only the planes whose slab can reach this wavefront's points get the exact test (slab_mask)
    unsigned long long live = slab_mask(t, s_pl, np, eps, lane);
    if ((uint32_t)lane < np) s_cnt[wave][lane] = 0u;
    while (live) {                       // wave-uniform
        const uint32_t hh = (uint32_t)__ffsll((long long)live) - 1u;
        live &= live - 1ull;
        const float4 pl = s_pl[hh];
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const bool in = t.valid[k] && compatible(pl, t.px[k], t.py[k], t.pz[k], t.qx[k], t.qy[k], t.qz[k], eps, cos_t);
            c += (uint32_t)__popcll(__ballot(in));
        }
        if (lane == 0) s_cnt[wave][hh] = c;
    }
    __syncthreads();
    if (threadIdx.x < np) {
        const uint32_t tot = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
        if (tot) atomicAdd(&S->pool_cnt[threadIdx.x], tot);
    }
    if (tile == 0 && threadIdx.x == 0) S->n_rescores[phase] += 1;
    }
}

// CandidateFailureProbability (RansacShapeDetector.h:61-67, reqSamples = 3)
__device__ __forceinline__ float fail_prob(float cand_size, float n_pts, float drawn) {
    const float levels = (float)(R_MAX_LEVEL - R_MIN_LEVEL + 1);
    return fminf(powf(1.f - cand_size / (n_pts * levels * 4.f), drawn), 1.f);
}

// the pool is empty: draw another round, or stop (RansacShapeDetector.cpp:856-858 and the loop head)
__device__ void next_round_or_stop(RState *S) {
    if (S->n_remaining < S->min_support) S->done = 1;
    else if (S->round > 0 && fail_prob((float)S->min_support, (float)S->n_remaining, S->drawn) <= S->overlook_p) S->done = 1;
    else if (S->round >= R_MAX_ROUNDS) S->done = 1;
    else S->sampling = 1;
}

// Can a point be an inlier (|dist| < 3 eps AND |n.n_p| >= cos_t) of both planes?  Provably not when
//  (a) the normals are far apart: n_p within acos(cos_t) of both +-n_a and +-n_b needs
//      angle(n_a, n_b) <= 2 acos(cos_t) (or its supplement); a 5 degree margin covers the refits; or
//  (b) the planes are nearly parallel and, everywhere inside the cloud's bounding box, more than
//      8 eps apart (the two 3 eps bands plus refit drift cannot meet): the difference of the signed
//      distances is linear in p, so constant sign at the 8 corners + min |.| at a corner decide it.
// Anything else counts as a conflict and the two candidates are accepted one after the other.
// cos_gate: cosine of 2 acos(cos_t) + 5 degrees when that angle is below 90 degrees, else -1 (criterion (a) never holds)
__device__ __forceinline__ float conflict_gate(float cos_t) {
    const float two_theta = 2.f * acosf(fminf(1.f, cos_t)) + 0.0873f;
    return two_theta < 1.5707f ? cosf(two_theta) : -1.f;
}
__device__ bool conflict_free(const float4 &a, const float4 &b, float eps, float cos_gate, const float *bbmin, const float *bbmax) {
    const float c = fabsf(a.x * b.x + a.y * b.y + a.z * b.z);
    if (c < cos_gate) return true;
    if (c < 0.97f) return false;
    const float s = (a.x * b.x + a.y * b.y + a.z * b.z) >= 0 ? 1.f : -1.f;
    float mn = INFINITY, mx = -INFINITY;
    for (int k = 0; k < 8; ++k) {
        const float px = (k & 1) ? bbmax[0] : bbmin[0], py = (k & 2) ? bbmax[1] : bbmin[1], pz = (k & 4) ? bbmax[2] : bbmin[2];
        const float da = a.x * px + a.y * py + a.z * pz - a.w, db = s * (b.x * px + b.y * py + b.z * pz) - s * b.w;
        mn = fminf(mn, da - db);
        mx = fmaxf(mx, da - db);
    }
    return (mn > 8 * eps) || (mx < -8 * eps);
}

// After the re-score: candidates that can no longer reach min_support are dropped (RansacShapeDetector.cpp:826-832),
// the rest is ordered by support, and the best candidate plus every further one whose support provably cannot touch
// the support of any better candidate become this iteration's acceptance chains (accepting them concurrently equals
// accepting them one by one, best first).  One wavefront per cloud.
__global__ __launch_bounds__(64) void k_r_select(const RKArgs A, int phase) {
    const RCloudArgsK &C = cloud_args(A, blockIdx.x);
    RState *S = C.st;
    const int lane = threadIdx.x;
    if (phase == 0) {
        // leftovers of the previous iteration: prune + batch; an empty pool asks for a new round, which the kernels that
        // follow in this same iteration draw
        if (S->done || S->sampling) { if (lane == 0) { S->nc = 0; S->aj_n = 0; } return; }
    } else {
        if (S->done || !S->fresh) return;   // the batch (if any) was chosen in phase 0
    }
    __shared__ float4 s_pl[R_TOP], s_pos[R_TOP];
    __shared__ uint32_t s_cnt[R_TOP], s_raw[R_TOP], s_keep[R_TOP];
    const uint32_t np = S->npool, ms = S->min_support;
    uint32_t cnt = 0;
    bool keep = false;
    float4 pl = make_float4(0, 0, 0, 0), pos = pl;
    if ((uint32_t)lane < np) { cnt = S->pool_cnt[lane]; keep = cnt >= ms; pl = S->pool_pl[lane]; pos = S->pool_pos[lane]; }
    if (lane < (int)R_TOP) { s_raw[lane] = cnt; s_keep[lane] = keep ? 1u : 0u; }
    __syncthreads();
    // stable descending order of the kept entries
    uint32_t rank = 0;
    if (keep)
        for (uint32_t j = 0; j < np; ++j) rank += s_keep[j] && ((s_raw[j] > cnt) || (s_raw[j] == cnt && j < (uint32_t)lane));
    const uint32_t np2 = (uint32_t)__popcll(__ballot(keep));
    if (keep) { s_pl[rank] = pl; s_pos[rank] = pos; s_cnt[rank] = cnt; }
    __syncthreads();
    if (np2 == 0) {
        if (lane == 0) { S->npool = 0; S->nc = 0; S->aj_n = 0; S->fresh = 0; next_round_or_stop(S); }
        return;
    }
    const float eps = S->eps, cos_t = conflict_gate(S->cos_t);
    float bbmin[3], bbmax[3];
    for (int k = 0; k < 3; ++k) { bbmin[k] = S->bbmin[k]; bbmax[k] = S->bbmax[k]; }
    __shared__ uint32_t s_batch[R_B];
    // candidate i joins the batch only if its support cannot touch the support of ANY better candidate of the pool, in
    // the batch or not: it would then be accepted before a conflicting better one, which the reference, taking the best
    // candidate first every time (RansacShapeDetector.cpp:548-617), never does.  Whether i qualifies therefore does not
    // depend on who is in the batch: the rows are tested one after the other without a barrier (lane j holds candidate j),
    // the batch is the best candidate plus the first R_B - 1 candidates that qualify
    unsigned long long free_rows = 1ull;
    for (uint32_t i = 1; i < np2; ++i) {
        bool conflict = false;
        if ((uint32_t)lane < i) conflict = !conflict_free(s_pl[i], s_pl[lane], eps, cos_t, bbmin, bbmax);
        if (__ballot(conflict) == 0ull) free_rows |= 1ull << i;
    }
    const uint32_t nb = min((uint32_t)__popcll(free_rows), (uint32_t)R_B);
    if ((uint32_t)lane < nb) {          // lane l: the l-th qualifying candidate
        unsigned long long m = free_rows;
        for (int q = 0; q < lane; ++q) m &= m - 1ull;
        s_batch[lane] = (uint32_t)__ffsll((long long)m) - 1u;
    }
    __syncthreads();
    // the ordered pool back to the state, counts cleared for the next re-score
    if ((uint32_t)lane < np2) { S->pool_pl[lane] = s_pl[lane]; S->pool_pos[lane] = s_pos[lane]; }
    if (lane < (int)R_TOP) S->pool_cnt[lane] = 0;
    if ((uint32_t)lane < nb) {
        const uint32_t bi = s_batch[lane];
        S->batch_idx[lane] = bi;
        ChainPtr ch = chain_of(C, lane);
        ch.hdr->cand[0] = s_pl[bi];
        ch.hdr->cand[1] = s_pos[bi];
        ch.hdr->pool_index = bi;
        state_from_hyp(&ch.hdr->st[0], s_pl[bi], s_pos[bi]);
    }
    if (lane == 0) { S->npool = np2; S->nc = nb; S->aj_n = 0; S->fresh = 0; S->n_batches += 1; }
}

// ------------------------------------------------------------------------------------------------
// acceptance chain, slot k: GlobalWeightedScore (Candidate.h:293-302) = score(3 eps) -> ConnectedComponent ->
// weighted score, then the LS fit of the result list.
//
// (1) mark: ONE pass over the cloud for all chains of the cloud: 4-bit inlier masks per lane, per-tile counts, and the
//     aggregates of every chain's list (ChainAgg: length, bounding box of the inliers' (u, v) plane parameters,
//     BitmapPrimitiveShape.h:113-126, entries per supertile)
__global__ __launch_bounds__(TPB) void k_r_mark(const RKArgs A, int k, unsigned long long *clk) {
    const ClockScope clock_scope(clk);
    __shared__ float4 s_pl[R_B];
    __shared__ float s_fr[R_B][9];
    __shared__ uint32_t s_skip[R_B], s_need[R_B];
    __shared__ uint32_t s_w[R_B][TPB / 64];
    __shared__ float s_bb[R_B][4][TPB / 64];
    __shared__ uint32_t s_start[R_G + 1];
    // virtual grid: the view tiles of the clouds that have chains in this iteration
    const uint32_t total = scan_grid_build(A, s_start, [&](uint32_t g) -> uint32_t {
        const RCloudArgsK &C = cloud_args(A, g);
        const RState *S = C.st;
        return S->nc == 0 ? 0u : max(1u, cdiv_d(scan_n(C, S), TILE));
    });
    for (uint32_t vt = blockIdx.x; vt < total; vt += gridDim.x) {
    const int g = __builtin_amdgcn_readfirstlane(scan_grid_find(s_start, vt));
    const uint32_t tile = vt - s_start[g];
    __syncthreads();   // the previous round's readers of the LDS arrays are done
    const RCloudArgsK &C = cloud_args(A, g);
    RState *S = C.st;
    const uint32_t nc = S->nc;
    if (threadIdx.x < nc) {
        const PlaneState *st = &chain_of(C, threadIdx.x).hdr->st[k];
        s_skip[threadIdx.x] = st->converged;
        // Can this plane's 3 eps slab reach the tile at all?  The interval of n.p over the tile's bounding box, widened far beyond
        // the rounding of the three-term sums (as slab_mask does per wavefront); a plane that cannot gets a zero count for the
        // tile without a point being loaded.  (A NaN distance -- a slot whose LS fit was impossible -- reaches nothing.)
        uint32_t need = 1u;
        if (C.tile_box && !(C.view[0] && S->view_sel)) {   // (the boxes are those of the cloud's own tiles)
            const float *bx = C.tile_box + 8 * (size_t)tile;
            const float e3 = S->eps3;
            const float lo = fminf(st->n[0] * bx[0], st->n[0] * bx[3]) + fminf(st->n[1] * bx[1], st->n[1] * bx[4]) + fminf(st->n[2] * bx[2], st->n[2] * bx[5]);
            const float hi = fmaxf(st->n[0] * bx[0], st->n[0] * bx[3]) + fmaxf(st->n[1] * bx[1], st->n[1] * bx[4]) + fmaxf(st->n[2] * bx[2], st->n[2] * bx[5]);
            const float slack = 1.001f * e3 + 1e-5f * (fabsf(lo) + fabsf(hi) + fabsf(st->dist));
            need = (st->dist == st->dist && !(st->dist - hi > slack || lo - st->dist > slack)) ? 1u : 0u;
            if (!(lo == lo && hi == hi)) need = st->dist == st->dist ? 1u : 0u;   // a NaN normal: let the exact test decide
        }
        s_need[threadIdx.x] = need;
        s_pl[threadIdx.x] = make_float4(st->n[0], st->n[1], st->n[2], st->dist);
        float *fr = s_fr[threadIdx.x];
        fr[0] = st->pos[0]; fr[1] = st->pos[1]; fr[2] = st->pos[2];
        fr[3] = st->a0[0]; fr[4] = st->a0[1]; fr[5] = st->a0[2];
        fr[6] = st->a1[0]; fr[7] = st->a1[1]; fr[8] = st->a1[2];
    }
    const float eps = S->eps3, cos_t = S->cos_t;
    const ScanSrc V = scan_src(C, S);
    __syncthreads();
    uint32_t active = 0, needed = 0;
    for (uint32_t j = 0; j < nc; ++j) { active += s_skip[j] ? 0u : 1u; needed += (!s_skip[j] && s_need[j]) ? 1u : 0u; }
    if (tile * TILE >= V.n) needed = 0;      // beyond the (compacted) view: an empty tile
    if (needed == 0) {   // uniform: no slab reaches this tile (about half of all (tile, launch) pairs): nothing is loaded
        if (threadIdx.x < nc && !s_skip[threadIdx.x]) chain_of(C, threadIdx.x).bc1[tile] = 0u;
        if (tile == 0 && threadIdx.x == 0 && active) { S->n_mark_launches += 1; S->n_mark_chains += active; }
        continue;
    }
    Tile t;
    load_tile(t, V.x, V.y, V.z, V.nx, V.ny, V.nz, V.map ? nullptr : C.assigned, nullptr, V.n, tile * TILE + threadIdx.x * PPT);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t j = 0; j < nc; ++j) {
        if (s_skip[j]) continue;   // uniform
        if (!s_need[j]) { if (lane == 0) s_w[j][wave] = 0u; continue; }   // uniform: provably no inlier here
        const float4 pl = s_pl[j];
        uint32_t m = 0, c = 0;
        float mn0 = INFINITY, mn1 = INFINITY, mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const bool in = t.valid[q] && compatible(pl, t.px[q], t.py[q], t.pz[q], t.qx[q], t.qy[q], t.qz[q], eps, cos_t);
            m |= (in ? 1u : 0u) << q;
            c += (uint32_t)__popcll(__ballot(in));
            if (in) {
                float u, v;
                plane_uv(s_fr[j], t.px[q], t.py[q], t.pz[q], u, v);
                mn0 = fminf(mn0, u); mn1 = fminf(mn1, v); mx0 = fmaxf(mx0, u); mx1 = fmaxf(mx1, v);
            }
        }
        chain_of(C, j).masks1[tile * TPB + threadIdx.x] = (uint8_t)m;
        if (c) {   // wave-uniform
            for (int d = 32; d >= 1; d >>= 1) {
                mn0 = fminf(mn0, __shfl_xor(mn0, d, 64)); mn1 = fminf(mn1, __shfl_xor(mn1, d, 64));
                mx0 = fmaxf(mx0, __shfl_xor(mx0, d, 64)); mx1 = fmaxf(mx1, __shfl_xor(mx1, d, 64));
            }
        }
        if (lane == 0) { s_w[j][wave] = c; s_bb[j][0][wave] = mn0; s_bb[j][1][wave] = mn1; s_bb[j][2][wave] = mx0; s_bb[j][3][wave] = mx1; }
    }
    __syncthreads();
    if (threadIdx.x < nc && !s_skip[threadIdx.x]) {
        const uint32_t j = threadIdx.x;
        const ChainPtr ch = chain_of(C, j);
        const uint32_t tot = s_w[j][0] + s_w[j][1] + s_w[j][2] + s_w[j][3];
        ch.bc1[tile] = tot;
        if (tot)
            agg_add(ch.agg, tile, tot, fminf(fminf(s_bb[j][0][0], s_bb[j][0][1]), fminf(s_bb[j][0][2], s_bb[j][0][3])),
                    fminf(fminf(s_bb[j][1][0], s_bb[j][1][1]), fminf(s_bb[j][1][2], s_bb[j][1][3])),
                    fmaxf(fmaxf(s_bb[j][2][0], s_bb[j][2][1]), fmaxf(s_bb[j][2][2], s_bb[j][2][3])),
                    fmaxf(fmaxf(s_bb[j][3][0], s_bb[j][3][1]), fmaxf(s_bb[j][3][2], s_bb[j][3][3])));
    }
    if (tile == 0 && threadIdx.x == 0 && active) { S->n_mark_launches += 1; S->n_mark_chains += active; }
    }
}

// seam S1c: the score list is given by the caller.  Same outputs as k_r_mark for chain 0 over LIST POSITIONS
// (all positions < m are "inliers"); tiles past the list get a zero count.
__global__ __launch_bounds__(TPB) void k_r_list_mark(const RKArgs A, uint32_t m) {
    __shared__ float s_bb[4][TPB / 64];
    const RCloudArgsK &C = cloud_args(A, 0);
    const ChainPtr ch = chain_of(C, 0);
    const PlaneState *st = &ch.hdr->st[0];
    const uint32_t tile = blockIdx.x, first = tile * TILE + threadIdx.x * PPT;
    float fr[9] = {st->pos[0], st->pos[1], st->pos[2], st->a0[0], st->a0[1], st->a0[2], st->a1[0], st->a1[1], st->a1[2]};
    uint32_t mk = 0;
    float mn0 = INFINITY, mn1 = INFINITY, mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
    for (int q = 0; q < PPT; ++q)
        if (first + q < m) {
            mk |= 1u << q;
            const uint32_t p = C.list_values[first + q];
            float u, v;
            plane_uv(fr, C.cv.x[p], C.cv.y[p], C.cv.z[p], u, v);
            mn0 = fminf(mn0, u); mn1 = fminf(mn1, v); mx0 = fmaxf(mx0, u); mx1 = fmaxf(mx1, v);
        }
    ch.masks1[tile * TPB + threadIdx.x] = (uint8_t)mk;
    for (int d = 32; d >= 1; d >>= 1) {
        mn0 = fminf(mn0, __shfl_xor(mn0, d, 64)); mn1 = fminf(mn1, __shfl_xor(mn1, d, 64));
        mx0 = fmaxf(mx0, __shfl_xor(mx0, d, 64)); mx1 = fmaxf(mx1, __shfl_xor(mx1, d, 64));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_bb[0][wave] = mn0; s_bb[1][wave] = mn1; s_bb[2][wave] = mx0; s_bb[3][wave] = mx1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t b0 = tile * TILE;
        const uint32_t tot = b0 >= m ? 0u : min((uint32_t)TILE, m - b0);
        ch.bc1[tile] = tot;
        if (tot)
            agg_add(ch.agg, tile, tot, fminf(fminf(s_bb[0][0], s_bb[0][1]), fminf(s_bb[0][2], s_bb[0][3])),
                    fminf(fminf(s_bb[1][0], s_bb[1][1]), fminf(s_bb[1][2], s_bb[1][3])),
                    fmaxf(fmaxf(s_bb[2][0], s_bb[2][1]), fmaxf(s_bb[2][2], s_bb[2][3])),
                    fmaxf(fmaxf(s_bb[3][0], s_bb[3][1]), fmaxf(s_bb[3][2], s_bb[3][3])));
    }
}

// BitmapExtent (PlanePrimitiveShape.cpp:185-191)
__device__ __forceinline__ bool cc_dims(const float bb[4], uint32_t count, float eps, uint32_t &ue, uint32_t &ve) {
    ue = 2; ve = 2;
    if (count) {
        const float mnu = bb[0], mnv = bb[1], mxu = bb[2], mxv = bb[3];
        const float fu = ceilf((mxu - mnu) / eps), fv = ceilf((mxv - mnv) / eps);
        ue = (fu < 4.0e6f ? (uint32_t)fu : 4000000u) + 1;
        ve = (fv < 4.0e6f ? (uint32_t)fv : 4000000u) + 1;
        if (ue < 2) ue = 2;
        if (ve < 2) ve = 2;
    }
    if ((uint64_t)ue * ve > CC_MAXPIX) { ue = ve = 2; return false; }
    return true;
}

// (2) compact + rasterise: the ordered score list (ascending point position) with its (u, v) parameters, and
//     BuildBitmap (BitmapPrimitiveShape.h:139-150) with InBitmap (PlanePrimitiveShape.cpp:193-199).  The list's length and
//     bounding box come from the aggregates of the mark pass (ChainAgg), the output offset of a tile from the supertile
//     sums + the counts of the tiles of its own supertile: no scan launch, a few dozen loads per workgroup.  The bitmap is
//     all-zero on entry (the labelling kernel clears what it used).  grid (loop_grid(), chains of all clouds): a workgroup
//     owns tiles blockIdx.x, blockIdx.x + gridDim.x, ... OWN_MAX of them per round (the host sizes gridDim.x so that one
//     round usually does: loop_grid()); a workgroup whose tiles hold nothing of the list returns after one round of loads.
static_assert(OWN_MAX * 32 == TPB, "32 lanes per owned tile");

// one workgroup of a chain's own grid: number vx_i of vx (virtual grid: the kernel below)
__device__ __forceinline__ void compact_raster_wg(const RKArgs &A, int k, int g, uint32_t b, uint32_t vx_i, uint32_t vx, uint32_t *s_w,
                                                  uint32_t *s_pre, uint32_t *s_cnt) {
    const RCloudArgsK &C = cloud_args(A, g);
    RState *S = C.st;
    const uint32_t nb = cdiv_d(scan_n(C, S), TILE);   // tiles of what the mark pass scanned (it wrote no count beyond them)
    const ChainPtr ch = chain_of(C, b);
    PlaneState *st = &ch.hdr->st[k];
    const uint32_t tot_early = ch.agg->tot;
    const bool lead = vx_i == 0;
    const float fr[9] = {st->pos[0], st->pos[1], st->pos[2], st->a0[0], st->a0[1], st->a0[2], st->a1[0], st->a1[1], st->a1[2]};
    const float eps = S->bitmap_eps;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int oj = threadIdx.x >> 5, ol = threadIdx.x & 31;     // 32 lanes look after owned tile number oj
    uint32_t *__restrict__ idxA = ch.idxA(k);
    // the whole list: length and bounding box
    const ScanSrc V = scan_src(C, S);
    const uint32_t tot = tot_early;
    float bbv[4] = {INFINITY, INFINITY, -INFINITY, -INFINITY};
    if (tot) { bbv[0] = dec_f(~ch.agg->bb[0]); bbv[1] = dec_f(~ch.agg->bb[1]); bbv[2] = dec_f(ch.agg->bb[2]); bbv[3] = dec_f(ch.agg->bb[3]); }
    uint32_t ue, ve;
    const bool ok = cc_dims(bbv, tot, eps, ue, ve);
    if (lead && threadIdx.x == 0) {
        st->ue = ue; st->ve = ve; st->n_list = tot;
        if (!ok) st->err = 1;
        for (int q = 0; q < 4; ++q) st->bb[q] = bbv[q];
    }
    if (!ok || tot == 0) return;
    const float mnu = bbv[0], mnv = bbv[1];
    for (uint32_t t0 = vx_i; t0 < nb; t0 += vx * OWN_MAX) {
        // the owned tiles' counts, and for the non-empty ones the number of list entries in front of them
        const uint32_t my_tile = t0 + (uint32_t)oj * vx;
        const uint32_t my_cnt = my_tile < nb ? ch.bc1[my_tile] : 0u;
        uint32_t part = 0;
        if (my_cnt) {
            const uint32_t ns = my_tile >> SUP_SHIFT;
            for (uint32_t q = ol; q < ns; q += 32) part += ch.agg->sup[q];
            for (uint32_t q = (ns << SUP_SHIFT) + ol; q < my_tile; q += 32) part += ch.bc1[q];
        }
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);   // stays inside the 32-lane half
        __syncthreads();   // the previous round's readers of the LDS arrays are done
        if (ol == 0) { s_pre[oj] = part; s_cnt[oj] = my_cnt; }
        __syncthreads();
        for (int j = 0; j < OWN_MAX; ++j) {
            const uint32_t tile = t0 + j * vx;
            if (tile >= nb) break;
            if (s_cnt[j] == 0) continue;   // uniform; most tiles of a plane's score list are empty
            const uint32_t m = ch.masks1[tile * TPB + threadIdx.x];
            const uint32_t first = tile * TILE + threadIdx.x * PPT;
            uint32_t pv[PPT];
            float cx[PPT], cy[PPT], cz[PPT];
            // the masks are over positions of what the mark pass scanned: the cloud, the compacted view (the list then holds VIEW
            // positions: the selection pass reads the view's contiguous coordinates, the assign kernel translates the accepted
            // slot's entries into Morton positions through the view's map) or a caller's list (seam S1c: the list gives the points)
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                pv[q] = first + q;
                if ((m & (1u << q)) && C.list_values) pv[q] = C.list_values[first + q];
            }
#pragma unroll
            for (int q = 0; q < PPT; ++q)
                if (m & (1u << q)) { cx[q] = V.x[pv[q]]; cy[q] = V.y[pv[q]]; cz[q] = V.z[pv[q]]; }
            const uint32_t c = __popc(m);
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d, 64);
                if (lane >= d) incl += o;
            }
            __syncthreads();
            if (lane == 63) s_w[wave] = incl;
            __syncthreads();
            uint32_t off = s_pre[j] + incl - c;
            for (int w = 0; w < wave; ++w) off += s_w[w];
#pragma unroll
            for (int q = 0; q < PPT; ++q)
                if (m & (1u << q)) {
                    float u, v;
                    plane_uv(fr, cx[q], cy[q], cz[q], u, v);
                    int bu = (int)floorf((u - mnu) / eps), bv = (int)floorf((v - mnv) / eps);
                    bu = min(max(bu, 0), (int)ue - 1);
                    bv = min(max(bv, 0), (int)ve - 1);
                    const uint32_t px = (uint32_t)bu + (uint32_t)bv * ue;
                    idxA[off] = pv[q];
                    ch.bmp[px] = 1;
                    ++off;
                }
        }
    }
}


__global__ __launch_bounds__(TPB) void k_r_compact_raster(const RKArgs A, int k) {
    __shared__ uint32_t s_w[TPB / 64];
    __shared__ uint32_t s_pre[OWN_MAX], s_cnt[OWN_MAX];
    __shared__ uint32_t s_start[CH_E + 1], s_carry;
    // what every chain needs: nothing (no such chain / converged), its lead workgroup alone (an empty list: the slot's state is
    // still written), or the grid that covers the view's tiles in one round
    const uint32_t total = chain_grid_build(A, s_start, &s_carry, [&](uint32_t g, uint32_t b) -> uint32_t {
        const RCloudArgsK &C = cloud_args(A, g);
        const RState *S = C.st;
        if (b >= S->nc) return 0u;
        const ChainPtr ch = chain_of(C, b);
        if (ch.hdr->st[k].converged) return 0u;
        if (ch.agg->tot == 0u) return 1u;
        const uint32_t nbv = cdiv_d(scan_n(C, S), TILE);
        return max(1u, min(loop_grid(nbv), nbv));
    });
    for (uint32_t vt = blockIdx.x; vt < total; vt += gridDim.x) {
        const uint32_t e = __builtin_amdgcn_readfirstlane(chain_grid_find(s_start, vt));   // (uniform: the table's loads stay scalar)
        __syncthreads();   // the previous virtual workgroup's readers of the LDS arrays are done
        compact_raster_wg(A, k, (int)(e / R_B), e % R_B, vt - s_start[e], s_start[e + 1] - s_start[e], s_w, s_pre, s_cnt);
    }
}

// (3) closing (DilateCross + ErodeCross, ransac/Bitmap.cpp:154-260, 459-570; no wrapping for planes),
// 8-connected labelling (Components, Bitmap.cpp:633-834) and selection of the component with most
// pixels, first in raster order on ties (BitmapPrimitiveShape.cpp:168-173).  One workgroup per chain; bitmaps of
// up to CC_LDS_PIX pixels (every realistic plane at bitmap eps = 2 % of the scene) live in LDS.
// The labelling proper works on a bitmap that lives either in LDS or in global memory; it is force-inlined into both
// branches of k_r_label so that the LDS instance is compiled to ds_* instructions (a pointer selected at run time
// would make every access a flat_* one).
__device__ __forceinline__ void cc_label_body(uint8_t *bmp, uint8_t *tmp, uint32_t *label, uint32_t *sizes, int ue, int ve, int npx,
                                              int do_filter, unsigned long long *s_best_p, PlaneState *st) {
    unsigned long long &s_best = *s_best_p;
    if (do_filter) {
        for (int p = threadIdx.x; p < npx; p += blockDim.x) {
            const int u = p % ue, v = p / ue;
            bool r = bmp[p];
            if (u > 0) r = r || bmp[p - 1];
            if (u < ue - 1) r = r || bmp[p + 1];
            if (v > 0) r = r || bmp[p - ue];
            if (v < ve - 1) r = r || bmp[p + ue];
            tmp[p] = r;
        }
        __syncthreads();
        for (int p = threadIdx.x; p < npx; p += blockDim.x) {
            const int u = p % ue, v = p / ue;
            bool r = tmp[p];
            if (u > 0) r = r && tmp[p - 1];
            if (u < ue - 1) r = r && tmp[p + 1];
            if (v > 0) r = r && tmp[p - ue];
            if (v < ve - 1) r = r && tmp[p + ue];
            bmp[p] = r;
        }
        __syncthreads();
    }
    // 8-connected labelling: (1) every pixel gets the first pixel of its horizontal run as label (one wavefront
    // per row, 64 columns at a time: run head = prefix maximum of the run-start columns), (2) runs are united with
    // the runs they touch in the row above (N, NW, NE) by lock-free union-find on the run heads; the smaller index
    // always becomes the parent, so a component's root is its first pixel in raster order
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
        for (int v = wave; v < ve; v += nwaves) {
            int carry = -1;   // head column of the run that reaches the end of the previous 64-column chunk
            for (int u0 = 0; u0 < ue; u0 += 64) {
                const int u = u0 + lane;
                const int p = v * ue + u;
                const bool fg = u < ue && bmp[p];
                const bool left_fg = (u > 0 && u < ue) ? (bool)bmp[p - 1] : false;
                int h = (fg && !left_fg) ? u : -1;   // a run starts here
                if (lane == 0 && fg && left_fg) h = carry;   // the run continues from the previous chunk
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(h, d, 64);
                    if (lane >= d) h = max(h, o);
                }
                if (u < ue) { label[p] = fg ? (uint32_t)(v * ue + h) : 0xffffffffu; sizes[p] = 0; }
                const int last_h = __shfl(h, 63, 64);
                const bool last_fg = __shfl((int)fg, 63, 64) != 0;
                carry = last_fg ? last_h : -1;
            }
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < npx; p += blockDim.x) {
        if (!bmp[p] || p < ue) continue;
        const int u = p % ue;
        const int nb[3] = {u > 0 ? p - ue - 1 : -1, p - ue, u < ue - 1 ? p - ue + 1 : -1};
        for (int e = 0; e < 3; ++e) {
            if (nb[e] < 0 || !bmp[nb[e]]) continue;
            // N also covers NW / NE whenever N is set (same run above): skip the redundant unions
            if (e != 1 && bmp[p - ue]) continue;
            uint32_t a = label[p], b = label[nb[e]];
            for (;;) {
                while (label[a] != a) a = label[a];
                while (label[b] != b) b = label[b];
                if (a == b) break;
                if (a < b) { const uint32_t t = a; a = b; b = t; }   // a > b: hang a under b
                const uint32_t old = atomicMin(&label[a], b);
                if (old == a) break;
                a = old;
            }
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < npx; p += blockDim.x) {
        if (!bmp[p]) continue;
        uint32_t r = label[p];
        while (label[r] != r) r = label[r];
        label[p] = r;   // races only write the final root or an ancestor: harmless
    }
    __syncthreads();
    for (int p = threadIdx.x; p < npx; p += blockDim.x)
        if (bmp[p]) atomicAdd(&sizes[label[p]], 1u);
    if (threadIdx.x == 0) s_best = 0ull;
    __syncthreads();
    for (int p = threadIdx.x; p < npx; p += blockDim.x)
        if (bmp[p] && label[p] == (uint32_t)p) {
            // max size, then smallest raster-first pixel
            const unsigned long long key = ((unsigned long long)sizes[p] << 32) | (0xffffffffu - (uint32_t)p);
            atomicMax(&s_best, key);
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_best == 0ull) { st->best_root = 0xffffffffu; st->n_fg = 0; }
        else { st->best_root = 0xffffffffu - (uint32_t)(s_best & 0xffffffffu); st->n_fg = (uint32_t)(s_best >> 32); }
    }
}

__global__ __launch_bounds__(1024) void k_r_label(const RKArgs A, int k, int do_filter) {
    const int g = blockIdx.x / R_B;
    const uint32_t b = blockIdx.x % R_B;
    if (g >= (int)A.ng) return;
    const RCloudArgsK &C = cloud_args(A, g);
    if (b >= C.st->nc) return;
    const ChainPtr ch = chain_of(C, b);
    PlaneState *st = &ch.hdr->st[k];
    uint8_t *__restrict__ g_bmp = ch.bmp, *__restrict__ g_tmp = ch.tmp;
    uint32_t *__restrict__ g_label = ch.label, *__restrict__ g_sizes = ch.sizes;
    __shared__ unsigned long long s_best;
    __shared__ uint32_t s_label[CC_LDS_PIX];
    __shared__ uint32_t s_sizes[CC_LDS_PIX];
    __shared__ uint8_t s_bmp[CC_LDS_PIX], s_tmp[CC_LDS_PIX];
    if (st->converged) return;                      // the mark pass skipped this chain: nothing was added
    agg_clear(ch.agg, C.L.nb, threadIdx.x, blockDim.x);   // the compaction pass has consumed the mark pass's aggregates
    if (st->err) return;
    const int ue = (int)st->ue, ve = (int)st->ve, npx = ue * ve;
    if (npx <= CC_LDS_PIX) {
        for (int p = threadIdx.x; p < npx; p += blockDim.x) { s_bmp[p] = g_bmp[p]; g_bmp[p] = 0; }  // also leaves it clean
        __syncthreads();
        cc_label_body(s_bmp, s_tmp, s_label, s_sizes, ue, ve, npx, do_filter, &s_best, st);
        // the selection pass reads the labels from global memory; the bitmap is left all-zero for the next raster
        for (int p = threadIdx.x; p < npx; p += blockDim.x) g_label[p] = s_label[p];
    } else {
        cc_label_body(g_bmp, g_tmp, g_label, g_sizes, ue, ve, npx, do_filter, &s_best, st);
        __syncthreads();
        for (int p = threadIdx.x; p < npx; p += blockDim.x) g_bmp[p] = 0;
    }
}

// Sums of FIT_COLS (= 14) columns over the 64 lanes of a wavefront with 16 exchanges instead of 14 x 6 (a 64-bit exchange is
// two ds_bpermute; the plain butterflies were most of the selection kernel's tail): at every step a lane hands half of the
// columns it still holds to its partner and adds the partner's contribution to the half it keeps (7 + 4 + 2 + 1 exchanges),
// the last two steps add up single values.  Lane l ends with the sum of column 7 b5 + (4 b4 + 2 b3 + b2) (bits of l; the four
// lanes that differ in b1 b0 hold the same value) when 4 b4 + 2 b3 + b2 < 7; the lanes with b4 b3 b2 = 111 hold the padding of
// the 7 -> 4 + 4 split and get col = FIT_COLS (callers store only col < FIT_COLS).  Fixed order: deterministic.
__device__ __forceinline__ double wave_reduce_cols(const double (&a)[FIT_COLS], int lane, int &col) {
    static_assert(FIT_COLS == 14, "the exchange pattern is written for 14 columns");
    const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8, h2 = lane & 4;
    double b[7], c[4], d[2], e;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const double send = h5 ? a[q] : a[q + 7], mine = h5 ? a[q + 7] : a[q];
        b[q] = mine + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double hi = q + 4 < 7 ? b[q + 4 < 7 ? q + 4 : 0] : 0.0;
        const double send = h4 ? b[q] : hi, mine = h4 ? hi : b[q];
        c[q] = mine + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const double send = h3 ? c[q] : c[q + 2], mine = h3 ? c[q + 2] : c[q];
        d[q] = mine + __shfl_xor(send, 8, 64);
    }
    {
        const double send = h2 ? d[0] : d[1], mine = h2 ? d[1] : d[0];
        e = mine + __shfl_xor(send, 4, 64);
    }
    e += __shfl_xor(e, 2, 64);
    e += __shfl_xor(e, 1, 64);
    // column within the half this lane ended up in; 7 = the zero padding of the second step (b4 b3 b2 = 111), which belongs
    // to no column: those lanes report FIT_COLS so that no caller stores their 0.0 over a real sum
    const int local = (h4 ? 4 : 0) + (h3 ? 2 : 0) + (h2 ? 1 : 0);
    col = local < 7 ? (h5 ? 7 : 0) + local : FIT_COLS;
    return e;
}

// (4) selection of the list entries whose pixel belongs to the largest component (4-bit masks per lane over list
// positions + per-tile counts).  The same pass accumulates, over the kept points, the LS-fit moments (12 sums),
// Candidate::WeightedScore (ransac/Candidate.cpp:77-87 with weigh(), ScoreComputer.h:10-16) of the slot's plane and the
// kept count: one row of FIT_COLS doubles per 1024 list positions, reduced by k_r_fit in a fixed order.
__device__ __forceinline__ void select_cc_wg(const RKArgs &A, int k, int g, uint32_t b, uint32_t vx_i, uint32_t vx, uint32_t *s_w,
                                             double (*s)[FIT_COLS]) {
    const RCloudArgsK &C = cloud_args(A, g);
    const ChainPtr ch = chain_of(C, b);
    const PlaneState *st = &ch.hdr->st[k];
    const uint32_t m = st->n_list, best = st->best_root;
    const ScanSrc c = scan_src(C, C.st);       // the list entries are positions in what the mark pass scanned
    const float eps = C.st->eps3;
    const uint32_t *__restrict__ label = ch.label, *__restrict__ idx = ch.idxA(k);
    const float n0 = st->n[0], n1 = st->n[1], n2 = st->n[2], dist = st->dist;
    // the pixel of a list entry is recomputed from its coordinates exactly as the rasterisation computed it (same fp32
    // operations on the same values) instead of being written there and read here: 8 bytes of traffic per entry less
    const float fr[9] = {st->pos[0], st->pos[1], st->pos[2], st->a0[0], st->a0[1], st->a0[2], st->a1[0], st->a1[1], st->a1[2]};
    const float mnu = st->bb[0], mnv = st->bb[1], beps = C.st->bitmap_eps;
    const int ue = (int)st->ue, ve = (int)st->ve;
    // rows of 1024 list positions: workgroup vx_i of the chain's vx takes rows vx_i, vx_i + vx, ... (k_r_fit reads
    // ceil(m / 1024) rows)
    for (uint32_t row = vx_i; row * 1024u < m; row += vx) {
    const uint32_t base = row * 1024 + threadIdx.x * 4;
    uint32_t mk = 0, cnt = 0;
    double a[FIT_COLS];
#pragma unroll
    for (int q = 0; q < FIT_COLS; ++q) a[q] = 0.0;
    // two rounds of independent loads (list entries, then everything they point to) instead of a chain of four:
    // nearly every listed point is kept, so the coordinates are fetched before the label test is known
    uint32_t pi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pi[q] = idx[min(base + q, m - 1)];
    uint32_t lb[4];
    float fx[4], fy[4], fz[4], gx[4], gy[4], gz[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t p = pi[q];
        fx[q] = c.x[p]; fy[q] = c.y[p]; fz[q] = c.z[p];
        gx[q] = c.nx[p]; gy[q] = c.ny[p]; gz[q] = c.nz[p];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float u, v;
        plane_uv(fr, fx[q], fy[q], fz[q], u, v);
        int bu = (int)floorf((u - mnu) / beps), bv = (int)floorf((v - mnv) / beps);
        bu = min(max(bu, 0), ue - 1);
        bv = min(max(bv, 0), ve - 1);
        lb[q] = label[(uint32_t)bu + (uint32_t)bv * (uint32_t)ue];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool in = base + q < m && best != 0xffffffffu && lb[q] == best;
        mk |= (in ? 1u : 0u) << q;
        cnt += (uint32_t)__popcll(__ballot(in));
        if (in) {
            const double x = fx[q], y = fy[q], z = fz[q];
            a[0] += x; a[1] += y; a[2] += z;
            a[3] += x * x; a[4] += x * y; a[5] += x * z; a[6] += y * y; a[7] += y * z; a[8] += z * z;
            a[9] += gx[q]; a[10] += gy[q]; a[11] += gz[q];
            float d = n0 * fx[q];
            d += n1 * fy[q];
            d += n2 * fz[q];
            d = fabsf(dist - d);
            a[12] += (double)expf(-d * d / (2.f / 9.f * eps * eps));
            a[13] += 1.0;
        }
    }
    ch.masks2(k)[row * TPB + threadIdx.x] = (uint8_t)mk;
    int col;
    const double colsum = wave_reduce_cols(a, threadIdx.x & 63, col);
    __syncthreads();   // the previous row's readers are done
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
    if ((threadIdx.x & 3) == 0 && col < FIT_COLS) s[threadIdx.x >> 6][col] = colsum;
    __syncthreads();
    if (threadIdx.x == 0) ch.bc2(k)[row] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (threadIdx.x < FIT_COLS)
        ch.part[(size_t)row * FIT_COLS + threadIdx.x] = (s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]);
    }
}


__global__ __launch_bounds__(TPB) void k_r_select_cc(const RKArgs A, int k) {
    __shared__ uint32_t s_w[4];
    __shared__ double s[4][FIT_COLS];
    __shared__ uint32_t s_start[CH_E + 1], s_carry;
    // a chain's grid: one workgroup per row of 1024 list positions, at most loop_grid(tiles of its cloud) of them
    const uint32_t total = chain_grid_build(A, s_start, &s_carry, [&](uint32_t g, uint32_t b) -> uint32_t {
        const RCloudArgsK &C = cloud_args(A, g);
        if (b >= C.st->nc) return 0u;
        const PlaneState *st = &chain_of(C, b).hdr->st[k];
        if (st->converged | st->err) return 0u;
        return min(loop_grid(C.L.nb), cdiv_d(st->n_list, 1024u));
    });
    for (uint32_t vt = blockIdx.x; vt < total; vt += gridDim.x) {
        const uint32_t e = __builtin_amdgcn_readfirstlane(chain_grid_find(s_start, vt));
        __syncthreads();
        select_cc_wg(A, k, (int)(e / R_B), e % R_B, vt - s_start[e], s_start[e + 1] - s_start[e], s_w, s);
    }
}

// (5) LS refit (PlanePrimitiveShape::LSFit -> Plane::LeastSquaresFit, ransac/Plane.cpp:169-176,
// Plane.h:65-74: mean + covariance about the mean + smallest-|eigenvalue| eigenvector).
// Accumulated in fp64 with a fixed reduction tree (deterministic); the reference accumulates in
// fp32 sequentially, which is the noisier of the two.
__device__ void jacobi3_d(double a[3][3], double d[3], double v[3][3]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) v[i][j] = i == j;
    const double scale = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off <= 1e-24 * scale) break;   // far below fp64 resolution of the eigenvectors (quadratic convergence)
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(a[p][q]) < 1e-300) continue;
                const double theta = (a[q][q] - a[p][p]) / (2 * a[p][q]);
                const double t = (theta >= 0 ? 1 : -1) / (fabs(theta) + sqrt(theta * theta + 1));
                const double cs = 1 / sqrt(t * t + 1), sn = t * cs;
                for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = cs * akp - sn * akq; a[k][q] = sn * akp + cs * akq; }
                for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = cs * apk - sn * aqk; a[q][k] = sn * apk + cs * aqk; }
                for (int k = 0; k < 3; ++k) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = cs * vkp - sn * vkq; v[k][q] = sn * vkp + cs * vkq; }
            }
    }
    for (int i = 0; i < 3; ++i) d[i] = a[i][i];
}

// Sums the per-tile rows of slot k's selection pass (fixed tree => deterministic): weighted score, normal sum and
// kept count of slot k; for k < 3 additionally the LS plane of the kept points = the NEXT slot's plane.  When that
// plane is bitwise equal to slot k's the chain has converged: every later slot would reproduce slot k's results, so
// they are flagged, copy them and their kernels return immediately.  One workgroup of 256 lanes per chain.
__global__ __launch_bounds__(256) void k_r_fit(const RKArgs A, int k) {
    __shared__ double s_red[4][FIT_COLS];
    const int g = blockIdx.x / R_B;
    const uint32_t b = blockIdx.x % R_B;
    if (g >= (int)A.ng) return;
    const RCloudArgsK &C = cloud_args(A, g);
    if (b >= C.st->nc) return;
    const ChainPtr ch = chain_of(C, b);
    PlaneState *cur = &ch.hdr->st[k];
    PlaneState *nxt = k < 3 ? &ch.hdr->st[k + 1] : nullptr;
    if (cur->converged) {   // repeat the predecessor
        if (threadIdx.x == 0) {
            const PlaneState *prev = &ch.hdr->st[k - 1];
            cur->wscore = prev->wscore; cur->n_kept = prev->n_kept; cur->n_list = prev->n_list; cur->ue = prev->ue; cur->ve = prev->ve;
            cur->nsum[0] = prev->nsum[0]; cur->nsum[1] = prev->nsum[1]; cur->nsum[2] = prev->nsum[2];
            if (nxt) { *nxt = *cur; nxt->converged = 1; nxt->err = 0; }
        }
        return;
    }
    if (cur->err) {   // bitmap too large: nothing was selected
        if (threadIdx.x == 0) { cur->wscore = 0; cur->n_kept = 0; if (nxt) { *nxt = *cur; nxt->converged = 0; nxt->err = 2; } }
        return;
    }
    const double *__restrict__ part = ch.part;
    const uint32_t rows = (cur->n_list + 1023u) / 1024u;
    double a[FIT_COLS];
    for (int q = 0; q < FIT_COLS; ++q) a[q] = 0.0;
    for (uint32_t r = threadIdx.x; r < rows; r += blockDim.x)
        for (int q = 0; q < FIT_COLS; ++q) a[q] += part[(size_t)r * FIT_COLS + q];
    {
        int col;
        const double colsum = wave_reduce_cols(a, threadIdx.x & 63, col);
        if ((threadIdx.x & 3) == 0 && col < FIT_COLS) s_red[threadIdx.x >> 6][col] = colsum;
    }
    __syncthreads();
    if (threadIdx.x) return;
    for (int q = 0; q < FIT_COLS; ++q) a[q] = (s_red[0][q] + s_red[1][q]) + (s_red[2][q] + s_red[3][q]);
    cur->wscore = a[12];
    cur->nsum[0] = (float)a[9]; cur->nsum[1] = (float)a[10]; cur->nsum[2] = (float)a[11];
    const double m = a[13];
    cur->n_kept = (uint32_t)m;
    if (!nxt) return;
    // The reference's refit loop ends at the first refit whose weighted score does not beat its predecessor's
    // (RansacShapeDetector.cpp:633-655; k_r_decide replays exactly that on the slots' results): no later slot of this chain is
    // ever looked at, so they are flagged like the slots of a converged chain and their kernels return at once.  (On the bench's
    // 64 pairs a quarter of the chains stop at refit 1 or 2.)
    if (k >= 1 && !(a[12] > ch.hdr->st[k - 1].wscore)) { *nxt = *cur; nxt->converged = 1; nxt->err = 0; return; }
    nxt->err = 0;
    nxt->converged = 0;
    nxt->n_list = nxt->n_kept = 0;
    nxt->wscore = 0.0;
    if (m < 3) {   // LSFit impossible: the slot gets a plane nothing is compatible with
        nxt->err = 2;
        nxt->n[0] = nxt->n[1] = nxt->n[2] = 0.f; nxt->dist = __int_as_float(0x7fc00000);
        return;
    }
    const double mx = a[0] / m, my = a[1] / m, mz = a[2] / m;
    double cv[3][3];
    cv[0][0] = a[3] / m - mx * mx; cv[0][1] = a[4] / m - mx * my; cv[0][2] = a[5] / m - mx * mz;
    cv[1][1] = a[6] / m - my * my; cv[1][2] = a[7] / m - my * mz; cv[2][2] = a[8] / m - mz * mz;
    cv[1][0] = cv[0][1]; cv[2][0] = cv[0][2]; cv[2][1] = cv[1][2];
    double ev[3], vec[3][3];
    jacobi3_d(cv, ev, vec);
    int mi = 0;
    for (int i = 1; i < 3; ++i) if (fabs(ev[i]) < fabs(ev[mi])) mi = i;
    const float n[3] = {(float)vec[0][mi], (float)vec[1][mi], (float)vec[2][mi]};
    nxt->n[0] = n[0]; nxt->n[1] = n[1]; nxt->n[2] = n[2];
    nxt->pos[0] = (float)mx; nxt->pos[1] = (float)my; nxt->pos[2] = (float)mz;
    float dist = nxt->pos[0] * n[0];   // Plane(p1, normal): m_dist = m_pos.dot(m_normal) (Plane.cpp:21-26)
    dist += nxt->pos[1] * n[1];
    dist += nxt->pos[2] * n[2];
    nxt->dist = dist;
    hcs_axes(nxt->n, nxt->a0, nxt->a1);
    nxt->converged = (n[0] == cur->n[0] && n[1] == cur->n[1] && n[2] == cur->n[2] && dist == cur->dist &&
                      nxt->pos[0] == cur->pos[0] && nxt->pos[1] == cur->pos[1] && nxt->pos[2] == cur->pos[2]) ? 1u : 0u;
}

// ------------------------------------------------------------------------------------------------
// After the four slots: replay of the reference's refit loop on the four results of every chain, removal
// bookkeeping (RansacShapeDetector.cpp:666-675), output planes (plane_extraction.cpp:134-149), the pool without
// the batch, and what the next iteration does.  One lane per cloud does the sequential part.
constexpr int DEC_T = 256;   // lanes of k_r_decide: the staging loads and the deferral tests use all of them, the rest wave 0
__global__ __launch_bounds__(DEC_T) void k_r_decide(const RKArgs A) {
    const RCloudArgsK &C = cloud_args(A, blockIdx.x);
    RState *S = C.st;
    RResult *R = C.res;
    if (!S->active) return;
    // Everything the sequential part touches is fetched by all lanes at once (one lane alone pays a round trip to L2 per
    // field, and every store to the state forces the next load to be re-issued): the chains' results, the pool, the scalars.
    __shared__ PlaneState s_st[R_B][4];
    __shared__ float4 s_pool_pl[R_TOP], s_pool_pos[R_TOP];
    __shared__ uint32_t s_batch[R_B], s_keep_pos[R_TOP];
    const uint32_t done_in = S->done, nc_all = done_in ? 0u : S->nc, np = S->npool;
    const uint32_t min_support = S->min_support, orient = S->orient;
    uint32_t n_remaining = S->n_remaining, n_acc = S->n_acc, out_off = S->out_off, n_accepts = S->n_accepts, err = S->err, done = done_in;
    float drawn = S->drawn;
    {
        constexpr int WORDS = sizeof(PlaneState) / 4;
        for (uint32_t i = threadIdx.x; i < nc_all * 4 * WORDS; i += DEC_T) {
            const uint32_t b = i / (4 * WORDS), r = i % (4 * WORDS);
            reinterpret_cast<uint32_t *>(&s_st[b][0])[r] = reinterpret_cast<const uint32_t *>(&chain_of(C, b).hdr->st[0])[r];
        }
        if (nc_all) {
            if (threadIdx.x < np && threadIdx.x < R_TOP) { s_pool_pl[threadIdx.x] = S->pool_pl[threadIdx.x]; s_pool_pos[threadIdx.x] = S->pool_pos[threadIdx.x]; }
            if (threadIdx.x < nc_all) s_batch[threadIdx.x] = S->batch_idx[threadIdx.x];
        }
    }
    __syncthreads();
    // (1) per chain: the reference's refit loop replayed on the four slots' results (RansacShapeDetector.cpp:633-655)
    __shared__ int s_final[R_B], s_stop[R_B];
    __shared__ uint32_t s_defer[R_B];
    if (threadIdx.x < nc_all) {
        const PlaneState *st = s_st[threadIdx.x];
        int final_slot = 0, stop = 4;
        double newScore = st[0].wscore;
        for (int fittingIter = 1; fittingIter <= 3; ++fittingIter) {
            const double oldScore = newScore;
            if (st[fittingIter - 1].n_kept < 3 || st[fittingIter].err) { stop = fittingIter; break; }   // LSFit impossible
            newScore = st[fittingIter].wscore;
            const uint32_t newSize = st[fittingIter].n_kept;
            if (newScore > oldScore && newSize > min_support) final_slot = fittingIter;  // clone.Clone(&candidates.back())
            if (!(newScore > oldScore)) { stop = fittingIter; break; }
        }
        s_final[threadIdx.x] = final_slot;
        s_stop[threadIdx.x] = stop;
        s_defer[threadIdx.x] = 0;
    }
    __syncthreads();
    // (2) Accepting the chains of a batch together equals accepting them one by one, best first, only if no chain's score
    // lists touch what a better candidate takes.  k_r_select made sure of that for the HYPOTHESES; the LS refits move the
    // planes, and a refit that starts on a small face can tilt into a large neighbouring surface and end up ON it (seen on
    // the 100-plane hall of BASELINE configs[4]: a box top 1.5 m under the ceiling, 20 degrees tilted, took the whole
    // ceiling in its third refit -- together with the ceiling's own chain).  So the same provable-disjointness test is
    // repeated on the planes the slots really used: chain b is DEFERRED (stays in the pool, nothing of it is removed now;
    // the next iteration re-scores it without the points accepted in this one) when any plane it scored with conflicts
    // with any plane a better chain of the batch scored with, or -- beyond its hypothesis -- with a better pool candidate
    // that is still waiting.  The best chain is never deferred.
    {
        const float eps = S->eps, cos_t = conflict_gate(S->cos_t);
        float bbmin[3], bbmax[3];
        for (int q = 0; q < 3; ++q) { bbmin[q] = S->bbmin[q]; bbmax[q] = S->bbmax[q]; }
        const uint32_t n_other = 4 * R_B + R_TOP;          // per (chain b, slot sb): 4 slots of every chain, then the pool
        for (uint32_t t = threadIdx.x; t < nc_all * 4 * n_other; t += DEC_T) {
            const uint32_t b = t / (4 * n_other), r = t % (4 * n_other), sb = r / n_other, o = r % n_other;
            if (b == 0 || (int)sb > min(s_stop[b], 3) || s_st[b][sb].err) continue;
            const float4 pb = make_float4(s_st[b][sb].n[0], s_st[b][sb].n[1], s_st[b][sb].n[2], s_st[b][sb].dist);
            float4 po;
            if (o < 4 * R_B) {
                const uint32_t a = o >> 2, sa = o & 3;
                if (a >= b || (int)sa > min(s_stop[a], 3) || s_st[a][sa].err) continue;
                po = make_float4(s_st[a][sa].n[0], s_st[a][sa].n[1], s_st[a][sa].n[2], s_st[a][sa].dist);
            } else {
                const uint32_t q = o - 4 * R_B;
                if (sb == 0 || q >= np || q >= s_batch[b]) continue;     // better candidates only; hypotheses were checked before
                bool in_batch = false;
                for (uint32_t x = 0; x < nc_all; ++x) in_batch = in_batch || s_batch[x] == q;
                if (in_batch) continue;
                po = s_pool_pl[q];
            }
            if (!conflict_free(pb, po, eps, cos_t, bbmin, bbmax)) s_defer[b] = 1;
        }
    }
    __syncthreads();
    uint32_t n_aj = 0;
    if (threadIdx.x == 0) {
        uint32_t n_final[4] = {0, 0, 0, 0}, n_stop[5] = {0, 0, 0, 0, 0};
        for (uint32_t b = 0; b < nc_all; ++b) {
            const PlaneState *st = s_st[b];
            if (st[0].err == 1) { err = 1; done = 1; break; }   // connected-component bitmap too large
            if (s_defer[b]) { S->n_deferred += 1; continue; }
            n_accepts += 1;
            const int final_slot = s_final[b], stop = s_stop[b];
            n_final[final_slot] += 1;
            n_stop[stop] += 1;
            const PlaneState &cs = st[final_slot];
            const uint32_t cand_size = cs.n_kept;
            if (cand_size == 0) continue;
            if (n_acc >= R_MAXP) { err = 3; done = 1; break; }
            const uint32_t id = n_acc;
            S->aj_chain[n_aj] = b; S->aj_slot[n_aj] = (uint32_t)final_slot; S->aj_id[n_aj] = (int32_t)id;
            uint32_t aj_out = 0xffffffffu;
            const float frac = 1.f - (cand_size / float(n_remaining));
            drawn = frac * frac * frac * drawn;    // std::pow(1.f - |S| / n, 3.f) * drawnCandidates
            n_remaining -= min(n_remaining, cand_size);
            // plane_extraction.cpp:134-149: shapes below min_support are skipped, d = -n.p with n re-normalised
            uint32_t support = 0;
            if (cand_size >= min_support) {
                float nn[3] = {cs.n[0], cs.n[1], cs.n[2]};
                float l = nn[0] * nn[0];
                l += nn[1] * nn[1];
                l += nn[2] * nn[2];
                l = sqrtf(l);
                if (l > 0) { nn[0] /= l; nn[1] /= l; nn[2] /= l; }
                float d = -(nn[0] * cs.pos[0] + nn[1] * cs.pos[1] + nn[2] * cs.pos[2]);
                if (orient) {  // mean inlier normal (the intent of plane_extraction.cpp:43-58)
                    if (cs.nsum[0] * nn[0] + cs.nsum[1] * nn[1] + cs.nsum[2] * nn[2] < 0) { nn[0] = -nn[0]; nn[1] = -nn[1]; nn[2] = -nn[2]; d = -d; }
                }
                S->acc_coef[id][0] = nn[0]; S->acc_coef[id][1] = nn[1]; S->acc_coef[id][2] = nn[2]; S->acc_coef[id][3] = d;
                support = cand_size;
                aj_out = out_off;
            }
            S->acc_support[id] = support;
            S->acc_dbg[id] = (S->it << 16) | (b << 8) | (uint32_t)final_slot;
            S->acc_offset[id] = out_off;
            S->aj_out[n_aj] = aj_out;
            out_off += support;
            n_acc = id + 1;
            ++n_aj;
        }
        for (int q = 0; q < 4; ++q) if (n_final[q]) S->n_final[q] += n_final[q];
        for (int q = 0; q < 5; ++q) if (n_stop[q]) S->n_stop[q] += n_stop[q];
        S->aj_n = n_aj;
        S->view_dirty = (n_aj && C.view[0]) ? 1u : 0u;   // points will be taken: the scan view is rebuilt behind k_r_assign
        S->n_accepts = n_accepts; S->n_remaining = n_remaining; S->n_acc = n_acc; S->out_off = out_off; S->drawn = drawn; S->err = err;
    }
    __syncthreads();
    // the pool without the batch (batch indices are ascending), order kept: every lane moves its own entry
    if (nc_all) {
        bool in_batch = false;
        uint32_t before = 0;
        for (uint32_t q = 0; q < nc_all; ++q) {   // the batch leaves the pool, except the chains deferred above
            const bool gone = !s_defer[q];
            in_batch = in_batch || (gone && s_batch[q] == threadIdx.x);
            before += (gone && s_batch[q] < threadIdx.x) ? 1u : 0u;
        }
        const bool keep = threadIdx.x < np && !in_batch;   // np <= R_TOP < 64: wave 0 holds the whole pool
        if (keep) { S->pool_pl[threadIdx.x - before] = s_pool_pl[threadIdx.x]; S->pool_pos[threadIdx.x - before] = s_pool_pos[threadIdx.x]; }
        if (threadIdx.x == 0) s_keep_pos[0] = 0;
        const uint32_t w = (uint32_t)__popcll(__ballot(keep));   // wave 0's ballot is the one lane 0 uses
        if (threadIdx.x == 0) {
            // (n_remaining, done, err were updated by this lane above; visible here in its own registers)
            uint32_t npool = w;
            S->nc = 0;
            if (!done) {
                if (n_remaining < min_support) npool = 0;
                S->npool = npool;
                if (npool == 0) next_round_or_stop(S);
                else if (S->topup) { if (S->round >= R_MAX_ROUNDS) S->done = 1; else S->sampling = 1; }
            } else { S->npool = npool; S->done = 1; }
        }
    } else if (threadIdx.x == 0 && done && !done_in) S->done = 1;
    if (threadIdx.x == 0) S->it += 1;
    __syncthreads();
    // The host-mapped block is written once, when the call ends (stores over PCIe are slow: all 64 lanes share them);
    // until then the host only needs the iteration count.
    const uint32_t done_now = S->done;
    if (done_now) {
        const uint32_t na = S->n_acc;
        for (uint32_t i = threadIdx.x; i < 4 * na; i += DEC_T) (&R->coef[0][0])[i] = (&S->acc_coef[0][0])[i];
        for (uint32_t i = threadIdx.x; i < na; i += DEC_T) { R->support[i] = S->acc_support[i]; R->offset[i] = S->acc_offset[i]; R->dbg[i] = S->acc_dbg[i]; }
        if (threadIdx.x == 0) {
            R->n_acc = na; R->out_off = S->out_off; R->err = S->err; R->remaining = S->n_remaining;
            R->n_rounds = S->n_rounds; R->n_rescores = S->n_rescores[0] + S->n_rescores[1]; R->n_batches = S->n_batches; R->n_accepts = S->n_accepts;
            R->n_mark_launches = S->n_mark_launches; R->n_mark_chains = S->n_mark_chains; R->n_deferred = S->n_deferred;
            for (int q = 0; q < 4; ++q) R->n_final[q] = S->n_final[q];
            for (int q = 0; q < 5; ++q) R->n_stop[q] = S->n_stop[q];
        }
    }
    // the result block must be visible before the flag says "done"; an iteration count alone orders nothing (the host only
    // uses it to decide when to queue the next iteration), and a system-scope fence is an L2 write-back on this part
    if (done_now) __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        *reinterpret_cast<volatile uint32_t *>(&R->flag) = (S->it & 0xffffffu) | ((S->gen & 0x7fu) << 24) | (done_now ? 0x80000000u : 0u);
}


// Point removal + output index lists of the accepted candidates: the chosen slot's list entries that belong to the
// largest component, in list order (ordered compaction of the selection masks; offsets from the per-row counts as in
// k_r_compact_raster).  Virtual grid over the jobs of all clouds.
__device__ __forceinline__ void assign_wg(const RKArgs &A, int g, uint32_t j, uint32_t vx_i, uint32_t vx, uint32_t (*s_pre)[TPB / 64],
                                          uint32_t *s_w) {
    const RCloudArgsK &C = cloud_args(A, g);
    const RState *S = C.st;
    const int k = (int)S->aj_slot[j];
    const ChainPtr ch = chain_of(C, S->aj_chain[j]);
    const uint32_t m = ch.hdr->st[k].n_list;
    const uint32_t rows = (m + 1023u) / 1024u;
    const int32_t id = S->aj_id[j];
    const uint32_t out_off = S->aj_out[j];
    const uint32_t *__restrict__ bc = ch.bc2(k);
    const uint8_t *__restrict__ masks = ch.masks2(k);
    const uint32_t *__restrict__ idx = ch.idxA(k);
    int32_t *__restrict__ out = out_off == 0xffffffffu ? nullptr : C.out_idx + out_off;
    const uint32_t *__restrict__ vmap = scan_src(C, S).map;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t r0 = vx_i; r0 < rows; r0 += vx * OWN_MAX) {
        uint32_t pre[OWN_MAX];
#pragma unroll
        for (int q = 0; q < OWN_MAX; ++q) pre[q] = 0;
        for (uint32_t q = threadIdx.x; q < rows; q += TPB) {
            const uint32_t c = bc[q];
#pragma unroll
            for (int o = 0; o < OWN_MAX; ++o) pre[o] += q < r0 + o * vx ? c : 0u;
        }
        for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
            for (int o = 0; o < OWN_MAX; ++o) pre[o] += __shfl_xor(pre[o], d, 64);
        }
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int o = 0; o < OWN_MAX; ++o) s_pre[o][wave] = pre[o];
        }
        __syncthreads();
        for (int o = 0; o < OWN_MAX; ++o) {
            const uint32_t row = r0 + o * vx;
            if (row >= rows) break;
            if (bc[row] == 0) continue;   // uniform
            const uint32_t mk = masks[row * TPB + threadIdx.x];
            const uint32_t base = row * 1024 + threadIdx.x * 4;
            uint32_t p[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) p[q] = (mk & (1u << q)) ? idx[base + q] : 0u;
            if (vmap)   // the lists hold positions of the scan view: here they become Morton positions
                for (int q = 0; q < 4; ++q) if (mk & (1u << q)) p[q] = vmap[p[q]];
            const uint32_t c = __popc(mk);
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t v = __shfl_up(incl, d, 64);
                if (lane >= d) incl += v;
            }
            __syncthreads();
            if (lane == 63) s_w[wave] = incl;
            __syncthreads();
            uint32_t off = (s_pre[o][0] + s_pre[o][1] + s_pre[o][2] + s_pre[o][3]) + incl - c;
            for (int w = 0; w < wave; ++w) off += s_w[w];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (mk & (1u << q)) {
                    if (C.taken) C.taken[p[q]] = 1;
                    if (C.sub_assigned && p[q] % C.sub_stride == 0 && p[q] / C.sub_stride < C.n_sub) C.sub_assigned[p[q] / C.sub_stride] = id;
                    if (out) {
                        out[off] = (int32_t)(C.orig ? C.orig[p[q]] : p[q]);
                        if (C.out_pos) C.out_pos[out_off + off] = p[q];
                    }
                    ++off;
                }
        }
    }
}


__global__ __launch_bounds__(TPB) void k_r_assign(const RKArgs A) {
    __shared__ uint32_t s_pre[OWN_MAX][TPB / 64], s_w[TPB / 64];
    __shared__ uint32_t s_start[CH_E + 1], s_carry;
    // a removal job's grid: one workgroup per row of its slot's list, at most loop_grid(tiles of its cloud)
    const uint32_t total = chain_grid_build(A, s_start, &s_carry, [&](uint32_t g, uint32_t j) -> uint32_t {
        const RCloudArgsK &C = cloud_args(A, g);
        const RState *S = C.st;
        if (j >= S->aj_n) return 0u;
        const uint32_t m = chain_of(C, S->aj_chain[j]).hdr->st[S->aj_slot[j]].n_list;
        return min(loop_grid(C.L.nb), cdiv_d(m, 1024u));
    });
    for (uint32_t vt = blockIdx.x; vt < total; vt += gridDim.x) {
        const uint32_t e = __builtin_amdgcn_readfirstlane(chain_grid_find(s_start, vt));
        __syncthreads();
        assign_wg(A, (int)(e / R_B), e % R_B, vt - s_start[e], s_start[e + 1] - s_start[e], s_pre, s_w);
    }
}

// ------------------------------------------------------------------------------------------------
// The scan view (RState::view_sel): behind an iteration that took points, the points nobody has taken are copied, in order,
// into the other view buffer -- coordinates, normals and Morton positions -- so that the next iteration's five full passes (four
// mark passes, one re-score) read ~30 %, then ~15 %, ... of the cloud instead of all of it with most points masked out
// (r3 / verdict: 13.5 of 18 mark launches per registration re-scanned 2 x 28 MB).  Same points in the same order: the masks,
// lists and counts of every pass are unchanged, bit for bit.  Three launches: count the survivors per view tile (+ supertile
// sums by atomics), compact (a tile's offset from the supertile sums + the counts of its own supertile, as the chains'
// compaction does), commit (flip the view, clear the sums).  Nothing to do (view_dirty = 0): the workgroups return at once.
__global__ __launch_bounds__(TPB) void k_view_count(const RKArgs A) {
    __shared__ uint32_t s_w[TPB / 64];
    __shared__ uint32_t s_start[R_G + 1];
    const uint32_t total = scan_grid_build(A, s_start, [&](uint32_t g) -> uint32_t {   // virtual grid: the tiles of the views that are rebuilt
        const RCloudArgsK &C = cloud_args(A, g);
        const RState *S = C.st;
        return (!C.view[0] || !S->view_dirty || S->done) ? 0u : cdiv_d(scan_n(C, S), TILE);
    });
    for (uint32_t vt = blockIdx.x; vt < total; vt += gridDim.x) {
    const int g = __builtin_amdgcn_readfirstlane(scan_grid_find(s_start, vt));
    const uint32_t tile = vt - s_start[g];
    __syncthreads();
    const RCloudArgsK &C = cloud_args(A, g);
    RState *S = C.st;
    const ScanSrc V = scan_src(C, S);
    const uint32_t first = tile * TILE + threadIdx.x * PPT;
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < PPT; ++q)
        if (first + q < V.n) {
            const uint32_t p = V.map ? V.map[first + q] : first + q;
            c += C.taken[p] ? 0u : 1u;
        }
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        C.view_cnt[tile] = tot;
        // (the view's new length is the sum of the supertile sums, taken by k_view_commit: an atomic per tile on ONE word per cloud
        //  -- ~1000 of them in the first iteration -- was most of this kernel's duration)
        if (tot) atomicAdd(&C.view_sup[tile >> SUP_SHIFT], tot);
    }
    }
}

__global__ __launch_bounds__(TPB) void k_view_compact(const RKArgs A) {
    __shared__ uint32_t s_w[TPB / 64], s_pre;
    __shared__ uint32_t s_start[R_G + 1];
    const uint32_t total = scan_grid_build(A, s_start, [&](uint32_t g) -> uint32_t {
        const RCloudArgsK &C = cloud_args(A, g);
        const RState *S = C.st;
        return (!C.view[0] || !S->view_dirty || S->done) ? 0u : cdiv_d(scan_n(C, S), TILE);
    });
    for (uint32_t vt = blockIdx.x; vt < total; vt += gridDim.x) {
    const int g = __builtin_amdgcn_readfirstlane(scan_grid_find(s_start, vt));
    const uint32_t tile = vt - s_start[g];
    __syncthreads();
    const RCloudArgsK &C = cloud_args(A, g);
    const RState *S = C.st;
    const ScanSrc V = scan_src(C, S);
    if (C.view_cnt[tile] == 0) continue;   // uniform
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // entries in front of this tile: complete supertiles + the tiles of its own supertile before it
    uint32_t part = 0;
    if (wave == 0) {
        const uint32_t ns = tile >> SUP_SHIFT;
        for (uint32_t q = lane; q < ns; q += 64) part += C.view_sup[q];
        for (uint32_t q = (ns << SUP_SHIFT) + lane; q < tile; q += 64) part += C.view_cnt[q];
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        if (lane == 0) s_pre = part;
    }
    const uint32_t first = tile * TILE + threadIdx.x * PPT;
    // the lane's four points of all six arrays as 16-byte loads, issued BEFORE anything is known about them: the chain
    // map -> taken -> prefix -> coordinates was four dependent round trips (the wavefronts of this kernel waited 94 % of their
    // time, profiles/r6_pmc_sq.csv); now the coordinates travel beside the map and the flags (the arrays are padded to whole
    // 16-byte groups, as the scan kernels rely on)
    static_assert(PPT == 4, "one float4 per lane and array");
    float4 cx4 = make_float4(0.f, 0.f, 0.f, 0.f), cy4 = cx4, cz4 = cx4, nx4 = cx4, ny4 = cx4, nz4 = cx4;
    if (first < V.n) {
        cx4 = *reinterpret_cast<const float4 *>(V.x + first); cy4 = *reinterpret_cast<const float4 *>(V.y + first);
        cz4 = *reinterpret_cast<const float4 *>(V.z + first); nx4 = *reinterpret_cast<const float4 *>(V.nx + first);
        ny4 = *reinterpret_cast<const float4 *>(V.ny + first); nz4 = *reinterpret_cast<const float4 *>(V.nz + first);
    }
    uint32_t pos[PPT], m = 0;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        pos[q] = 0;
        if (first + q < V.n) {
            pos[q] = V.map ? V.map[first + q] : first + q;
            m |= (C.taken[pos[q]] ? 0u : 1u) << q;
        }
    }
    const uint32_t c = __popc(m);
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t off = s_pre + incl - c;
    for (int w = 0; w < wave; ++w) off += s_w[w];
    const uint32_t dst_sel = S->view_sel == 1u ? 1u : 0u;     // from the cloud or from A into ... A (0) unless A is the source
    float *b = C.view[dst_sel];
    uint32_t *dmap = C.view_map[dst_sel];
    const size_t pitch = ((size_t)C.cv.n + 3) & ~(size_t)3;
    const float ax[4] = {cx4.x, cx4.y, cx4.z, cx4.w}, ay[4] = {cy4.x, cy4.y, cy4.z, cy4.w}, az[4] = {cz4.x, cz4.y, cz4.z, cz4.w};
    const float bx[4] = {nx4.x, nx4.y, nx4.z, nx4.w}, by[4] = {ny4.x, ny4.y, ny4.z, ny4.w}, bz[4] = {nz4.x, nz4.y, nz4.z, nz4.w};
#pragma unroll
    for (int q = 0; q < PPT; ++q)
        if (m & (1u << q)) {
            b[off] = ax[q]; b[pitch + off] = ay[q]; b[2 * pitch + off] = az[q];
            b[3 * pitch + off] = bx[q]; b[4 * pitch + off] = by[q]; b[5 * pitch + off] = bz[q];
            dmap[off] = pos[q];
            ++off;
        }
    }
}

__global__ __launch_bounds__(256) void k_view_commit(const RKArgs A) {
    const RCloudArgsK &C = cloud_args(A, blockIdx.x);
    RState *S = C.st;
    if (!C.view[0] || !S->view_dirty || S->done) { if (C.view[0] && threadIdx.x == 0 && S->view_dirty) S->view_dirty = 0; return; }
    __shared__ uint32_t s_sum[256 / 64];
    const uint32_t ntiles = (scan_n(C, S) + TILE - 1) / TILE;
    uint32_t sum = 0;
    for (uint32_t q = threadIdx.x; q < (ntiles >> SUP_SHIFT) + 1; q += blockDim.x) { sum += C.view_sup[q]; C.view_sup[q] = 0u; }
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        S->view_sel = S->view_sel == 1u ? 2u : 1u;
        S->view_n = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];   // the survivors counted by k_view_count
        S->view_next_n = 0;
        S->view_dirty = 0;
    }
}

// seam S1c: one chain, slot 0, from a caller-given plane
__global__ void k_r_seam_init(const RKArgs A, float4 hyp, float4 pos, float w_eps, float bitmap_eps) {
    agg_clear(chain_of(cloud_args(A, 0), 0).agg, cloud_args(A, 0).L.nb, threadIdx.x, blockDim.x);
    if (threadIdx.x) return;
    const RCloudArgsK &C = cloud_args(A, 0);
    RState *S = C.st;
    S->n = C.cv.n; S->active = 1; S->done = 0; S->sampling = 0; S->nc = 1; S->npool = 0;
    S->eps3 = w_eps; S->bitmap_eps = bitmap_eps; S->min_support = 0; S->orient = 0; S->err = 0;
    S->aj_n = 1; S->aj_chain[0] = 0; S->aj_slot[0] = 0; S->aj_id[0] = 0; S->aj_out[0] = 0;
    S->n_mark_launches = S->n_mark_chains = 0;
    ChainPtr ch = chain_of(C, 0);
    ch.hdr->cand[0] = hyp; ch.hdr->cand[1] = pos;
    state_from_hyp(&ch.hdr->st[0], hyp, pos);
}

// ------------------------------------------------------------------------------------------------
// average_spacing (code/PLADE/util.cpp:1619-1648) from the Morton order the extraction has built anyway: the k = 6 nearest
// neighbours (FLANN fp32 distances) of <= 10000 strided sample points.  The octree cell of a point at level L is a
// contiguous range of the sorted cloud; a table of the 8^L range starts (one binary search per cell) turns the exact ring
// search of k_knn_grid (k_voxel.hip) into look-ups on data that is already in HBM -- no grid build (cell ids, a 1M-key sort,
// a gather) for 10^4 queries.  The k smallest fp32 distances are the same numbers whatever structure finds them.
struct Cells6Args { const uint32_t *codes[R_G]; uint32_t n[R_G]; uint32_t *table[R_G]; };
__global__ void k_cells6(const Cells6Args A) {   // the sampler's level-6 table of every cloud of the sequence
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x, ncell = 1u << 18;
    if (c > ncell) return;
    const int g = blockIdx.y;
    A.table[g][c] = c == ncell ? A.n[g] : lb_u32(A.codes[g], A.n[g], c << 6);
}
// (both spacing kernels serve the source clouds of all pairs of a group in one launch: blockIdx.y selects the cloud)
struct SpCellsArgs { const uint32_t *codes[PLADE_GROUP_MAX]; uint32_t n[PLADE_GROUP_MAX]; int level[PLADE_GROUP_MAX]; uint32_t *table[PLADE_GROUP_MAX]; };
__global__ void k_sp_cells(const SpCellsArgs A) {
    const int q = blockIdx.y;
    const int level = A.level[q];
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x, ncell = 1u << (3 * level);
    if (c > ncell) return;
    A.table[q][c] = c == ncell ? A.n[q] : lb_u32(A.codes[q], A.n[q], c << (24 - 3 * level));
}

struct SpArgs {
    const float *x, *y, *z;          // Morton-ordered SoA
    const uint32_t *table;           // 8^L + 1 range starts
    const float *aos;                // the cloud as it came (queries are taken in ORIGINAL order, util.cpp:1626-1633)
    uint32_t n, step, nq;
    int level, k;
    float mnx, mny, mnz, inv_cube, cell;   // Morton quantisation (k_morton), cell edge at `level`
    uint32_t dense_limit;
};
constexpr int SPK = 8;
struct SpBatch { SpArgs a[PLADE_GROUP_MAX]; double *avg[PLADE_GROUP_MAX]; uint32_t *nbs[PLADE_GROUP_MAX], *dense[PLADE_GROUP_MAX]; };
__global__ __launch_bounds__(256) void k_sp_knn(const SpBatch B) {
    const SpArgs &A = B.a[blockIdx.y];
    double *__restrict__ avg_out = B.avg[blockIdx.y];
    uint32_t *__restrict__ nbs_out = B.nbs[blockIdx.y], *__restrict__ too_dense = B.dense[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const uint32_t qi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (qi >= A.nq) return;
    const uint32_t pi = min(A.n - 1, qi * A.step);
    const f3 q(A.aos[(size_t)pi * 6], A.aos[(size_t)pi * 6 + 1], A.aos[(size_t)pi * 6 + 2]);
    const int sh = 8 - A.level, dim = 1 << A.level;
    const int cx = (int)(min(255u, (uint32_t)max(0.f, (q.x - A.mnx) * A.inv_cube * 256.f)) >> sh);
    const int cy = (int)(min(255u, (uint32_t)max(0.f, (q.y - A.mny) * A.inv_cube * 256.f)) >> sh);
    const int cz = (int)(min(255u, (uint32_t)max(0.f, (q.z - A.mnz) * A.inv_cube * 256.f)) >> sh);
    float best[SPK];
#pragma unroll
    for (int b = 0; b < SPK; ++b) best[b] = INFINITY;
    float kth[SPK];
    int found = 0;
    bool dense = false;
    // Cells of the outer rings that cannot hold one of the k nearest are not scanned: once k candidates are known, a cell
    // whose box is farther from q than the current k-th distance holds nothing closer (a point AT that distance would
    // not change the sum).  The box is the cell's nominal box widened by 0.1 % of a cell against the rounding of the
    // Morton quantisation.  After the home cell that leaves 1-4 of the 26 neighbours for a point of a surface.
    float prune2 = INFINITY;
    for (int ring = 0; ring <= dim; ++ring) {
        const int w = 2 * ring + 1;
        // the cells of the shell, 64 at a time: every lane looks up one cell's range, then the wavefront walks the non-empty
        // cells one after the other with all lanes striding over the cell's points (coalesced loads from the sorted cloud)
        for (int t0 = 0; t0 < w * w * w; t0 += 64) {
            const int t = t0 + lane;
            uint32_t jb = 0, je = 0;
            if (t < w * w * w) {
                const int ddx = t % w - ring, ddy = (t / w) % w - ring, ddz = t / (w * w) - ring;
                const int x = cx + ddx, y = cy + ddy, z = cz + ddz;
                if (max(abs(ddx), max(abs(ddy), abs(ddz))) == ring && x >= 0 && y >= 0 && z >= 0 && x < dim && y < dim && z < dim) {
                    const float slack = 1e-3f * A.cell;
                    const float lx = A.mnx + (float)x * A.cell, ly = A.mny + (float)y * A.cell, lz = A.mnz + (float)z * A.cell;
                    const float gx = fmaxf(0.f, fmaxf(lx - q.x, q.x - (lx + A.cell)) - slack);
                    const float gy = fmaxf(0.f, fmaxf(ly - q.y, q.y - (ly + A.cell)) - slack);
                    const float gz = fmaxf(0.f, fmaxf(lz - q.z, q.z - (lz + A.cell)) - slack);
                    if (!(gx * gx + gy * gy + gz * gz > prune2)) {
                        const uint32_t c = (spread3((uint32_t)z) << 2) | (spread3((uint32_t)y) << 1) | spread3((uint32_t)x);
                        jb = A.table[c]; je = A.table[c + 1];
                        if (je - jb > A.dense_limit) { dense = true; je = jb; }
                    }
                }
            }
            unsigned long long todo = __ballot(je > jb);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const uint32_t b = __shfl(jb, src, 64), e = __shfl(je, src, 64);
                for (uint32_t j = b + lane; j < e; j += 64) {
                    float d = flann_d2(q, f3(A.x[j], A.y[j], A.z[j]));
                    if (d < best[SPK - 1]) {
#pragma unroll
                        for (int bb = 0; bb < SPK; ++bb)
                            if (d < best[bb]) { const float tt = best[bb]; best[bb] = d; d = tt; }
                    }
                }
            }
        }
        if (__ballot(dense)) break;   // a cell with thousands of points: the caller takes the adaptive grid instead
        // wave-wide K smallest (lists are ascending: every lane offers its current head)
        int head = 0;
        found = 0;
        for (int r = 0; r < A.k; ++r) {
            float v = INFINITY;
#pragma unroll
            for (int b = 0; b < SPK; ++b) if (b == head) v = best[b];
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
        // everything in ring + 1 and beyond is at least ring * cell away from q
        const float bound = (float)ring * A.cell * 0.999f;
        if (found == A.k && kth[A.k - 1] < bound * bound) break;
        if (found == A.k) prune2 = kth[A.k - 1];
    }
    const bool any_dense = __ballot(dense) != 0ull;
    if (lane == 0) {
        if (any_dense) *too_dense = 1u;
        double avg = 0.0;
        for (int r = 1; r < found; ++r) avg += (double)sqrtf(kth[r]);  // util.cpp:1640-1642: starts from 1 to exclude itself
        avg_out[qi] = avg;
        nbs_out[qi] = (uint32_t)found;
    }
}

// ---- seam S1a (plade_score_planes / plade_score_planes_subset): the caller's hypotheses through the loop's OWN K1
// kernels -- counts by k_r_rescore (the pool re-score), ordered lists by k_r_mark + k_r_compact_raster (slot 0 of the
// acceptance chains), subset counts by k_r_score_sub.  These kernels only put the hypotheses where the loop keeps them.
__global__ void k_r_seam_pool(const RKArgs A, const float4 *__restrict__ planes, uint32_t nh, float eps, float cos_t) {
    const RCloudArgsK &C = cloud_args(A, 0);
    RState *S = C.st;
    if (threadIdx.x == 0) {
        S->n = C.cv.n; S->active = 1; S->done = 0; S->sampling = 0; S->fresh = 1; S->npool = nh; S->nc = 0; S->aj_n = 0;
        S->eps = eps; S->cos_t = cos_t; S->n_rescores[0] = S->n_rescores[1] = 0;
    }
    if (threadIdx.x < R_TOP) {
        S->pool_cnt[threadIdx.x] = 0;
        if (threadIdx.x < nh) S->pool_pl[threadIdx.x] = planes[threadIdx.x];
    }
}

__global__ void k_r_seam_chains(const RKArgs A, const float4 *__restrict__ planes, uint32_t nb, float eps, float cos_t) {
    const RCloudArgsK &C = cloud_args(A, 0);
    RState *S = C.st;
    if (threadIdx.x == 0) {
        S->n = C.cv.n; S->active = 1; S->done = 0; S->sampling = 0; S->fresh = 0; S->npool = 0; S->nc = nb; S->aj_n = 0;
        S->eps3 = eps; S->cos_t = cos_t; S->min_support = 0; S->orient = 0; S->err = 0;
        S->bitmap_eps = INFINITY;   // one-pixel bitmaps: only the ordered list of the compaction is asked for
        S->n_mark_launches = S->n_mark_chains = 0;
    }
    for (uint32_t b2 = 0; b2 < nb; ++b2) agg_clear(chain_of(C, b2).agg, C.L.nb, threadIdx.x, blockDim.x);
    if (threadIdx.x < nb) {
        ChainPtr ch = chain_of(C, threadIdx.x);
        const float4 hyp = planes[threadIdx.x], pos = make_float4(0.f, 0.f, 0.f, 0.f);
        ch.hdr->cand[0] = hyp; ch.hdr->cand[1] = pos;
        PlaneState *st = &ch.hdr->st[0];
        state_from_hyp(st, hyp, pos);
        bool fin = true;
        for (int q = 0; q < 3; ++q) fin = fin && fabsf(st->a0[q]) <= 2.f && fabsf(st->a1[q]) <= 2.f;
        if (!fin) {   // a hypothesis without a direction has no in-plane frame: any fixed one serves the list
            st->a0[0] = 1.f; st->a0[1] = 0.f; st->a0[2] = 0.f; st->a1[0] = 0.f; st->a1[1] = 1.f; st->a1[2] = 0.f;
        }
    }
}

__global__ void k_r_seam_hyps(const RKArgs A, const float4 *__restrict__ planes, uint32_t nh, float eps, float cos_t) {
    const RCloudArgsK &C = cloud_args(A, 0);
    RState *S = C.st;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        S->n = C.cv.n; S->active = 1; S->done = 0; S->sampling = 1; S->fresh = 0; S->npool = 0; S->nc = 0; S->aj_n = 0;
        S->eps = eps; S->cos_t = cos_t; S->sub_unassigned = 0;
    }
    if (t >= R_H) return;
    C.hyp[t] = t < nh ? planes[t] : make_float4(0.f, 0.f, 0.f, __int_as_float(0x7fc00000));
    C.hyp_counts[t] = 0;
}

__global__ void k_r_seam_subset(const CloudView cv, const uint32_t *__restrict__ sub_index, uint32_t m, float *__restrict__ sub,
                                uint32_t pitch) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= m) return;
    const uint32_t i = sub_index[s];
    const size_t sp = pitch;
    sub[s] = cv.x[i]; sub[sp + s] = cv.y[i]; sub[2 * sp + s] = cv.z[i];
    sub[3 * sp + s] = cv.nx[i]; sub[4 * sp + s] = cv.ny[i]; sub[5 * sp + s] = cv.nz[i];
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host side
struct RansacSlot {
    uint32_t n = 0;
    const CloudDev *cloud = nullptr;
    CloudDev sorted;
    DBuf<uint32_t> codes, orig, sub_index;
    DBuf<int32_t> out_idx, sub_assigned;
    DBuf<uint8_t> taken;
    DBuf<float> tile_box, view_a, view_b;
    DBuf<uint32_t> view_map_a, view_map_b, view_cnt, view_sup;
    uint32_t sub_stride = 1;
    DBuf<uint32_t> out_pos;
    DBuf<float> sub;
    uint32_t sub_pitch = 0, n_sub = 0;
    DBuf<char> round_block;          // hypotheses / positions / counts of the current round
    DBuf<char> fixed, var;           // the chains' slabs
    DBuf<RState> state;
    RResult *res = nullptr, *res_dev = nullptr;   // host-mapped result block
    ChainLayout L{};
    DBuf<uint32_t> seam_list;
    // average spacing from the Morton order (ransac_spacing_*): cell table, results in host-mapped memory
    DBuf<uint32_t> sp_table, cells6;
    char *sp_host = nullptr, *sp_dev = nullptr;   // nq doubles | nq counts | flag
    uint32_t sp_nq = 0, sp_cap = 0;
    uint64_t sp_epoch = 0;                        // plade_ctx::wait_epoch when the query was queued
    float cube_inv = 0.f, cube = 0.f;             // Morton quantisation of this slot (ransac_prepare)
    ~RansacSlot() { if (res) (void)hipHostFree(res); if (sp_host) (void)hipHostFree(sp_host); }
};

struct RansacWork {
    RansacSlot slot[R_G];
    int ng = 0;
    uint32_t generation = 0;         // detect calls issued on this work area
    DBuf<uint32_t> keys_in, vals_in, keys, perm;
    DBuf<RCloudArgs> args_dev;       // the per-cloud argument table the kernels read (args_commit)
    uint64_t args_hash = 0;          // of the table uploaded last
    std::map<uint64_t, hipGraphExec_t> graphs;   // the iteration sequence, keyed on everything baked into its launches
    std::map<uint64_t, std::vector<std::string>> graph_tags;   // of the graphs captured with clock stamps (plade_ctx::graph_clocks)
    ~RansacWork() { for (auto &kv : graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second); }
};

RansacWork *ransac_work_create() { return new RansacWork; }
void ransac_work_destroy(RansacWork *w) { delete w; }

namespace {

void slot_buffers(plade_ctx *ctx, RansacSlot &s, uint32_t n) {
    s.n = n;
    s.L = make_layout(n);
    if (s.fixed.cap < (size_t)R_B * F_BYTES) {   // fresh bitmaps must be all-zero
        s.fixed.ensure((size_t)R_B * F_BYTES);
        HIP_TRY(hipMemsetAsync(s.fixed.p, 0, s.fixed.cap, ctx->stream));
    }
    s.var.ensure((size_t)R_B * s.L.bytes);
    s.state.ensure(1);
    s.round_block.ensure((size_t)R_H * 36 + 64);
    s.out_idx.ensure((size_t)n + 4);
    s.out_pos.ensure((size_t)n + 4);
    s.taken.ensure((size_t)n + 16);
    {   // the two scan views (compacted copies of the points not yet taken) and the tables of their rebuild
        const size_t pitch = ((size_t)n + 3) & ~(size_t)3;
        s.view_a.ensure(6 * pitch + 4); s.view_b.ensure(6 * pitch + 4);
        s.view_map_a.ensure((size_t)n + 4); s.view_map_b.ensure((size_t)n + 4);
        s.view_cnt.ensure(s.L.nb + 4); s.view_sup.ensure((s.L.nb >> SUP_SHIFT) + 4);
    }
    if (!s.res) {
        HIP_TRY(hipHostMalloc((void **)&s.res, sizeof(RResult), hipHostMallocMapped | hipHostMallocCoherent));
        memset(s.res, 0, sizeof(RResult));
        HIP_TRY(hipHostGetDevicePointer((void **)&s.res_dev, s.res, 0));
    }
}

RArgs make_args(RansacWork &W, int ng, bool topup) {
    RArgs A;
    memset(&A, 0, sizeof(A));
    A.ng = (uint32_t)ng;
    for (int g = 0; g < R_G; ++g) {
        RansacSlot &s = W.slot[g < ng ? g : 0];
        RCloudArgs &C = A.c[g];
        const CloudDev &c = s.sorted;
        // field by field: the bytes of this struct key the captured graphs, padding included (A was zeroed)
        C.cv.x = c.x(); C.cv.y = c.y(); C.cv.z = c.z(); C.cv.nx = c.nx(); C.cv.ny = c.ny(); C.cv.nz = c.nz(); C.cv.n = s.n;
        C.codes = s.codes.p; C.cells6 = s.cells6.p; C.orig = s.orig.p; C.assigned = nullptr; C.taken = s.taken.p; C.tile_box = s.tile_box.p;
        C.view[0] = s.view_a.p; C.view[1] = s.view_b.p; C.view_map[0] = s.view_map_a.p; C.view_map[1] = s.view_map_b.p;
        C.view_cnt = s.view_cnt.p; C.view_sup = s.view_sup.p;
        C.sub = s.sub.p; C.sub_index = s.sub_index.p; C.sub_pitch = s.sub_pitch; C.n_sub = s.n_sub;
        C.sub_assigned = s.sub_assigned.p; C.sub_stride = s.sub_stride;
        C.st = s.state.p; C.res = s.res_dev;
        C.hyp = reinterpret_cast<float4 *>(s.round_block.p);
        C.hyp_pos = C.hyp + R_H;
        C.hyp_counts = reinterpret_cast<uint32_t *>(C.hyp_pos + R_H);
        C.out_idx = s.out_idx.p;
        C.out_pos = s.out_pos.p;
        C.fixed = s.fixed.p; C.var = s.var.p;
        C.list_values = nullptr;
        memcpy(&C.L, &s.L, sizeof(ChainLayout));
    }
    for (int g = 0; g < R_G; ++g) A.tile_start[g + 1] = A.tile_start[g] + (g < ng ? A.c[g].L.nb : 0u);
    A.topup = topup ? 1u : 0u;   // plade_params.ransac_topup
    return A;
}

uint64_t hash_bytes(const void *p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

// The table of per-cloud arguments goes to the device (once per change: every launch that reads it is queued on ctx->stream behind
// the upload, and a call with the same buffers as the last one -- the steady state of a batch -- uploads nothing).
void args_commit(plade_ctx *ctx, RansacWork &W, RArgs &A) {
    W.args_dev.ensure(R_G);
    A.dev = W.args_dev.p;
    const uint64_t h = hash_bytes(A.c, sizeof(A.c)) | 1ull;
    if (h != W.args_hash) {
        const bool staged = ctx->h2d(W.args_dev.p, A.c, sizeof(A.c));
        if (!staged) ctx->sync();
        W.args_hash = h;
    }
}


// one iteration of the detect loop: 30 launches (32 without the top-up rule)
void enqueue_iteration(plade_ctx *ctx, const RArgs &A) {
    hipStream_t st = ctx->stream;
    const uint32_t ng = A.ng;
    uint32_t tiles = 0, nb_max = 0, sub_tiles = 0;
    for (uint32_t g = 0; g < ng; ++g) {
        tiles += A.c[g].L.nb;
        nb_max = std::max(nb_max, A.c[g].L.nb);
        sub_tiles = std::max(sub_tiles, cdiv(A.c[g].n_sub, TILE));
    }
    // what the previous iteration left in the pool: re-score, prune, pick a batch ...  (with the top-up rule the old pool
    // goes into the new round instead and these two launches are not part of the sequence)
    if (!A.topup) {
        ctx->ev_begin("score_multi", 0.0);
        hipLaunchKernelGGL(k_r_rescore, dim3(std::min(VGRID_WGS, tiles)), dim3(TPB), 0, st, A, 0, ctx->ev_clock());
        ctx->ev_end();
        hipLaunchKernelGGL(k_r_select, dim3(ng), dim3(64), 0, st, A, 0);
    }
    // ... and if nothing is left (or at the start), a new round: sample, score on the subset, leaders, re-score, batch
    hipLaunchKernelGGL(k_r_sample, dim3(cdiv(R_H, 256), ng), dim3(256), 0, st, A);
    hipLaunchKernelGGL(k_r_score_sub, dim3(std::min(VGRID_WGS, sub_tiles * (R_H / HCHUNK) * ng)), dim3(TPB), 0, st, A);
    hipLaunchKernelGGL(k_r_leaders, dim3(ng), dim3(1024), 0, st, A);
    ctx->ev_begin("score_multi", 0.0);
    hipLaunchKernelGGL(k_r_rescore, dim3(std::min(VGRID_WGS, tiles)), dim3(TPB), 0, st, A, 1, ctx->ev_clock());
    ctx->ev_end();
    hipLaunchKernelGGL(k_r_select, dim3(ng), dim3(64), 0, st, A, 1);
    for (int k = 0; k < 4; ++k) {
        ctx->ev_begin("score_mark", 0.0);
        hipLaunchKernelGGL(k_r_mark, dim3(std::min(VGRID_WGS, tiles)), dim3(TPB), 0, st, A, k, ctx->ev_clock());
        ctx->ev_end();
        hipLaunchKernelGGL(k_r_compact_raster, dim3(std::min(VGRID_WGS, loop_grid(nb_max) * R_B * ng)), dim3(TPB), 0, st, A, k);
        hipLaunchKernelGGL(k_r_label, dim3(R_B * ng), dim3(1024), 0, st, A, k, 1);
        hipLaunchKernelGGL(k_r_select_cc, dim3(std::min(VGRID_WGS, loop_grid(nb_max) * R_B * ng)), dim3(TPB), 0, st, A, k);
        hipLaunchKernelGGL(k_r_fit, dim3(R_B * ng), dim3(256), 0, st, A, k);
    }
    hipLaunchKernelGGL(k_r_decide, dim3(ng), dim3(DEC_T), 0, st, A);
    hipLaunchKernelGGL(k_r_assign, dim3(std::min(VGRID_WGS, loop_grid(nb_max) * R_B * ng)), dim3(TPB), 0, st, A);
    // the scan view of the next iteration: the points this one left
    hipLaunchKernelGGL(k_view_count, dim3(std::min(VGRID_WGS, tiles)), dim3(TPB), 0, st, A);
    hipLaunchKernelGGL(k_view_compact, dim3(std::min(VGRID_WGS, tiles)), dim3(TPB), 0, st, A);
    hipLaunchKernelGGL(k_view_commit, dim3(ng), dim3(256), 0, st, A);
}

void launch_iteration(plade_ctx *ctx, RansacWork &W, const RArgs &A) {
    const bool stamped = ctx->graph_clocks();   // profiled on the graph path: its own graph, with clock pointers in the scan launches
    static const bool no_graph = getenv("PLADE_NO_GRAPH") != nullptr;   // A/B timing hook, looked up once
    if ((ctx->profiling() && !stamped) || no_graph) { enqueue_iteration(ctx, A); HIP_TRY(hipGetLastError()); return; }
    const uint64_t key = hash_bytes(&A, sizeof(A)) ^ (stamped ? 0x9e3779b97f4a7c15ull : 0ull);
    auto it = W.graphs.find(key);
    if (it == W.graphs.end()) {
        if (W.graphs.size() > 64) {   // bounded cache (a batch of differently sized clouds)
            for (auto &kv : W.graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
            W.graphs.clear();
            W.graph_tags.clear();
        }
        hipGraph_t graph = nullptr;
        if (stamped) { ctx->ensure_clk(); ctx->capturing = true; ctx->cap_slot = 0; ctx->cap_tags.clear(); }
        HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        try { enqueue_iteration(ctx, A); }
        catch (...) { ctx->capturing = false; (void)hipStreamEndCapture(ctx->stream, &graph); if (graph) (void)hipGraphDestroy(graph); throw; }
        ctx->capturing = false;
        HIP_TRY(hipStreamEndCapture(ctx->stream, &graph));
        if (stamped) W.graph_tags[key] = ctx->cap_tags;
        hipGraphExec_t e = nullptr;
        HIP_TRY(hipGraphInstantiate(&e, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        it = W.graphs.emplace(key, e).first;
    }
    HIP_TRY(hipGraphLaunch(it->second, ctx->stream));
    if (stamped) ctx->ev_graph_launched(W.graph_tags[key]);
}

// Waits until every listed result block reports at least `want` completed iterations (or the end of its detect call).
// The blocks are host-mapped and written by the device while the stream keeps running; should a flag not become visible
// (it always has), the stream running dry ends the wait: everything is visible then.
inline bool flag_done(const RResult *r, uint32_t gen) {
    const uint32_t f = *reinterpret_cast<const volatile uint32_t *>(&r->flag);
    return ((f >> 24) & 0x7fu) == (gen & 0x7fu) && (f & 0x80000000u);
}
void wait_iterations(plade_ctx *ctx, RResult *const *res, int nres, uint32_t want, uint32_t gen) {
    auto reached = [&]() {
        for (int i = 0; i < nres; ++i) {
            const uint32_t f = *reinterpret_cast<volatile uint32_t *>(&res[i]->flag);
            if (((f >> 24) & 0x7fu) != (gen & 0x7fu)) return false;          // still the previous call's reports
            if (!(f & 0x80000000u) && (f & 0xffffffu) < want) return false;
        }
        return true;
    };
    if (ctx->params.host_wait != 0) relax_timer_slack();
    const double cpu0 = thread_cpu_seconds();
    struct Acc { Stats &st; double c0; ~Acc() { st.add("cpu_ransac_wait_polls", thread_cpu_seconds() - c0); } } acc{ctx->stats, cpu0};
    // the flag is what is waited for; the stream query only catches a device loop that died without reporting (it takes the
    // runtime's locks, so the sleeping modes ask every 16th poll, the spinning mode every 64th)
    const uint32_t query_mask = 15u;
    for (uint32_t polls = 0;; ++polls) {
        if (reached()) break;
        if ((polls & 63u) == 63u || (ctx->params.host_wait != 0 && (polls & query_mask) == query_mask)) {
            const hipError_t e = hipStreamQuery(ctx->stream);
            if (e == hipSuccess) { if (reached()) break; throw Err{PLADE_EDEVICE, "plane extraction: the device loop did not report"}; }
            if (e != hipErrorNotReady) throw Err{PLADE_EDEVICE, std::string("hipStreamQuery: ") + hipGetErrorString(e)};
        }
        if (ctx->params.host_wait != 0) poll_sleep((int)polls, ctx->in_group);
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    ++ctx->wait_epoch;   // everything queued before the reported iteration has finished
}

}  // namespace

void ransac_prepare(plade_ctx *ctx, RansacWork &W, const CloudDev *const clouds[RANSAC_SLOTS], int n_clouds) {
    PLADE_REQUIRE(n_clouds >= 1 && n_clouds <= R_G, PLADE_EINVAL, "ransac: too many clouds");
    W.ng = n_clouds;
    MortonArgs M;
    GatherArgs G;
    memset(&M, 0, sizeof(M));
    memset(&G, 0, sizeof(G));
    M.ng = G.ng = (uint32_t)n_clouds;
    uint64_t total = 0;
    for (int g = 0; g < n_clouds; ++g) {
        const CloudDev &c = *clouds[g];
        RansacSlot &s = W.slot[g];
        s.cloud = clouds[g];
        slot_buffers(ctx, s, c.n);
        M.start[g] = G.start[g] = (uint32_t)total;
        total += c.n;
        const float cube = std::max({c.bbmax[0] - c.bbmin[0], c.bbmax[1] - c.bbmin[1], c.bbmax[2] - c.bbmin[2], 1e-30f});
        M.c[g] = MortonIn{c.x(), c.y(), c.z(), c.n, c.bbmin[0], c.bbmin[1], c.bbmin[2], 1.f / cube};
        s.cube = cube; s.cube_inv = 1.f / cube;
        s.sorted.n = c.n;
        s.sorted.pitch = c.pitch;
        s.sorted.soa.ensure(6 * c.pitch + 4);
        for (int k = 0; k < 3; ++k) { s.sorted.bbmin[k] = c.bbmin[k]; s.sorted.bbmax[k] = c.bbmax[k]; }
        s.codes.ensure((size_t)c.n + 4); s.orig.ensure((size_t)c.n + 4);
        // stratified subset: every stride-th point of the Morton order
        const uint32_t stride = std::max(1u, c.n / 16384u);
        s.n_sub = c.n ? (c.n + stride - 1) / stride : 0;
        s.sub_pitch = (s.n_sub + 3) & ~3u;
        s.sub.ensure(6 * (size_t)s.sub_pitch + 4);
        s.sub_index.ensure((size_t)s.sub_pitch + 4);
        s.sub_assigned.ensure((size_t)s.sub_pitch + 8);
        s.sub_stride = stride;
        G.c[g] = GatherOut{c.aos.p, s.sorted.soa.p, (uint32_t)c.pitch, s.codes.p, s.orig.p, nullptr, s.sub.p, s.sub_pitch, s.n_sub,
                           stride, s.sub_index.p, c.n};
    }
    PLADE_REQUIRE(total < (1ull << 31), PLADE_ELIMIT, "ransac: too many points");
    if (total == 0) return;
    for (int g = n_clouds; g < R_G; ++g) { M.c[g] = M.c[0]; G.c[g] = G.c[0]; }   // never selected (ng), kept valid
    for (int g = n_clouds; g <= R_G; ++g) M.start[g] = G.start[g] = (uint32_t)total;
    W.keys_in.ensure(total); W.vals_in.ensure(total); W.keys.ensure(total); W.perm.ensure(total);
    hipLaunchKernelGGL(k_morton, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, M, W.keys_in.p, W.vals_in.p);
    // One sort per CLOUD (the slot bits above bit 23 are equal inside a cloud's range) -- as SEGMENTS of one launch sequence
    // (radix_sort.hip: every cloud keeps its own histogram and prefix chain; one chain over the keys of all clouds of a group
    // was measured at 426 us per pass under load against ~25 per 1M-key cloud, sixteen sorts one after the other are 64 launches)
    {
        uint32_t seg_off[R_G + 1];
        int nseg = 0;
        for (int g = 0; g < n_clouds; ++g) {
            if (M.start[g + 1] == M.start[g]) continue;   // an empty cloud has nothing to sort
            seg_off[nseg] = M.start[g];
            seg_off[++nseg] = M.start[g + 1];
        }
        if (nseg) radix_sort_segments_u32(ctx, W.keys_in.p, W.keys.p, W.vals_in.p, W.perm.p, seg_off, nseg, 24);
    }
    hipLaunchKernelGGL(k_gather_cloud, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, G, W.keys.p, W.perm.p);
    Cells6Args C6;
    memset(&C6, 0, sizeof(C6));
    for (int g = 0; g < n_clouds; ++g) {
        RansacSlot &s = W.slot[g];
        s.cells6.ensure((1u << 18) + 2);
        C6.codes[g] = s.codes.p; C6.n[g] = s.n; C6.table[g] = s.cells6.p;
    }
    hipLaunchKernelGGL(k_cells6, dim3(cdiv((1u << 18) + 1, 256), n_clouds), dim3(256), 0, ctx->stream, C6);
    TileBoxArgs TB;
    memset(&TB, 0, sizeof(TB));
    TB.ng = (uint32_t)n_clouds;
    for (int g = 0; g < R_G; ++g) {
        RansacSlot &s = W.slot[g < n_clouds ? g : 0];
        if (g < n_clouds) s.tile_box.ensure(8 * (size_t)s.L.nb + 8);
        TB.x[g] = s.sorted.x(); TB.y[g] = s.sorted.y(); TB.z[g] = s.sorted.z(); TB.box[g] = s.tile_box.p; TB.n[g] = s.n;
        TB.tile_start[g + 1] = TB.tile_start[g] + (g < n_clouds ? s.L.nb : 0u);
    }
    if (TB.tile_start[R_G]) hipLaunchKernelGGL(k_tile_boxes, dim3(TB.tile_start[R_G]), dim3(K1_TPB), 0, ctx->stream, TB);
    HIP_TRY(hipGetLastError());
}

void ransac_detect_prepared(plade_ctx *ctx, RansacWork &W, RansacJob jobs[RANSAC_SLOTS]) {
    const int ng = W.ng;
    PLADE_REQUIRE(ng >= 1, PLADE_EINVAL, "ransac: not prepared");
    Clock::time_point t0 = Clock::now();
    RArgs A = make_args(W, ng, ctx->params.ransac_topup != 0);
    args_commit(ctx, W, A);
    RInit I;
    memset(&I, 0, sizeof(I));
    RResult *res[R_G];
    int nres = 0;
    const uint32_t gen = (++W.generation) & 0x7fu;
    for (int g = 0; g < ng; ++g) {
        RansacJob &J = jobs[g];
        RansacSlot &s = W.slot[g];
        if (!J.active || s.n < 3) {
            if (J.active && J.out) {   // plane_extraction.cpp:181-184: fewer than three points, no planes
                J.out->coef.clear(); J.out->offsets.assign(1, 0); J.out->idx.clear(); J.out->d_idx = nullptr; J.out->remaining = s.n;
                J.out->n_score_passes = 0; J.out->score_bytes = 0;
            }
            J.active = false;
            continue;
        }
        const CloudDev &c = *s.cloud;
        // scale exactly as plane_extraction.cpp:71-80 + PointCloud.h:94-98 (Z bug: maxZ stays -FLT_MAX, so the Z extent
        // never wins the max)
        const float scale = std::max(c.bbmax[0] - c.bbmin[0], c.bbmax[1] - c.bbmin[1]);
        const float eps = J.rp.dist_rel * scale, bitmap_eps = J.rp.bitmap_rel * scale;
        PLADE_REQUIRE(eps > 0.f && bitmap_eps > 0.f, PLADE_EINVAL, "plane extraction: degenerate bounding box");
        RInitCloud &P = I.c[g];
        P.active = 1; P.min_support = J.rp.min_support; P.orient = J.rp.orient_normals ? 1u : 0u;
        P.eps = eps; P.eps3 = 3 * eps;   // RansacShapeDetector.cpp:471-473
        P.bitmap_eps = bitmap_eps; P.cos_t = J.rp.cos_thresh; P.overlook_p = J.rp.overlook_p;
        for (int k = 0; k < 3; ++k) { P.bbmin[k] = c.bbmin[k]; P.bbmax[k] = c.bbmax[k]; }
        P.seed = J.rp.seed;
        P.gen = gen;
        P.topup = A.topup;
        // a slot that sat out earlier calls still shows the flag of its last one; with a 7-bit tag that would look like
        // THIS call's "done" after 128 calls.  (A speculative iteration of the previous call may still write its own,
        // older tag over the zero: harmless.)
        *reinterpret_cast<volatile uint32_t *>(&s.res->flag) = 0u;
        res[nres++] = s.res;
    }
    if (nres == 0) return;
    uint32_t tiles = 0;
    for (int g = 0; g < ng; ++g) tiles += A.c[g].L.nb;
    hipLaunchKernelGGL(k_r_init, dim3(tiles), dim3(TPB), 0, ctx->stream, A, I);
    HIP_TRY(hipGetLastError());
    uint32_t iterations = 0;
    if (ctx->profiling()) {
        // profiled run (HIP events around the scan kernels): one iteration at a time; the device-side counters say
        // which clouds each scan launch really served, i.e. its algorithmic bytes (28 B per point and
        // launch + one mask byte per 4 points and chain)
        uint32_t seen_rescore[R_G][2] = {}, seen_mark[R_G] = {}, seen_chains[R_G] = {};
        const size_t hdr_bytes = offsetof(RState, acc_coef);
        std::vector<char> hdr(R_G * hdr_bytes);
        for (;; ++iterations) {
            const size_t ev0 = ctx->evs.size();
            launch_iteration(ctx, W, A);
            for (int g = 0; g < ng; ++g)
                if (jobs[g].active) ctx->d2h(hdr.data() + g * hdr_bytes, W.slot[g].state.p, hdr_bytes);
            ctx->sync();
            if (ctx->graph_clocks()) ctx->ev_graph_clocks(ev0);
            double rescore_bytes[2] = {0, 0}, mark_bytes = 0;
            uint32_t mark_launches = 0;
            for (int g = 0; g < ng; ++g) {
                if (!jobs[g].active) continue;
                const RState &S = *reinterpret_cast<const RState *>(hdr.data() + g * hdr_bytes);
                const double n = W.slot[g].n;
                for (int ph = 0; ph < 2; ++ph) {
                    rescore_bytes[ph] += 28.0 * n * (S.n_rescores[ph] - seen_rescore[g][ph]);
                    seen_rescore[g][ph] = S.n_rescores[ph];
                }
                mark_bytes += 28.0 * n * (S.n_mark_launches - seen_mark[g]) + 0.25 * n * (S.n_mark_chains - seen_chains[g]);
                mark_launches = std::max(mark_launches, S.n_mark_launches - seen_mark[g]);
                if (getenv("PLADE_TRACE_RANSAC"))
                    fprintf(stderr, "[ransac] it %u cloud %d: round %u sampling %u pool %u accepted %u remaining %u batches %u accepts %u (chains this it: %u) done %u\n",
                            iterations, g, S.round, S.sampling, S.npool, S.n_acc, S.n_remaining, S.n_batches, S.n_accepts,
                            S.n_mark_chains - seen_chains[g] ? (S.n_mark_chains - seen_chains[g]) : 0u, S.done);
                seen_mark[g] = S.n_mark_launches; seen_chains[g] = S.n_mark_chains;
            }
            uint32_t mk = 0, rs = 0;
            for (size_t e = ev0; e < ctx->evs.size(); ++e) {
                plade_ctx::EvRec &r = ctx->evs[e];
                if (r.tag == "score_multi") { r.bytes = rs < 2 && rescore_bytes[rs] > 0 ? rescore_bytes[rs] : -1.0; ++rs; }
                else if (r.tag == "score_mark") { r.bytes = mk < mark_launches ? mark_bytes / mark_launches : -1.0; ++mk; }
            }
            bool all = true;
            for (int i = 0; i < nres; ++i) all = all && flag_done(res[i], gen);
            if (all) break;
        }
    } else {
        // Sleeping host waits (several registrations in flight): the next iteration is queued before the current one has
        // reported, so the GPU never waits for the host; should the loop have ended, that iteration's kernels return at
        // once (27 empty launches).  A spinning host reacts within microseconds and queues an iteration only when needed.
        // host_wait = 2 only: with many registrations in flight the hardware queues are never idle, and the 27 empty launches
        // of the one speculative iteration that finds the loop finished cost more (3 % of the throughput at 8 in flight)
        // than the stream's idle time while the host reacts
        const bool speculate = ctx->params.host_wait == 2;
        if (speculate) launch_iteration(ctx, W, A);
        for (;; ++iterations) {
            launch_iteration(ctx, W, A);
            wait_iterations(ctx, res, nres, iterations + 1, gen);
            bool all = true;
            for (int i = 0; i < nres; ++i) all = all && flag_done(res[i], gen);
            if (all) break;
            PLADE_REQUIRE(iterations < 100000, PLADE_EDEVICE, "plane extraction: the device loop does not end");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    ctx->stats.add("ransac_iterations", iterations + 1);
    ctx->stats.add("ransac_t_detect", secs_since(t0));
    for (int g = 0; g < ng; ++g) {
        RansacJob &J = jobs[g];
        if (!J.active) continue;
        RansacSlot &s = W.slot[g];
        const RResult &R = *s.res;
        PLADE_REQUIRE(R.err != 1, PLADE_ELIMIT, "plane extraction: connected-component bitmap too large");
        PLADE_REQUIRE(R.err != 3, PLADE_ELIMIT, "plane extraction: too many shapes");
        PlaneSetOut &out = *J.out;
        if (getenv("PLADE_TRACE_RANSAC"))
            for (uint32_t i = 0; i < R.n_acc && i < 24; ++i)
                fprintf(stderr, "[ransac] cloud %d shape %u: iteration %u chain %u slot %u support %u offset %u coef %.4f %.4f %.4f %.4f\n", g, i,
                        R.dbg[i] >> 16, (R.dbg[i] >> 8) & 255u, R.dbg[i] & 255u, R.support[i], R.offset[i], R.coef[i][0], R.coef[i][1], R.coef[i][2], R.coef[i][3]);
        out.coef.clear(); out.offsets.assign(1, 0); out.idx.clear();
        for (uint32_t i = 0; i < R.n_acc; ++i) {
            if (!R.support[i]) continue;
            out.coef.insert(out.coef.end(), R.coef[i], R.coef[i] + 4);
            out.offsets.push_back((int32_t)(R.offset[i] + R.support[i]));
        }
        out.d_idx = reinterpret_cast<const uint32_t *>(s.out_idx.p);
        out.d_pos = s.out_pos.p;
        out.m_x = s.sorted.x(); out.m_y = s.sorted.y(); out.m_z = s.sorted.z();
        out.remaining = R.remaining;
        out.n_score_passes = R.n_rescores + R.n_mark_launches;
        out.score_bytes = 28.0 * s.n * (R.n_rescores + R.n_mark_launches) + 0.25 * s.n * R.n_mark_chains;
        Stats &jst = J.stats ? *J.stats : ctx->stats;
        jst.add("ransac_rounds", R.n_rounds);
        jst.add("ransac_accepts", R.n_accepts);
        jst.add("ransac_batches", R.n_batches);
        jst.add("ransac_rescore_launches", R.n_rescores);
        jst.add("ransac_mark_launches", R.n_mark_launches);
        jst.add("ransac_deferred_chains", R.n_deferred);
        for (int q = 0; q < 4; ++q) jst.add("ransac_final_slot" + std::to_string(q), R.n_final[q]);
        for (int q = 1; q < 5; ++q) jst.add("ransac_loop_stops_at" + std::to_string(q), R.n_stop[q]);
        if (J.rp.host_indices) {
            out.idx.resize(R.out_off);
            if (R.out_off) ctx->d2h(out.idx.data(), s.out_idx.p, 4 * (size_t)R.out_off);
        }
    }
    bool any_host = false;
    for (int g = 0; g < ng; ++g) any_host = any_host || (jobs[g].active && jobs[g].rp.host_indices);
    if (any_host) ctx->sync();
}

void ransac_detect(plade_ctx *ctx, RansacWork &W, const CloudDev &cloud, const RansacParams &rp, PlaneSetOut &out) {
    const CloudDev *cl[R_G] = {&cloud};
    ransac_prepare(ctx, W, cl, 1);
    RansacJob jobs[R_G];
    jobs[0].active = true; jobs[0].rp = rp; jobs[0].out = &out;
    ransac_detect_prepared(ctx, W, jobs);
}

void plane_component(plade_ctx *ctx, RansacWork &W, const CloudDev &cloud, const float normal[3], const float point[3],
                     const int32_t *idx, uint32_t m, float bitmap_eps, bool closing_filter, float w_eps, ComponentOut &out) {
    out.kept.clear();
    out.wscore = 0;
    out.err = 0;
    for (float &f : out.fit) f = 0.f;
    const uint32_t n = cloud.n;
    PLADE_REQUIRE(m <= n && bitmap_eps > 0.f, PLADE_EINVAL, "plane_component: bad argument");
    if (m == 0) return;
    RansacSlot &s = W.slot[0];
    W.ng = 0;   // the slot no longer holds a prepared cloud
    slot_buffers(ctx, s, n);
    s.seam_list.ensure((size_t)n + 4);
    hipStream_t st = ctx->stream;
    ctx->h2d(s.seam_list.p, idx, 4 * (size_t)m);
    RArgs A;
    memset(&A, 0, sizeof(A));
    A.ng = 1;
    for (int g = 0; g < R_G; ++g) {
        RCloudArgs &C = A.c[g];
        C.cv = CloudView{cloud.x(), cloud.y(), cloud.z(), cloud.nx(), cloud.ny(), cloud.nz(), n};
        C.st = s.state.p; C.res = s.res_dev; C.out_idx = s.out_idx.p; C.out_pos = nullptr; C.fixed = s.fixed.p; C.var = s.var.p; C.list_values = s.seam_list.p; C.L = s.L;
    }
    for (int g = 0; g < R_G; ++g) A.tile_start[g + 1] = A.tile_start[g] + (g == 0 ? s.L.nb : 0u);
    args_commit(ctx, W, A);
    // Plane(point, normal): dist = point . normal with Vec3f::dot's left-to-right sum (Plane.cpp:21-26)
    float dist = point[0] * normal[0];
    dist += point[1] * normal[1];
    dist += point[2] * normal[2];
    hipLaunchKernelGGL(k_r_seam_init, dim3(1), dim3(64), 0, st, A, make_float4(normal[0], normal[1], normal[2], dist),
                       make_float4(point[0], point[1], point[2], 0.f), w_eps, bitmap_eps);
    const uint32_t nb = s.L.nb;
    hipLaunchKernelGGL(k_r_list_mark, dim3(nb), dim3(TPB), 0, st, A, m);
    hipLaunchKernelGGL(k_r_compact_raster, dim3(std::min(VGRID_WGS, loop_grid(nb))), dim3(TPB), 0, st, A, 0);
    hipLaunchKernelGGL(k_r_label, dim3(R_B), dim3(1024), 0, st, A, 0, closing_filter ? 1 : 0);
    hipLaunchKernelGGL(k_r_select_cc, dim3(std::min(VGRID_WGS, loop_grid(nb))), dim3(TPB), 0, st, A, 0);
    hipLaunchKernelGGL(k_r_fit, dim3(R_B), dim3(256), 0, st, A, 0);
    hipLaunchKernelGGL(k_r_assign, dim3(std::min(VGRID_WGS, loop_grid(nb))), dim3(TPB), 0, st, A);
    PlaneState hst[2];
    ctx->d2h(hst, s.fixed.p, 2 * sizeof(PlaneState));   // chain 0's header
    ctx->sync(st);
    HIP_TRY(hipGetLastError());
    out.err = hst[0].err;
    if (out.err) return;
    const uint32_t nk = hst[0].n_kept;
    out.kept.resize(nk);
    if (nk) HIP_TRY(hipMemcpy(out.kept.data(), s.out_idx.p, 4 * (size_t)nk, hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) { out.fit[k] = hst[1].n[k]; out.fit[3 + k] = hst[1].pos[k]; }
    out.fit[6] = hst[1].dist;
    out.wscore = hst[0].wscore;
}

// ---- average spacing from the Morton order -------------------------------------------------------------------------
void ransac_spacing_enqueue(plade_ctx *ctx, RansacWork &W, const int *slots, int count, int k, uint32_t samples) {
    PLADE_REQUIRE(count >= 1 && count <= PLADE_GROUP_MAX, PLADE_EINVAL, "spacing: bad argument");
    SpCellsArgs CA;
    SpBatch B;
    memset(&CA, 0, sizeof(CA));
    memset(&B, 0, sizeof(B));
    uint32_t max_cells = 0, max_nq = 0;
    int used = 0;
    for (int qi = 0; qi < count; ++qi) {
        const int slot = slots[qi];
        PLADE_REQUIRE(k >= 1 && k <= SPK && slot < W.ng, PLADE_EINVAL, "spacing: bad argument");
        RansacSlot &s = W.slot[slot];
        s.sp_nq = 0;
        const uint32_t n = s.n;
        if (n == 0) continue;
        const CloudDev &c = *s.cloud;
        size_t step = 1;
        if (n > samples) step = n / samples;                    // util.cpp:1626-1629
        const uint32_t nq = (uint32_t)((n + step - 1) / step);
        // level: ~25 points per occupied cell of a surface-like cloud.  The wavefront scans a cell with all lanes and skips the
        // neighbour cells that lie beyond the current k-th distance, so smaller cells mean fewer points looked at (measured at
        // 1M points under load, k_sp_cells + k_sp_knn: ~100 points per cell (level 6) 11 + 124 us, ~25 (level 7) 13 + 66 us;
        // level 8: 43 + 72 us, the table build takes over)
        const double ex = std::max(1e-9, (double)c.bbmax[0] - c.bbmin[0]), ey = std::max(1e-9, (double)c.bbmax[1] - c.bbmin[1]),
                     ez = std::max(1e-9, (double)c.bbmax[2] - c.bbmin[2]);
        const double area = 2 * (ex * ey + ey * ez + ex * ez);
        const double want = std::sqrt(32.0 * area / (double)n);
        int level = 2;
        static const int max_level = [] { const char *e = getenv("PLADE_SPACING_LEVEL"); return e ? atoi(e) : 7; }();
        while (level < max_level && (double)s.cube / (double)(1 << (level + 1)) >= 0.7 * want) ++level;
        const uint32_t ncell = 1u << (3 * level);
        s.sp_table.ensure((size_t)ncell + 2);
        const uint32_t need = nq * 12 + 64;
        if (s.sp_cap < need) {
            if (s.sp_host) HIP_TRY(hipHostFree(s.sp_host));
            s.sp_host = nullptr;
            HIP_TRY(hipHostMalloc((void **)&s.sp_host, need + need / 4, hipHostMallocMapped | hipHostMallocCoherent));
            HIP_TRY(hipHostGetDevicePointer((void **)&s.sp_dev, s.sp_host, 0));
            s.sp_cap = need + need / 4;
        }
        uint32_t *flag_host = reinterpret_cast<uint32_t *>(s.sp_host + (size_t)nq * 12);
        *flag_host = 0u;
        CA.codes[used] = s.codes.p; CA.n[used] = n; CA.level[used] = level; CA.table[used] = s.sp_table.p;
        B.a[used] = SpArgs{s.sorted.x(), s.sorted.y(), s.sorted.z(), s.sp_table.p, c.aos.p, n, (uint32_t)step, nq, level, k,
                           c.bbmin[0], c.bbmin[1], c.bbmin[2], s.cube_inv, s.cube / (float)(1 << level), 4096u};
        B.avg[used] = reinterpret_cast<double *>(s.sp_dev);
        B.nbs[used] = reinterpret_cast<uint32_t *>(s.sp_dev + (size_t)nq * 8);
        B.dense[used] = reinterpret_cast<uint32_t *>(s.sp_dev + (size_t)nq * 12);
        max_cells = std::max(max_cells, ncell + 1);
        max_nq = std::max(max_nq, nq);
        s.sp_nq = nq;
        s.sp_epoch = ctx->wait_epoch;
        ++used;
    }
    if (!used) return;
    hipLaunchKernelGGL(k_sp_cells, dim3(cdiv(max_cells, 256), used), dim3(256), 0, ctx->stream, CA);
    hipLaunchKernelGGL(k_sp_knn, dim3(cdiv(max_nq, 4), used), dim3(256), 0, ctx->stream, B);
    HIP_TRY(hipGetLastError());
}

// after the stream has passed the kernels above (any later sync of it); false: nothing was queued or the cloud is too
// clumped for the octree cells (the caller falls back to average_spacing_dev's adaptive grid)
bool ransac_spacing_finish(plade_ctx *ctx, RansacWork &W, int slot, float *spacing_out) {
    RansacSlot &s = W.slot[slot];
    const uint32_t nq = s.sp_nq;
    s.sp_nq = 0;
    if (!nq) return false;
    // the results sit in host-mapped memory behind two kernels of the context's stream: normally an iteration of the
    // extraction, queued after them, has reported meanwhile; if no wait has completed since they were queued (a detect
    // call that had nothing to do), wait here
    if (s.sp_epoch == ctx->wait_epoch) ctx->sync();
    std::atomic_thread_fence(std::memory_order_acquire);
    const double *avg = reinterpret_cast<const double *>(s.sp_host);
    const uint32_t *nbs = reinterpret_cast<const uint32_t *>(s.sp_host + (size_t)nq * 8);
    if (*reinterpret_cast<const volatile uint32_t *>(s.sp_host + (size_t)nq * 12)) return false;
    // util.cpp:1630-1647: sequential double accumulation in sample order
    double total = 0.0;
    size_t total_count = 0;
    for (uint32_t i = 0; i < nq; ++i) {
        const int nb = (int)nbs[i];
        if (nb <= 1) continue;
        total += (avg[i] / nb);
        ++total_count;
    }
    *spacing_out = static_cast<float>(total / total_count);
    return true;
}

// ---- seam S1a on the loop's kernels ------------------------------------------------------------------------------
namespace {
RArgs seam_args(plade_ctx *ctx, RansacWork &W, const CloudDev &cloud, const int32_t *d_assigned) {
    RansacSlot &s = W.slot[0];
    W.ng = 0;   // the slot no longer holds a prepared cloud
    slot_buffers(ctx, s, cloud.n);
    RArgs A;
    memset(&A, 0, sizeof(A));
    A.ng = 1;
    for (int g = 0; g < R_G; ++g) {
        RCloudArgs &C = A.c[g];
        C.cv = CloudView{cloud.x(), cloud.y(), cloud.z(), cloud.nx(), cloud.ny(), cloud.nz(), cloud.n};
        C.assigned = const_cast<int32_t *>(d_assigned);
        C.st = s.state.p; C.res = s.res_dev; C.out_idx = s.out_idx.p; C.out_pos = nullptr; C.fixed = s.fixed.p; C.var = s.var.p; C.L = s.L;
        C.hyp = reinterpret_cast<float4 *>(s.round_block.p);
        C.hyp_pos = C.hyp + R_H;
        C.hyp_counts = reinterpret_cast<uint32_t *>(C.hyp_pos + R_H);
    }
    for (int g = 0; g < R_G; ++g) A.tile_start[g + 1] = A.tile_start[g] + (g == 0 ? s.L.nb : 0u);
    args_commit(ctx, W, A);
    return A;
}
}  // namespace

void score_planes_seam(plade_ctx *ctx, RansacWork &W, const CloudDev &cloud, const int32_t *d_assigned, const float *planes,
                       uint32_t h, float eps, float cos_t, uint32_t *counts, uint32_t *idx_out, uint32_t cap) {
    for (uint32_t j = 0; j < h; ++j) counts[j] = 0;
    if (cloud.n == 0 || h == 0) return;
    RArgs A = seam_args(ctx, W, cloud, d_assigned);
    RansacSlot &s = W.slot[0];
    hipStream_t st = ctx->stream;
    DBuf<float4> d_planes;
    d_planes.ensure(h);
    HIP_TRY(hipMemcpyAsync(d_planes.p, planes, 16 * (size_t)h, hipMemcpyHostToDevice, st));
    const uint32_t tiles = s.L.nb;
    // counts: the hypotheses take the place of the candidate pool, R_TOP at a time (k_r_rescore, phase 1 = a fresh pool)
    for (uint32_t h0 = 0; h0 < h; h0 += R_TOP) {
        const uint32_t nh = std::min(R_TOP, h - h0);
        hipLaunchKernelGGL(k_r_seam_pool, dim3(1), dim3(64), 0, st, A, d_planes.p + h0, nh, eps, cos_t);
        hipLaunchKernelGGL(k_r_rescore, dim3(std::min(VGRID_WGS, tiles)), dim3(TPB), 0, st, A, 1, (unsigned long long *)nullptr);
        HIP_TRY(hipMemcpyAsync(counts + h0, reinterpret_cast<const char *>(s.state.p) + offsetof(RState, pool_cnt), 4 * (size_t)nh,
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipGetLastError());
    if (!idx_out || !cap) return;
    // ordered lists: the hypotheses take the place of the acceptance chains' candidates, R_B at a time: mark (4-bit masks +
    // per-tile counts) -> ordered compaction (+ a one-pixel bitmap, which the labelling kernel clears again)
    for (uint32_t h0 = 0; h0 < h; h0 += R_B) {
        const uint32_t nb = std::min((uint32_t)R_B, h - h0);
        hipLaunchKernelGGL(k_r_seam_chains, dim3(1), dim3(64), 0, st, A, d_planes.p + h0, nb, eps, cos_t);
        hipLaunchKernelGGL(k_r_mark, dim3(std::min(VGRID_WGS, tiles)), dim3(TPB), 0, st, A, 0, (unsigned long long *)nullptr);
        hipLaunchKernelGGL(k_r_compact_raster, dim3(std::min(VGRID_WGS, loop_grid(tiles) * R_B)), dim3(TPB), 0, st, A, 0);
        hipLaunchKernelGGL(k_r_label, dim3(R_B), dim3(1024), 0, st, A, 0, 0);
        PlaneState hst[R_B];
        for (uint32_t b = 0; b < nb; ++b)
            HIP_TRY(hipMemcpyAsync(&hst[b], s.fixed.p + (size_t)b * F_BYTES + offsetof(ChainHdr, st), sizeof(PlaneState),
                                   hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipGetLastError());
        for (uint32_t b = 0; b < nb; ++b) {
            PLADE_REQUIRE(hst[b].err == 0, PLADE_EDEVICE, "plade_score_planes: compaction failed");
            PLADE_REQUIRE(hst[b].n_list == counts[h0 + b], PLADE_EDEVICE, "plade_score_planes: count/compaction disagreement");
            const uint32_t w = std::min(hst[b].n_list, cap);
            if (w)
                HIP_TRY(hipMemcpy(idx_out + (size_t)(h0 + b) * cap, s.var.p + (size_t)b * s.L.bytes + s.L.idxA, 4 * (size_t)w,
                                  hipMemcpyDeviceToHost));
        }
    }
}

void score_subset_seam(plade_ctx *ctx, RansacWork &W, const CloudDev &cloud, const int32_t *d_assigned, const uint32_t *sub_index,
                       uint32_t m, const float *planes, uint32_t h, float eps, float cos_t, uint32_t *counts, uint32_t *n_unassigned) {
    for (uint32_t j = 0; j < h; ++j) counts[j] = 0;
    if (n_unassigned) *n_unassigned = 0;
    if (cloud.n == 0 || h == 0 || m == 0) return;
    RArgs A = seam_args(ctx, W, cloud, d_assigned);
    RansacSlot &s = W.slot[0];
    hipStream_t st = ctx->stream;
    s.n_sub = m;
    s.sub_pitch = (m + 3) & ~3u;
    s.sub.ensure(6 * (size_t)s.sub_pitch + 4);
    s.sub_index.ensure((size_t)s.sub_pitch + 4);
    HIP_TRY(hipMemcpyAsync(s.sub_index.p, sub_index, 4 * (size_t)m, hipMemcpyHostToDevice, st));
    for (int g = 0; g < R_G; ++g) { A.c[g].sub = s.sub.p; A.c[g].sub_index = s.sub_index.p; A.c[g].sub_pitch = s.sub_pitch; A.c[g].n_sub = m; }
    args_commit(ctx, W, A);
    hipLaunchKernelGGL(k_r_seam_subset, dim3(cdiv(m, 256)), dim3(256), 0, st, A.c[0].cv, s.sub_index.p, m, s.sub.p, s.sub_pitch);
    DBuf<float4> d_planes;
    d_planes.ensure(h);
    HIP_TRY(hipMemcpyAsync(d_planes.p, planes, 16 * (size_t)h, hipMemcpyHostToDevice, st));
    // the hypotheses take the place of a sampling round's, R_H at a time (k_r_score_sub)
    for (uint32_t h0 = 0; h0 < h; h0 += R_H) {
        const uint32_t nh = std::min(R_H, h - h0);
        hipLaunchKernelGGL(k_r_seam_hyps, dim3(cdiv(R_H, 256)), dim3(256), 0, st, A, d_planes.p + h0, nh, eps, cos_t);
        hipLaunchKernelGGL(k_r_score_sub, dim3(std::min(VGRID_WGS, cdiv(m, TILE) * (R_H / HCHUNK))), dim3(TPB), 0, st, A);
        HIP_TRY(hipMemcpyAsync(counts + h0, A.c[0].hyp_counts, 4 * (size_t)nh, hipMemcpyDeviceToHost, st));
        if (n_unassigned)
            HIP_TRY(hipMemcpyAsync(n_unassigned, reinterpret_cast<const char *>(s.state.p) + offsetof(RState, sub_unassigned), 4,
                                   hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipGetLastError());
}

}  // namespace plade
