// plade_amd/csrc/radix_sort.hip -- stable LSD radix sort of (key, u32 value) pairs for gfx950, one launch per
// 8-bit digit plus one histogram launch and NO memsets.
//
// Every grid of the path (Morton order, the four voxel grids, the verification / penetration / clustering grids)
// is built by sorting ~1M packed cell keys whose significant bit count is known.  rocPRIM's onesweep sort costs
// 12-15 stream commands for such a sort (a fill of the histogram, a fill of the look-back states and a fill of the
// tile counter per digit, the histogram, its scan, the passes); with ~14 sorts per registration that was a fifth of
// the GPU time of a registration and an eighth of its commands.  Here:
//   k_rs_histogram  128 workgroups count all digits of all passes in LDS and add their counts to ONE global histogram
//                   (integer atomics: order-independent), and clear the look-back states and tile counters of every
//                   pass.  The histogram lives in one of two buffers of the context that take turns from sort to sort:
//                   the passes of a sort zero the buffer of the NEXT one, so nothing has to be zeroed by a command.
//                   (r3 wrote per-workgroup partial histograms and every tile of every pass summed all 128 of them per
//                   digit: 256 KB of loads per tile, twice the traffic of the keys themselves.)
//   k_rs_pass       "onesweep": a workgroup takes the next tile (atomic ticket, so every predecessor is already
//                   running), ranks its keys (512 lanes x 8 or 16) by digit (wave-level match via 8 ballots per key: stable),
//                   publishes its digit counts, resolves its exclusive prefix over the preceding tiles (below), stages the
//                   tile in LDS in digit order and writes runs of equal digits to their final place.
//   prefix          TWO levels instead of a chained look-back.  The tiles form supertiles of ST ~ sqrt(#tiles) tiles; a tile
//                   publishes its counts (one word per digit) and adds them to its supertile's word (a 64-bit atomic add of
//                   count | 1 << 40: the high part counts the tiles that have contributed).  Its exclusive prefix is the sum
//                   of the COMPLETE supertiles before its own plus the counts of the tiles before it inside its own:
//                   <= #tiles / ST + ST independent loads, issued together and polled until valid.  r3's chained look-back
//                   (windows of 8 predecessors, aggregate or inclusive prefix) needed ~#tiles / 16 dependent round trips
//                   for the last tiles when all tiles of a sort run at once -- 15 trips of 1-2 us past the L2s for the 244
//                   tiles of a 1M-key sort, i.e. most of a 30 us pass.
// Algorithmic traffic per pass: read n x (sizeof(K) + 4) B, write the same.
#include "prims.h"

// gfx950 only: k_rs_pass keeps a whole 8192-pair tile in LDS (up to ~116 KB of the CU's 160 KB for 64-bit keys).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "radix_sort.hip is written for gfx950 (160 KB LDS per workgroup); build with --offload-arch=gfx950"
#endif

namespace plade {

namespace {

constexpr int RS_THREADS = 512;
constexpr int RS_WAVES = RS_THREADS / 64;
// keys per thread (template parameter IPT of the pass kernel): 8 up to 1.5 M items -- 4096-key tiles, twice the workgroups,
// 1M items then fill the 256 CUs once (u32 pass 25.7 -> 20.8 us, u64 34.6 -> 31.8 under load) -- and 16 above (the 2M-item
// Morton sort of a pair is faster with 8192-key tiles: 51 vs 54 us)
constexpr int RS_HBLOCKS = 128;                  // histogram workgroups (= partial histograms per digit place)
constexpr int RS_MAXP = 8;
constexpr int RS_LOOK = 8;                      // states fetched per round of the prefix loads
constexpr uint32_t RS_VALID = 1u << 31, RS_VALUE = RS_VALID - 1;
// One launch sequence can sort up to RS_MAXSEG independent ARRAYS ("segments": consecutive ranges of the same key / value
// buffers, each sorted on its own and left in its own range): the Morton order of the sixteen clouds of a group is 4 launches
// instead of 64.  A segment has its own global histogram, tiles, look-back words and supertile sums -- sixteen independent prefix
// chains in one grid, not one chain sixteen times as long (a single sort over the keys of all clouds was measured at 426 us per
// pass under load against ~25 per 1M-key cloud: the more tiles one chain has, the longer each of them waits).
constexpr int RS_MAXSEG = 16;
constexpr uint32_t RS_GH_SEG = RS_MAXP * 512;   // words of global histogram per segment
struct RSSegs {
    uint32_t nseg;
    uint32_t off[RS_MAXSEG + 1];          // segment s = items [off[s], off[s + 1])
    uint32_t tile_start[RS_MAXSEG + 1];   // first tile of segment s in the pass grid
    uint32_t sup_start[RS_MAXSEG + 1];    // first supertile word group of segment s
};
__device__ __forceinline__ uint32_t seg_of_tile(const RSSegs &S, uint32_t tile) {
    uint32_t s = 0;
#pragma unroll
    for (int q = 1; q < RS_MAXSEG; ++q) s += (q < (int)S.nseg && tile >= S.tile_start[q]) ? 1u : 0u;
    return s;
}
constexpr unsigned long long RS_SUP_ONE = 1ull << 40, RS_SUP_VALUE = RS_SUP_ONE - 1ull;

// Digits are 8 bits wide, or 9 where that saves a pass (25-27, 17-18 significant bits: the voxel and cell grids):
// 512 bins are one per lane of the 512-lane workgroup.
template <int DB, class K>
__device__ __forceinline__ uint32_t digit_of(K key, int shift) { return (uint32_t)(key >> shift) & ((1u << DB) - 1u); }

// ghist[p][d]: keys of digit d at place p (all-zero on entry)
template <class K, int DB>
__device__ void k_rs_histogram(const VB &vb, const K *__restrict__ keys_all, const RSSegs &S, int passes,
                               uint32_t *__restrict__ ghist_all, uint32_t *__restrict__ tile_ctr,
                               uint32_t *__restrict__ look, size_t look_words) {
    constexpr int NB = 1 << DB;
    __shared__ uint32_t s_h[RS_MAXP * NB];
    for (int i = threadIdx.x; i < passes * NB; i += RS_THREADS) s_h[i] = 0;
    // clear what the passes will use
    for (size_t i = (size_t)vb.bx * RS_THREADS + threadIdx.x; i < look_words; i += (size_t)vb.gx * RS_THREADS) look[i] = 0;
    if (vb.bx == 0 && threadIdx.x < RS_MAXP) tile_ctr[threadIdx.x] = 0;
    __syncthreads();
    // RS_HBLOCKS workgroups per segment
    const uint32_t seg = vb.bx / RS_HBLOCKS, hb = vb.bx % RS_HBLOCKS;
    const K *__restrict__ keys = keys_all + S.off[seg];
    uint32_t *__restrict__ ghist = ghist_all + (size_t)seg * RS_GH_SEG;
    const uint32_t n = S.off[seg + 1] - S.off[seg];
    const uint32_t per = (n + RS_HBLOCKS - 1) / RS_HBLOCKS;
    const uint32_t b0 = hb * per, b1 = min(n, b0 + per);
    // 8 independent loads per round: a one-load-per-iteration loop is bound by the load latency (39 us at 1M keys)
    for (uint32_t i0 = b0; i0 < b1; i0 += 8 * RS_THREADS) {
        K k[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t i = i0 + j * RS_THREADS + threadIdx.x;
            k[j] = i < b1 ? keys[i] : (K)0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (i0 + j * RS_THREADS + threadIdx.x < b1)
                for (int p = 0; p < passes; ++p) atomicAdd(&s_h[p * NB + digit_of<DB>(k[j], DB * p)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < passes * NB; i += RS_THREADS) { const uint32_t c = s_h[i]; if (c) atomicAdd(&ghist[i], c); }
}

template <class K, int DB, int RS_IPT>
__device__ void k_rs_pass(const VB &, const K *__restrict__ kin_all, K *__restrict__ kout_all,
                          const uint32_t *__restrict__ vin_all, uint32_t *__restrict__ vout_all,
                          const RSSegs &S, int pass, const uint32_t *__restrict__ ghist_all,
                          uint32_t *__restrict__ ghist_next, uint32_t zero_words, uint32_t *__restrict__ tile_ctr,
                          uint32_t *__restrict__ look_all, unsigned long long *__restrict__ sup_all, uint32_t st_shift) {
    constexpr int NB = 1 << DB;
    constexpr int RS_TILE = RS_THREADS * RS_IPT;
    static_assert(NB <= RS_THREADS, "one lane per digit");
    static_assert(RS_TILE * (sizeof(K) + 4) + (RS_WAVES + 2) * NB * 4 + 64 <= 160 * 1024, "tile does not fit gfx950's 160 KB LDS");
    __shared__ K s_keys[RS_TILE];
    __shared__ uint32_t s_vals[RS_TILE];
    __shared__ uint32_t s_cnt[RS_WAVES][NB];    // per wave: digit counters, then exclusive offsets inside the digit
    __shared__ uint32_t s_start[NB];            // first tile-local position of digit d
    __shared__ uint32_t s_dest[NB];             // global position of the tile's first key of digit d
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int shift = DB * pass;
    if (tid == 0) s_tile = atomicAdd(&tile_ctr[pass], 1u);
    for (int i = tid; i < RS_WAVES * NB; i += RS_THREADS) (&s_cnt[0][0])[i] = 0;
    // threads 0..NB-1 own one digit each.  Keys of that digit in the whole array: the global histogram (independent of the
    // tile: issued first, the latency hides behind the ticket and the key loads)
    const bool owner = tid < NB;
    __syncthreads();
    // ticket -> (segment, tile inside the segment): tickets are handed out in grid order, so every tile of a segment that
    // precedes this one holds an earlier ticket and is running
    const uint32_t gtile = s_tile;
    const uint32_t seg = seg_of_tile(S, gtile), tile = gtile - S.tile_start[seg];
    const uint32_t n = S.off[seg + 1] - S.off[seg];
    const K *__restrict__ kin = kin_all + S.off[seg];
    K *__restrict__ kout = kout_all + S.off[seg];
    const uint32_t *__restrict__ vin = vin_all + S.off[seg];
    uint32_t *__restrict__ vout = vout_all + S.off[seg];
    const uint32_t *__restrict__ ghist = ghist_all + (size_t)seg * RS_GH_SEG;
    uint32_t *__restrict__ look = look_all + (size_t)S.tile_start[seg] * NB;
    unsigned long long *__restrict__ sup = sup_all + (size_t)S.sup_start[seg] * NB;
    // keys of this thread's digit in the whole segment: the global histogram
    const uint32_t total = owner ? ghist[pass * NB + tid] : 0u;
    // the first tile of the first pass leaves the OTHER histogram buffer all-zero for the next sort of this context (its last
    // user, the previous sort on this stream, has finished): as many words as that sort's segments touched
    if (pass == 0 && gtile == 0)
        for (uint32_t i = tid; i < zero_words; i += RS_THREADS) ghist_next[i] = 0u;
    const uint32_t base = tile * RS_TILE + wave * (64 * RS_IPT);   // this wave's 1024 consecutive keys

    // ---- load + stable rank inside the wave ------------------------------------------------------
    K key[RS_IPT];
    uint32_t val[RS_IPT], rank[RS_IPT];
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        const uint32_t i = base + r * 64 + lane;
        const bool ok = i < n;
        // (each key and value is read once per pass: non-temporal, so that the streams do not evict the look-back words, the
        //  histograms and the other groups' tables from L2 / the Infinity Cache)
        key[r] = ok ? __builtin_nontemporal_load(kin + i) : (K)0;
        val[r] = ok ? __builtin_nontemporal_load(vin + i) : 0u;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    volatile uint32_t *cnt = s_cnt[wave];
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        const bool ok = base + r * 64 + lane < n;
        const uint32_t d = digit_of<DB>(key[r], shift);
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        // all lanes of a digit group read the counter, then its first lane advances it (one wave = in order)
        const uint32_t c = ok ? cnt[d] : 0u;
        rank[r] = c + (uint32_t)__popcll(peers & lt);
        if (ok && (peers & lt) == 0ull) cnt[d] = c + (uint32_t)__popcll(peers);
    }
    __syncthreads();

    // ---- per digit (thread d): wave offsets, tile count, global base, look-back -----------------------------
    const uint32_t d = tid & (NB - 1);
    uint32_t tile_count = 0;
    if (owner)
        for (int w = 0; w < RS_WAVES; ++w) { const uint32_t c = s_cnt[w][d]; s_cnt[w][d] = tile_count; tile_count += c; }
    // publish: this tile's counts, and their share in the supertile's sum
    if (owner) {
        __hip_atomic_store(look + (size_t)tile * NB + d, RS_VALID | tile_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(sup + (size_t)(tile >> st_shift) * NB + d, RS_SUP_ONE | (unsigned long long)tile_count, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    // block-wide exclusive scans of `total` (global) and `tile_count` (tile-local) over the NB digits
    uint32_t inc_g = total, inc_t = tile_count;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t a = __shfl_up(inc_g, o, 64), b = __shfl_up(inc_t, o, 64);
        if (lane >= o) { inc_g += a; inc_t += b; }
    }
    __shared__ uint32_t s_wg[NB / 64], s_wt[NB / 64];
    if (owner && lane == 63) { s_wg[wave] = inc_g; s_wt[wave] = inc_t; }
    __syncthreads();
    if (owner) {
        uint32_t off_g = 0, off_t = 0;
        for (int w = 0; w < wave; ++w) { off_g += s_wg[w]; off_t += s_wt[w]; }
        const uint32_t gbase = off_g + inc_g - total;       // exclusive: keys with a smaller digit in the whole array
        const uint32_t tstart = off_t + inc_t - tile_count; // exclusive
        // exclusive prefix of this digit over the preceding tiles: complete supertiles, then the tiles of the own one.
        // Agent-scope loads go past the per-XCD L2; RS_LOOK of them are in flight together, a word that is not there yet
        // (its tiles hold earlier tickets, so they are running) is polled.
        uint32_t excl = 0;
        const uint32_t my_sup = tile >> st_shift, st_tiles = 1u << st_shift;
        for (uint32_t s0 = 0; s0 < my_sup; s0 += RS_LOOK) {
            unsigned long long w[RS_LOOK];
#pragma unroll
            for (int j = 0; j < RS_LOOK; ++j)
                w[j] = (s0 + j < my_sup) ? __hip_atomic_load(sup + (size_t)(s0 + j) * NB + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                         : ((unsigned long long)st_tiles << 40);
#pragma unroll
            for (int j = 0; j < RS_LOOK; ++j) {
                while ((w[j] >> 40) != st_tiles) w[j] = __hip_atomic_load(sup + (size_t)(s0 + j) * NB + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                excl += (uint32_t)(w[j] & RS_SUP_VALUE);
            }
        }
        for (uint32_t t0 = my_sup << st_shift; t0 < tile; t0 += RS_LOOK) {
            uint32_t w[RS_LOOK];
#pragma unroll
            for (int j = 0; j < RS_LOOK; ++j)
                w[j] = (t0 + j < tile) ? __hip_atomic_load(look + (size_t)(t0 + j) * NB + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : RS_VALID;
#pragma unroll
            for (int j = 0; j < RS_LOOK; ++j) {
                while (!(w[j] & RS_VALID)) w[j] = __hip_atomic_load(look + (size_t)(t0 + j) * NB + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                excl += w[j] & RS_VALUE;
            }
        }
        s_start[d] = tstart;
        s_dest[d] = gbase + excl;
    }
    __syncthreads();

    // ---- stage the tile in digit order, then write runs ------------------------------------------------------
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        if (base + r * 64 + lane < n) {
            const uint32_t dg = digit_of<DB>(key[r], shift);
            const uint32_t pos = s_start[dg] + s_cnt[wave][dg] + rank[r];
            s_keys[pos] = key[r];
            s_vals[pos] = val[r];
        }
    }
    __syncthreads();
    const uint32_t in_tile = min((uint32_t)RS_TILE, n - tile * RS_TILE);
    for (uint32_t pos = tid; pos < in_tile; pos += RS_THREADS) {
        const K k = s_keys[pos];
        const uint32_t dg = digit_of<DB>(k, shift);
        const uint32_t out = s_dest[dg] + (pos - s_start[dg]);
        kout[out] = k;
        vout[out] = s_vals[pos];
    }
}

template <class K, int DB, int RS_IPT>
void radix_sort_run_ipt(plade_ctx *ctx, const K *ki, K *ko, const uint32_t *vi, uint32_t *vo, const uint32_t *seg_off, int nseg, int passes) {
    constexpr int NB = 1 << DB;
    constexpr int RS_TILE = RS_THREADS * RS_IPT;
    const size_t n = seg_off[nseg] - seg_off[0];
    if (n == 0) return;   // no tiles: nothing to launch
    RSSegs S;
    memset(&S, 0, sizeof(S));
    S.nseg = (uint32_t)nseg;
    uint32_t max_tiles = 1;
    for (int q = 0; q <= RS_MAXSEG; ++q) S.off[q] = seg_off[std::min(q, nseg)] - seg_off[0];
    for (int q = 0; q < nseg; ++q) {
        const uint32_t t = cdiv(S.off[q + 1] - S.off[q], RS_TILE);
        S.tile_start[q + 1] = S.tile_start[q] + t;
        max_tiles = std::max(max_tiles, t);
    }
    for (int q = nseg; q < RS_MAXSEG; ++q) S.tile_start[q + 1] = S.tile_start[q];
    const uint32_t tiles = S.tile_start[nseg];
    // supertiles of 2^st_shift ~ sqrt(tiles of a segment) tiles; per pass: one u32 per (tile, digit), then one u64 per (supertile, digit)
    uint32_t st_shift = 2;
    while ((1u << (2 * st_shift)) < max_tiles && st_shift < 8) ++st_shift;
    for (int q = 0; q < RS_MAXSEG; ++q)
        S.sup_start[q + 1] = S.sup_start[q] + (q < nseg ? ((S.tile_start[q + 1] - S.tile_start[q]) >> st_shift) + 1 : 0u);
    const uint32_t nsup = S.sup_start[nseg];
    const size_t pass_words = (size_t)tiles * NB + 2 * (size_t)nsup * NB, look_words = (size_t)passes * pass_words;
    // scratch: tile counters | look-back states | key ping buffer | value ping buffer
    const size_t off_ctr = 0, off_look = off_ctr + 64, off_keys = (off_look + look_words + 3) & ~(size_t)3;
    const size_t key_words = (n * sizeof(K) + 3) / 4, off_vals = (off_keys + key_words + 3) & ~(size_t)3;
    uint32_t *t = reinterpret_cast<uint32_t *>(ctx->scratch[7].ensure((off_vals + n + 64) * 4 + 256));
    K *tk = reinterpret_cast<K *>(t + off_keys);
    uint32_t *tv = t + off_vals;
    hipStream_t st = ctx->stream;
    // the two global histograms of this context (see the header): zeroed once, then by the sorts themselves -- every sort
    // zeroes, in the buffer of the NEXT sort, what the sort before it left there
    constexpr size_t GH = (size_t)RS_MAXSEG * RS_GH_SEG;
    if (!ctx->sort_ghist.p) {
        ctx->sort_ghist.ensure(2 * GH);
        ctx->fill_async(ctx->sort_ghist.p, 0, 2 * GH * 4);
        ctx->sort_segs[0] = ctx->sort_segs[1] = 0;
    }
    const uint32_t cur = ctx->sort_seq & 1u, nxt = cur ^ 1u;
    uint32_t *gh = ctx->sort_ghist.p + cur * GH, *gh_next = ctx->sort_ghist.p + nxt * GH;
    const uint32_t zero_words = ctx->sort_segs[nxt] * RS_GH_SEG;   // what the previous sort dirtied in the other buffer
    ctx->sort_segs[nxt] = 0;
    ctx->sort_segs[cur] = (uint32_t)nseg;
    ctx->sort_seq += 1;
    ki += seg_off[0]; ko += seg_off[0]; vi += seg_off[0]; vo += seg_off[0];
    ctx->ev_begin("sort_hist", (double)n * sizeof(K));
    launch<k_rs_histogram<K, DB>, RS_THREADS>(ctx, dim3(RS_HBLOCKS * nseg), 0, ki, S, passes, gh, t + off_ctr,
                       t + off_look, look_words);
    ctx->ev_end();
    const K *src_k = ki;
    const uint32_t *src_v = vi;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) & 1) == 0;    // the last pass lands in the caller's output
        K *dst_k = to_out ? ko : tk;
        uint32_t *dst_v = to_out ? vo : tv;
        uint32_t *pass_look = t + off_look + (size_t)p * pass_words;
        ctx->ev_begin("sort_pass", (double)n * 2.0 * (sizeof(K) + 4));   // algorithmic: keys + values read once, written once
        launch<k_rs_pass<K, DB, RS_IPT>, RS_THREADS>(ctx, dim3(tiles), 0, src_k, dst_k, src_v, dst_v, S, p, gh,
                           gh_next, zero_words, t + off_ctr, pass_look, reinterpret_cast<unsigned long long *>(pass_look + (size_t)tiles * NB), st_shift);
        ctx->ev_end();
        src_k = dst_k;
        src_v = dst_v;
    }
    HIP_TRY(hipGetLastError());
}

template <class K, int DB>
void radix_sort_run(plade_ctx *ctx, const K *ki, K *ko, const uint32_t *vi, uint32_t *vo, const uint32_t *seg_off, int nseg, int passes) {
    uint32_t largest = 0;
    for (int q = 0; q < nseg; ++q) largest = std::max(largest, seg_off[q + 1] - seg_off[q]);
    if (largest <= 3000000) radix_sort_run_ipt<K, DB, 8>(ctx, ki, ko, vi, vo, seg_off, nseg, passes);
    else radix_sort_run_ipt<K, DB, 16>(ctx, ki, ko, vi, vo, seg_off, nseg, passes);
}

template <class K>
void radix_sort_impl(plade_ctx *ctx, const K *ki, K *ko, const uint32_t *vi, uint32_t *vo, const uint32_t *seg_off, int nseg, int bits) {
    PLADE_REQUIRE(nseg >= 1 && nseg <= RS_MAXSEG, PLADE_EINVAL, "sort: one to sixteen segments");
    for (int q = 0; q < nseg; ++q) PLADE_REQUIRE(seg_off[q] <= seg_off[q + 1], PLADE_EINVAL, "sort: segment offsets must ascend");
    PLADE_REQUIRE((uint64_t)seg_off[nseg] - seg_off[0] < (1ull << 30), PLADE_ELIMIT, "sort: too many items");
    PLADE_REQUIRE(bits >= 1 && bits <= (int)sizeof(K) * 8, PLADE_EINVAL, "sort: bit range");
    const int p8 = (bits + 7) / 8, p9 = (bits + 8) / 9;
    static const bool no9 = getenv("PLADE_SORT_DIGIT8") != nullptr;
    // the last 9-bit digit must still lie inside the key: (p9 - 1) * 9 < key bits
    if (p9 < p8 && !no9 && (p9 - 1) * 9 < (int)sizeof(K) * 8) radix_sort_run<K, 9>(ctx, ki, ko, vi, vo, seg_off, nseg, p9);
    else {
        PLADE_REQUIRE(p8 <= RS_MAXP, PLADE_EINVAL, "sort: bit range");
        radix_sort_run<K, 8>(ctx, ki, ko, vi, vo, seg_off, nseg, p8);
    }
}

}  // namespace

void radix_sort_pairs_u32(plade_ctx *ctx, const uint32_t *ki, uint32_t *ko, const uint32_t *vi, uint32_t *vo, size_t n, int bits) {
    PLADE_REQUIRE(n < (1ull << 30), PLADE_ELIMIT, "sort: too many items");
    const uint32_t off[2] = {0u, (uint32_t)n};
    radix_sort_impl<uint32_t>(ctx, ki, ko, vi, vo, off, 1, bits);
}
void radix_sort_pairs_u64(plade_ctx *ctx, const uint64_t *ki, uint64_t *ko, const uint32_t *vi, uint32_t *vo, size_t n, int bits) {
    PLADE_REQUIRE(n < (1ull << 30), PLADE_ELIMIT, "sort: too many items");
    const uint32_t off[2] = {0u, (uint32_t)n};
    radix_sort_impl<uint64_t>(ctx, ki, ko, vi, vo, off, 1, bits);
}
void radix_sort_segments_u32(plade_ctx *ctx, const uint32_t *ki, uint32_t *ko, const uint32_t *vi, uint32_t *vo, const uint32_t *seg_off,
                             int nseg, int bits) {
    radix_sort_impl<uint32_t>(ctx, ki, ko, vi, vo, seg_off, nseg, bits);
}

}  // namespace plade
