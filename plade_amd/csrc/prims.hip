// plade_amd/csrc/prims.hip -- device-wide sort / scan plumbing: the sorts are radix_sort.hip, the prefix sums the
// single-launch scan below.  No library device code: everything on the path is hand-written HIP.
#include "prims.h"
namespace plade {

static bool trace_sorts() { static const bool v = getenv("PLADE_TRACE_SORT") != nullptr; return v; }

// every sort of the path, whatever its size, is the hand-written onesweep radix sort of radix_sort.hip (a list of a few
// hundred items is one tile: a histogram launch + one launch per digit)
void sort_pairs_u32(plade_ctx *ctx, const uint32_t *ki, uint32_t *ko, const uint32_t *vi, uint32_t *vo, size_t n,
                    int bits) {
    if (trace_sorts()) fprintf(stderr, "[sort] u32 n %zu bits %d\n", n, bits);
    radix_sort_pairs_u32(ctx, ki, ko, vi, vo, n, bits);
}

void sort_pairs_u64(plade_ctx *ctx, const uint64_t *ki, uint64_t *ko, const uint32_t *vi, uint32_t *vo, size_t n,
                    int bits) {
    if (trace_sorts()) fprintf(stderr, "[sort] u64 n %zu bits %d\n", n, bits);
    radix_sort_pairs_u64(ctx, ki, ko, vi, vo, n, bits);
}

// ------------------------------------------------------------------------------------------------
// Exclusive prefix sum of u32 in ONE launch (decoupled look-back).  Tiles of 4096 items; a workgroup takes the next
// tile by ticket (every predecessor is already running), publishes its aggregate, then its inclusive prefix, as a
// 64-bit word (call generation | status | value): words of earlier calls never match the generation, so neither the
// words nor the ticket counter are ever reset (the ticket base of a call is the host's running total).
namespace {
constexpr int SC_T = 256, SC_I = 16, SC_TILE = SC_T * SC_I;
constexpr uint64_t SC_AGG = 1ull << 32, SC_PREFIX = 2ull << 32, SC_STATUS = 3ull << 32;

__device__ void k_scan_u32(const VB &, const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n,
                           uint64_t *__restrict__ state, uint32_t *__restrict__ ticket, uint32_t base, uint32_t gen) {
    __shared__ uint32_t s_tile, s_w[SC_T / 64], s_excl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u) - base;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t first = tile * SC_TILE + tid * SC_I;
    uint32_t v[SC_I];
    if (first + SC_I <= n) {
#pragma unroll
        for (int q = 0; q < SC_I / 4; ++q) {
            const uint4 t = *reinterpret_cast<const uint4 *>(in + first + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < SC_I; ++q) v[q] = first + q < n ? in[first + q] : 0u;
    }
    uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < SC_I; ++q) { const uint32_t t = v[q]; v[q] = sum; sum += t; }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t agg = 0, woff = 0;
    for (int w = 0; w < SC_T / 64; ++w) { if (w < wave) woff += s_w[w]; agg += s_w[w]; }
    const uint64_t tag = (uint64_t)gen << 34;
    if (wave == 0) {
        if (lane == 0)
            __hip_atomic_store(state + tile, tag | (tile == 0 ? SC_PREFIX : SC_AGG) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        if (tile > 0) {
            int t = (int)tile - 1;
            for (;;) {   // a window of 64 predecessors per round
                const int idx = t - lane;
                uint64_t st = tag | SC_PREFIX;   // before the first tile: prefix 0
                if (idx >= 0) st = __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ready = (st >> 34) == gen && (st & SC_STATUS) != 0ull;
                const unsigned long long not_ready = __ballot(!ready), is_prefix = __ballot(ready && (st & SC_PREFIX));
                const int first_bad = not_ready ? __ffsll((long long)not_ready) - 1 : 64;
                const int first_pre = is_prefix ? __ffsll((long long)is_prefix) - 1 : 64;
                const int take = first_pre < first_bad ? first_pre + 1 : first_bad;   // lanes [0, take) are consumed
                uint32_t part = lane < take ? (uint32_t)st : 0u;
                for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
                excl += part;
                if (first_pre < first_bad) break;
                t -= take;
            }
            if (lane == 0)
                __hip_atomic_store(state + tile, tag | SC_PREFIX | (uint64_t)(excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_excl = excl;
    }
    __syncthreads();
    const uint32_t off = s_excl + woff + incl - sum;
    if (first + SC_I <= n) {
#pragma unroll
        for (int q = 0; q < SC_I / 4; ++q)
            *reinterpret_cast<uint4 *>(out + first + 4 * q) = make_uint4(off + v[4 * q], off + v[4 * q + 1], off + v[4 * q + 2], off + v[4 * q + 3]);
    } else {
#pragma unroll
        for (int q = 0; q < SC_I; ++q)
            if (first + q < n) out[first + q] = off + v[q];
    }
}
}  // namespace

ScanTicket scan_ticket(plade_ctx *ctx, size_t n, uint32_t tile_items) {
    ScanWork &w = ctx->scan;
    const uint32_t tiles = cdiv(n, tile_items);
    if (w.state.cap < tiles + 1 || !w.ticket.p) {   // fresh words must not look like this generation's
        w.state.ensure((size_t)tiles + 1);
        w.ticket.ensure(4);
        ctx->fill_async(w.state.p, 0, w.state.cap * 8);
        ctx->fill_async(w.ticket.p, 0, 16);
        w.base = 0;
    }
    ScanTicket t{w.state.p, w.ticket.p, w.base, ++w.gen & 0x3fffffffu, tiles};
    if (t.gen == 0) t.gen = w.gen = 1;
    w.base += tiles;
    return t;
}

void exclusive_scan_u32(plade_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n) {
    if (!n) return;
    PLADE_REQUIRE(n < (1ull << 32), PLADE_ELIMIT, "scan: too many items");
    const ScanTicket t = scan_ticket(ctx, n, SC_TILE);
    launch<k_scan_u32, SC_T>(ctx, dim3(t.tiles), 0, in, out, (uint32_t)n, t.state, t.ticket, t.base, t.gen);
}


// ---- device -> host hand-over (ctx.h: plade_ctx::d2h / sync) ----------------------------------------------------------
// grid (chunks, ranges): workgroup (x, r) copies words x * 1024 .. of range r into the arena; the workgroup that finishes
// last raises the flag.  Every workgroup makes its stores visible to the host before it takes its ticket, so when the last
// ticket is drawn all data are out.
namespace {
__global__ __launch_bounds__(256) void k_copy_out(const CopyOutArgs a, uint32_t *__restrict__ arena, uint32_t *__restrict__ counter,
                                                  uint32_t *flag, uint32_t seq) {
    const uint32_t r = blockIdx.y;
    const uint32_t n = a.words[r];
    const uint32_t *__restrict__ src = a.src[r];
    uint32_t *__restrict__ dst = arena + a.off_words[r];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(counter, 1u);
        if (t == gridDim.x * gridDim.y - 1u) {
            atomicExch(counter, 0u);
            if (flag) {
                __threadfence_system();
                __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
}  // namespace

void copy_out_launch(hipStream_t st, const CopyOutArgs &a, uint32_t *arena_words, uint32_t *counter, uint32_t *flag, uint32_t seq) {
    uint32_t most = 1;
    for (uint32_t q = 0; q < a.n; ++q) most = std::max(most, a.words[q]);
    const uint32_t gx = std::min(64u, cdiv(most, 1024u));   // a workgroup moves 4 KB per round
    hipLaunchKernelGGL(k_copy_out, dim3(gx, a.n), dim3(256), 0, st, a, arena_words, counter, flag, seq);
    HIP_TRY(hipGetLastError());
}

}  // namespace plade
