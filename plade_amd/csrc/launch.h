// plade_amd/csrc/launch.h -- how the stages behind the plane extraction reach the GPU: plade::launch<kernel, TPB>(ctx, grid, ...).
//
// Alone, a registration queues its ~60 small kernels on its own stream, one launch each.  In batch mode the pairs of a GROUP
// (plade_registration_pairs, up to eight) run the same stages at the same time, every one of them a chain of short,
// latency-bound kernels that holds one of the GPU's four hardware queues while it uses a fraction of a percent of the part
// (profiles/r4_experiments.md: every millisecond of such queue time per registration costs 14 % of the throughput).  So the
// pairs of a group are carried through those stages IN LOCK STEP: their host threads keep running the per-pair code
// unchanged, but what they launch is not queued at once -- a Combiner (one per group call) collects it, and when every
// pair of the group has reached its next host wait, ONE thread merges the eight queues: launches of the same kernel become
// one launch whose grid is the concatenation of the pairs' grids (each workgroup finds its pair, its place in that pair's
// grid and that pair's arguments in the kernel arguments), the small uploads become one copy kernel, the fills one fill
// kernel, the read-backs of all pairs one hand-over kernel, and there is one wait for the group instead of one per pair.
// Same kernels, same arguments, same arithmetic: every pair's result stays the bits of the pair alone
// (tests/test_gpu_groups.py).  Reference loop being batched: code/PLADE/main.cpp:122-148, stages plade.cpp:258-575.
//
// A kernel of these stages is written as a __device__ function whose first parameter is its place in ITS launch:
//     __device__ void k_foo(const VB &vb, A a, B b)     (vb.bx / vb.by = blockIdx.x / .y, vb.gx / vb.gy = gridDim.x / .y of
//                                                        the launch the pair asked for)
// and launched with launch<k_foo, TPB>(ctx, dim3(gx, gy), shared_bytes, a, b).  Without a Combiner on the context that is
// one ordinary launch on ctx->stream.
#pragma once
#include "common.h"
#include <condition_variable>
#include <functional>
#include <mutex>
#include <type_traits>

struct plade_ctx;

namespace plade {

struct VB { uint32_t bx, by, gx, gy; };

constexpr int BATCH_MAX = 8;                 // pairs of a group (PLADE_GROUP_MAX)
constexpr size_t PACK_MAX = 464;             // bytes of kernel arguments per pair: 8 of them + the header fit the 4 KB of a launch

template <class... A> struct Pack;
template <> struct Pack<> {};
template <class H, class... T> struct Pack<H, T...> { H head; Pack<T...> tail; };

template <class... A> inline Pack<A...> make_pack(A... a);
template <> inline Pack<> make_pack<>() { return Pack<>{}; }
template <class H, class... T> inline Pack<H, T...> make_pack_impl(H h, T... t) { Pack<H, T...> p; p.head = h; p.tail = make_pack<T...>(t...); return p; }
template <class... A> inline Pack<A...> make_pack(A... a) { return make_pack_impl<A...>(a...); }

// A kernel may take a larger struct as `const T &`: it is stored by value in the pack and the reference is bound to it THERE
// (in the kernel-argument segment), so indexing it with a run-time index stays a scalar load instead of a copy in scratch.
template <class F> struct BodyTraits;
template <class... A> struct BodyTraits<void (*)(const VB &, A...)> {
    using pack = Pack<std::decay_t<A>...>;
    static pack make(std::decay_t<A>... a) { return make_pack<std::decay_t<A>...>(a...); }
};

struct BatchHdr { uint32_t n, start[BATCH_MAX + 1], gx[BATCH_MAX], gy[BATCH_MAX]; };
template <class P, int N> struct BatchPacks { P p[N]; };

#ifdef __HIPCC__
template <auto Body, class... Done>
__device__ __forceinline__ void call_body(const VB &vb, const Pack<> &, const Done &...d) { Body(vb, d...); }
template <auto Body, class H, class... T, class... Done>
__device__ __forceinline__ void call_body(const VB &vb, const Pack<H, T...> &p, const Done &...d) { call_body<Body>(vb, p.tail, d..., p.head); }

// one launch for up to N pairs: workgroup blockIdx.x belongs to the pair e with start[e] <= blockIdx.x < start[e + 1]
// (MINW: the second argument of __launch_bounds__, wavefronts per SIMD the register allocation must leave room for)
template <auto Body, int TPB, int N, class P, int MINW = 1>
__global__ __launch_bounds__(TPB, MINW) void k_batch(const BatchHdr h, const BatchPacks<P, N> packs) {
    uint32_t e = 0;
    if (N > 1) {
#pragma unroll
        for (int q = 1; q < N; ++q) e += (q < (int)h.n && blockIdx.x >= h.start[q]) ? 1u : 0u;
    }
    const uint32_t lin = blockIdx.x - h.start[e], gx = h.gx[e];
    VB vb;
    vb.bx = lin % gx; vb.by = lin / gx; vb.gx = gx; vb.gy = h.gy[e];
    call_body<Body>(vb, packs.p[e]);
}
#endif

// ---- the queue of one pair --------------------------------------------------------------------------------------------------
struct QEntry {
    enum Kind : uint8_t { KERNEL, COPY_IN, FILL, FUNC } kind = KERNEL;
    const void *id = nullptr;                // KERNEL: identifies the instantiation (entries with the same id merge)
    void (*launch_many)(hipStream_t, int, QEntry *const *) = nullptr;
    uint32_t gx = 0, gy = 0, smem = 0;
    // COPY_IN: words from `src` (page-locked staging memory, as the device addresses it) to `dst`; FILL: `words` words of `value` at `dst`
    void *dst = nullptr; const void *src = nullptr; uint32_t words = 0, value = 0;
    std::function<void(hipStream_t)> fn;     // FUNC: anything else, issued on the group's stream at its place in the pair's order
    alignas(16) unsigned char pack[PACK_MAX];
};

// One per call of a group of several pairs (registration.hip).  Members = the pair contexts taking part; `lead` = the context whose
// stream carries everything.
struct Combiner {
    plade_ctx *lead = nullptr;
    std::mutex m;
    std::condition_variable cv;
    int members = 0, arrived = 0;
    uint64_t epoch = 0;
    std::string error;                        // a failed flush fails every member's wait
    int error_code = 0;
    plade_ctx *member[BATCH_MAX] = {};
    std::vector<QEntry> q[BATCH_MAX];
    // statistics of the call (registration.hip adds them to the lead's stats)
    uint64_t asked[BATCH_MAX] = {}, launches_issued = 0, waits = 0;   // operations the pairs queued / commands that reached the stream / group waits

    int join(plade_ctx *c);                   // -> slot
    void leave(plade_ctx *c);                 // the pair is done (or gave up): whatever it still has queued is issued
    void wait(plade_ctx *c);                  // the pair's host wait: returns when everything it queued has run and its read-backs are in
    QEntry &push(plade_ctx *c);
    int size() { std::lock_guard<std::mutex> lk(m); return members; }   // pairs taking part right now (a pair that fails leaves early)
    void bury(std::vector<void *> &gy);       // device allocations to free when the call has ended (combiner.hip)
    ~Combiner();
    const char *trace_stage = "";
    std::chrono::steady_clock::time_point t_last_out{};   // PLADE_TRACE_LOCKSTEP: when the previous group wait returned
private:
    std::vector<void *> graveyard;
    void flush_locked(std::unique_lock<std::mutex> &lk);
};

int combiner_slot(plade_ctx *c);
hipStream_t ctx_stream(plade_ctx *c);
Combiner *ctx_combiner(plade_ctx *c);

#ifdef __HIPCC__
template <auto Body, int TPB, class P, int MINW>
void launch_many_impl(hipStream_t st, int n, QEntry *const *es) {
    constexpr int N = sizeof(P) <= PACK_MAX ? BATCH_MAX : 1;
    for (int b = 0; b < n; b += N) {
        const int k = std::min(N, n - b);
        BatchHdr h;
        memset(&h, 0, sizeof(h));
        BatchPacks<P, N> packs;
        memset(static_cast<void *>(&packs), 0, sizeof(packs));
        uint32_t smem = 0;
        h.n = (uint32_t)k;
        for (int q = 0; q < k; ++q) {
            const QEntry &e = *es[b + q];
            h.gx[q] = std::max(1u, e.gx); h.gy[q] = std::max(1u, e.gy);
            h.start[q + 1] = h.start[q] + e.gx * e.gy;
            memcpy(static_cast<void *>(&packs.p[q]), e.pack, sizeof(P));
            smem = std::max(smem, e.smem);
        }
        for (int q = k; q < BATCH_MAX; ++q) h.start[q + 1] = h.start[k];
        if (h.start[k] == 0) continue;
        hipLaunchKernelGGL((k_batch<Body, TPB, N, P, MINW>), dim3(h.start[k]), dim3(TPB), smem, st, h, packs);
    }
}

// launch<k_foo, TPB>(ctx, grid, dynamic LDS bytes, arguments of k_foo behind its VB)
template <auto Body, int TPB, int MINW = 1, class... Args>
void launch(plade_ctx *ctx, dim3 grid, size_t smem, Args... args) {
    using Tr = BodyTraits<decltype(Body)>;
    using P = typename Tr::pack;
    static_assert(std::is_trivially_copyable<P>::value, "kernel arguments must be plain data");
    const P p = Tr::make(args...);
    if (grid.x == 0 || grid.y == 0) return;
    Combiner *cb = ctx_combiner(ctx);
    if (!cb || sizeof(P) > PACK_MAX) {
        if (cb) {   // arguments too large to travel eight at a time: its own launch, at its place in the pair's order
            QEntry &e = cb->push(ctx);
            e.kind = QEntry::FUNC;
            e.fn = [p, grid, smem](hipStream_t st) {
                BatchHdr h; memset(&h, 0, sizeof(h));
                h.n = 1; h.gx[0] = grid.x; h.gy[0] = grid.y;
                for (int q = 0; q < BATCH_MAX; ++q) h.start[q + 1] = grid.x * grid.y;
                BatchPacks<P, 1> packs; packs.p[0] = p;
                hipLaunchKernelGGL((k_batch<Body, TPB, 1, P, MINW>), dim3(grid.x * grid.y), dim3(TPB), smem, st, h, packs);
            };
            return;
        }
        BatchHdr h; memset(&h, 0, sizeof(h));
        h.n = 1; h.gx[0] = grid.x; h.gy[0] = grid.y;
        for (int q = 0; q < BATCH_MAX; ++q) h.start[q + 1] = grid.x * grid.y;
        BatchPacks<P, 1> packs; packs.p[0] = p;
        hipLaunchKernelGGL((k_batch<Body, TPB, 1, P, MINW>), dim3(grid.x * grid.y), dim3(TPB), smem, ctx_stream(ctx), h, packs);
        return;
    }
    QEntry &e = cb->push(ctx);
    e.kind = QEntry::KERNEL;
    e.launch_many = &launch_many_impl<Body, TPB, P, MINW>;
    e.id = reinterpret_cast<const void *>(e.launch_many);     // one per (kernel, workgroup size): entries with the same id merge
    e.gx = grid.x; e.gy = grid.y; e.smem = (uint32_t)smem;
    memcpy(e.pack, static_cast<const void *>(&p), sizeof(P));
}
#endif

}  // namespace plade
