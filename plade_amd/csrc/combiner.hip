// plade_amd/csrc/combiner.hip -- the pairs of a group in lock step behind the plane extraction: see launch.h.
#include "ctx.h"
#include <algorithm>

namespace plade {

namespace {
constexpr int RANGES_MAX = 32;
struct RangeArgs { const uint32_t *src[RANGES_MAX]; uint32_t *dst[RANGES_MAX]; uint32_t words[RANGES_MAX], value[RANGES_MAX], n; };

// grid (chunks, ranges): words of range r from src to dst (FILL = true: `value` instead); with `flag`, the workgroup that finishes
// last raises it for the host (the stores of every workgroup are visible to the host before it takes its ticket)
template <bool FILL>
__global__ __launch_bounds__(256) void k_ranges(const RangeArgs a, uint32_t *__restrict__ counter, uint32_t *flag, uint32_t seq) {
    const uint32_t r = blockIdx.y, n = a.words[r];
    uint32_t *__restrict__ dst = a.dst[r];
    if (FILL) {
        const uint32_t v = a.value[r];
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = v;
    } else {
        const uint32_t *__restrict__ src = a.src[r];
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = src[i];
    }
    if (!flag) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(counter, 1u);
        if (t == gridDim.x * gridDim.y - 1u) {
            atomicExch(counter, 0u);
            __threadfence_system();
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <bool FILL>
void ranges_launch(hipStream_t st, const RangeArgs &a, uint32_t *counter, uint32_t *flag, uint32_t seq) {
    uint32_t most = 1;
    for (uint32_t q = 0; q < a.n; ++q) most = std::max(most, a.words[q]);
    const uint32_t gx = std::min(64u, cdiv(most, 1024u));
    hipLaunchKernelGGL(k_ranges<FILL>, dim3(gx, a.n), dim3(256), 0, st, a, counter, flag, seq);
    HIP_TRY(hipGetLastError());
}
}  // namespace

int combiner_slot(plade_ctx *c) { return c->comb_slot; }
hipStream_t ctx_stream(plade_ctx *c) { return c->stream; }
Combiner *ctx_combiner(plade_ctx *c) { return c->comb; }

int Combiner::join(plade_ctx *c) {
    std::lock_guard<std::mutex> lk(m);
    PLADE_REQUIRE(members < BATCH_MAX, PLADE_EINVAL, "combiner: too many pairs");
    int slot = 0;
    while (member[slot]) ++slot;
    member[slot] = c;
    c->comb = this;
    c->comb_slot = slot;
    q[slot].clear();
    ++members;
    return slot;
}

QEntry &Combiner::push(plade_ctx *c) {
    // only the pair's own thread touches its queue between two flushes, and a flush runs while every member waits
    std::vector<QEntry> &v = q[c->comb_slot];
    v.emplace_back();
    ++asked[c->comb_slot];
    return v.back();
}

// PLADE_TRACE_LOCKSTEP=1 (printing only): one line per group wait -- the host time since the previous wait ended, what was
// queued, how long issuing it and waiting for the GPU took, and the first kernel of the batch (tools/lockstep_timeline.py)
static bool trace_lockstep() { static const bool v = getenv("PLADE_TRACE_LOCKSTEP") != nullptr; return v; }

void Combiner::flush_locked(std::unique_lock<std::mutex> &lk) {
    (void)lk;
    ++waits;
    const bool tr = trace_lockstep();
    const Clock::time_point t_in = Clock::now();
    size_t n_q = 0;
    uint64_t issued0 = launches_issued;
    if (tr) for (int s = 0; s < BATCH_MAX; ++s) n_q += q[s].size();
    Clock::time_point t_issued = t_in;
    struct TraceOut {
        Combiner *c; bool on; const Clock::time_point &t_in, &t_issued; size_t &n_q; uint64_t &issued0;
        ~TraceOut() {
            if (!on) return;
            const Clock::time_point t_out = Clock::now();
            auto us = [](Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            fprintf(stderr, "[lockstep] %p stage %s wait %llu members %d host_gap_us %.0f queued %zu commands %llu issue_us %.0f gpu_wait_us %.0f\n", (void *)c, c->trace_stage,
                    (unsigned long long)c->waits, c->members, c->t_last_out.time_since_epoch().count() ? us(c->t_last_out, t_in) : 0.0, n_q,
                    (unsigned long long)(c->launches_issued - issued0), us(t_in, t_issued), us(t_issued, t_out));
            c->t_last_out = t_out;
        }
    } trace_out{this, tr, t_in, t_issued, n_q, issued0};
    try {
        HIP_TRY(hipSetDevice(lead->device));
        hipStream_t st = lead->stream;
        size_t cur[BATCH_MAX] = {};
        for (;;) {
            // the operation most pairs are about to issue goes next (pairs whose sequences differ for a while fall back in step)
            int best = -1, best_n = 0;
            for (int s = 0; s < BATCH_MAX; ++s) {
                if (cur[s] >= q[s].size()) continue;
                const QEntry &e = q[s][cur[s]];
                int n = 0;
                for (int t = s; t < BATCH_MAX; ++t) {
                    if (cur[t] >= q[t].size()) continue;
                    const QEntry &f = q[t][cur[t]];
                    if (f.kind == e.kind && e.kind != QEntry::FUNC && f.id == e.id) ++n;
                }
                if (e.kind == QEntry::FUNC) n = 1;
                if (n > best_n) { best_n = n; best = s; }
            }
            if (best < 0) break;
            const QEntry &lead_e = q[best][cur[best]];
            QEntry *es[BATCH_MAX];
            int n = 0;
            if (lead_e.kind == QEntry::FUNC) { es[n++] = &q[best][cur[best]]; ++cur[best]; }
            else
                for (int t = 0; t < BATCH_MAX; ++t) {
                    if (cur[t] >= q[t].size()) continue;
                    QEntry &f = q[t][cur[t]];
                    if (f.kind == lead_e.kind && f.id == lead_e.id) { es[n++] = &f; ++cur[t]; }
                }
            switch (es[0]->kind) {
                case QEntry::KERNEL:
                    es[0]->launch_many(st, n, es);
                    HIP_TRY(hipGetLastError());
                    ++launches_issued;
                    break;
                case QEntry::FUNC:
                    es[0]->fn(st);
                    HIP_TRY(hipGetLastError());
                    ++launches_issued;
                    break;
                case QEntry::COPY_IN:
                case QEntry::FILL: {
                    // consecutive copies / fills of a pair travel together too: keep taking while the heads are of this kind
                    RangeArgs a;
                    memset(&a, 0, sizeof(a));
                    const QEntry::Kind kind = es[0]->kind;
                    auto put = [&](const QEntry &e) {
                        a.src[a.n] = static_cast<const uint32_t *>(e.src); a.dst[a.n] = static_cast<uint32_t *>(e.dst);
                        a.words[a.n] = e.words; a.value[a.n] = e.value; ++a.n;
                    };
                    for (int k = 0; k < n; ++k) put(*es[k]);
                    // the ranges of one launch run concurrently: a further entry joins only if what it writes is disjoint from
                    // everything already in the launch (a pair that clears a buffer and then fills part of it keeps its order)
                    auto disjoint = [&](const QEntry &e) {
                        const uintptr_t lo = reinterpret_cast<uintptr_t>(e.dst), hi = lo + 4ull * e.words;
                        for (uint32_t k = 0; k < a.n; ++k) {
                            const uintptr_t l2 = reinterpret_cast<uintptr_t>(a.dst[k]), h2 = l2 + 4ull * a.words[k];
                            if (lo < h2 && l2 < hi) return false;
                        }
                        return true;
                    };
                    for (bool more = true; more && a.n < RANGES_MAX;) {
                        more = false;
                        for (int t = 0; t < BATCH_MAX && a.n < RANGES_MAX; ++t)
                            if (cur[t] < q[t].size() && q[t][cur[t]].kind == kind && disjoint(q[t][cur[t]])) { put(q[t][cur[t]]); ++cur[t]; more = true; }
                    }
                    if (kind == QEntry::FILL) ranges_launch<true>(st, a, nullptr, nullptr, 0);
                    else ranges_launch<false>(st, a, nullptr, nullptr, 0);
                    ++launches_issued;
                    break;
                }
            }
        }
        for (int s = 0; s < BATCH_MAX; ++s) q[s].clear();
        t_issued = Clock::now();
        // the read-backs of all pairs: one hand-over kernel per 32 ranges, the last one raises the lead's flag
        struct R { const void *src; char *dst; size_t bytes; };
        std::vector<R> reads;
        for (int s = 0; s < BATCH_MAX; ++s) {
            plade_ctx *c = member[s];
            if (!c || c->pending_reads.empty()) continue;
            c->ensure_read_arena();
            for (const plade_ctx::PendingRead &r : c->pending_reads) reads.push_back(R{r.src, c->read_arena_dev + r.off, r.bytes});
        }
        const bool sleepy = lead->params.host_wait != 0;
        const bool crowd = true;      // 100 / 200 us polls (50 / 100 measured the same throughput: tools/exp_ab.sh)
        if (sleepy) relax_timer_slack();
        if (!reads.empty()) {
            lead->ensure_read_arena();
            volatile uint32_t *flag = reinterpret_cast<volatile uint32_t *>(lead->read_arena.p + plade_ctx::READ_ARENA_BYTES);
            const uint32_t seq = ++lead->read_seq ? lead->read_seq : ++lead->read_seq;
            for (size_t i = 0; i < reads.size(); i += RANGES_MAX) {
                RangeArgs a;
                memset(&a, 0, sizeof(a));
                a.n = (uint32_t)std::min<size_t>(RANGES_MAX, reads.size() - i);
                for (uint32_t k = 0; k < a.n; ++k) {
                    a.src[k] = static_cast<const uint32_t *>(reads[i + k].src); a.dst[k] = reinterpret_cast<uint32_t *>(reads[i + k].dst);
                    a.words[k] = (uint32_t)(reads[i + k].bytes / 4);
                }
                const bool last = i + RANGES_MAX >= reads.size();
                ranges_launch<false>(st, a, lead->read_counter.p, last ? reinterpret_cast<uint32_t *>(lead->read_arena_dev + plade_ctx::READ_ARENA_BYTES) : nullptr, seq);
                ++launches_issued;
            }
            for (uint32_t polls = 0; *flag != seq; ++polls) {
                if ((polls & (sleepy ? 63u : 0xfffffu)) == (sleepy ? 63u : 0xfffffu)) {
                    const hipError_t e = hipStreamQuery(st);
                    if (e != hipSuccess && e != hipErrorNotReady) throw Err{-2, std::string("hipStreamQuery: ") + hipGetErrorString(e)};
                    if (e == hipSuccess && *flag != seq) throw Err{-2, "group hand-over: the stream finished without the flag"};
                }
                if (sleepy) poll_sleep((int)polls, crowd);
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        } else {
            if (!sleepy) HIP_TRY(hipStreamSynchronize(st));
            else
                for (int polls = 0;; ++polls) {
                    const hipError_t e = hipStreamQuery(st);
                    if (e == hipSuccess) break;
                    if (e != hipErrorNotReady) throw Err{-2, std::string("hipStreamQuery: ") + hipGetErrorString(e)};
                    poll_sleep(polls, crowd);
                }
        }
    } catch (const Err &e) {
        for (int s = 0; s < BATCH_MAX; ++s) q[s].clear();
        error = e.msg;
        error_code = e.code ? e.code : PLADE_EDEVICE;
    }
}

void Combiner::wait(plade_ctx *c) {
    std::unique_lock<std::mutex> lk(m);
    trace_stage = c->stage_name;       // (the last pair to arrive names the wait)
    const double c0 = thread_cpu_seconds();
    ++arrived;
    const uint64_t my_epoch = epoch;
    if (arrived >= members) {
        flush_locked(lk);
        arrived = 0;
        ++epoch;
        cv.notify_all();
    } else {
        cv.wait(lk, [&]() { return epoch != my_epoch; });
    }
    const int code = error_code;
    const std::string msg = error;
    lk.unlock();
    c->stats.add("cpu_sync_polls", thread_cpu_seconds() - c0);
    if (code) {
        c->pending_reads.clear();
        c->read_arena_used = 0; c->write_arena_used = 0;
        throw Err{code, "group launch sequence: " + msg};
    }
    c->finish_reads();
    c->write_arena_used = 0;
    ++c->wait_epoch;
    if (tl_deferred_free && !tl_deferred_free->empty()) {   // everything that could name them has run
        for (void *p : *tl_deferred_free) dev_free(p);
        tl_deferred_free->clear();
    }
}

void Combiner::leave(plade_ctx *c) {
    std::unique_lock<std::mutex> lk(m);
    if (c->comb != this) return;
    const int slot = c->comb_slot;
    --members;
    // what the pair still has queued stays in its slot and is issued with the next flush; if everybody else is already
    // waiting -- or gone, with anything (of any pair) still queued -- that flush is this thread's to run
    bool anything = false;
    for (int s = 0; s < BATCH_MAX; ++s) anything = anything || !q[s].empty();
    (void)slot;
    if (members > 0 ? arrived >= members : anything) {
        const bool any_waiting = arrived > 0;
        flush_locked(lk);
        if (any_waiting) { arrived = 0; ++epoch; cv.notify_all(); }
    }
    c->pending_reads.clear();     // a pair that gives up leaves nothing to deliver
    c->read_arena_used = 0; c->write_arena_used = 0;
    c->comb = nullptr;
    c->comb_slot = -1;
}

// Allocations a pair replaced while launches that may name them were queued: what the pair still had queued when it left
// stays in its slot until the others' next flush, so they are freed when the call ends (every flush, the last one included,
// returns with the stream drained).
void Combiner::bury(std::vector<void *> &gy) {
    std::lock_guard<std::mutex> lk(m);
    graveyard.insert(graveyard.end(), gy.begin(), gy.end());
    gy.clear();
}
Combiner::~Combiner() {
    for (void *p : graveyard) dev_free(p);
}

}  // namespace plade
