// plade_amd/csrc/ctx.h -- the plade_ctx object behind the C ABI.
#pragma once
#include "common.h"
#include "plade_hip.h"
#include "launch.h"
#include <ctime>
#include <sys/prctl.h>

namespace plade {

// Device-resident oriented cloud in SoA form (coalesced 16 B/lane loads in every scan kernel).
struct CloudDev {
    uint32_t n = 0;
    DBuf<float> soa;  // 6 planes of n floats: x | y | z | nx | ny | nz  (each padded to 4)
    DBuf<float> aos;  // the N x 6 input layout (x y z nx ny nz), kept for the gather-style stages
    size_t pitch = 0; // floats per plane (n rounded up to 4)
    float bbmin[3] = {0, 0, 0}, bbmax[3] = {0, 0, 0};
    const float *x() const { return soa.p; }
    const float *y() const { return soa.p + pitch; }
    const float *z() const { return soa.p + 2 * pitch; }
    const float *nx() const { return soa.p + 3 * pitch; }
    const float *ny() const { return soa.p + 4 * pitch; }
    const float *nz() const { return soa.p + 5 * pitch; }
};

struct Stats {
    std::vector<std::string> names;
    std::vector<double> values;
    std::string joined;
    void clear() { names.clear(); values.clear(); }
    void add(const std::string &n, double v) {
        for (size_t i = 0; i < names.size(); ++i)
            if (names[i] == n) { values[i] += v; return; }
        names.push_back(n);
        values.push_back(v);
    }
    void merge(const Stats &o) { for (size_t i = 0; i < o.names.size(); ++i) add(o.names[i], o.values[i]); }
};

}  // namespace plade

namespace plade {
// One sleeping poll of the throughput modes (params.host_wait != 0).  50 us for the first polls, 100 us after: with
// eight registrations in flight the GPU is never idle while one host thread oversleeps, and against 15 / 40 us the
// process spends 12 % less CPU at the same throughput (profiles/r3_experiments.md).  `crowd` (the pairs of a group: 32
// registrations in flight in batch mode): 100 / 200 us -- the same throughput for another 0.25 ms less CPU per registration
// (10 / 20 / 50 / 100 us first polls: 762 / 763 / 761 / 762 registrations/s at 3.4 / 3.0 / 2.4 / 2.2 busy threads, r4).
inline void poll_sleep(int polls, bool crowd = false) {
    const long base_ns = crowd ? 100000L : 50000L;
    timespec ts{0, polls < 8 ? base_ns : 2 * base_ns};
    nanosleep(&ts, nullptr);
}
inline double thread_cpu_seconds() {
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
// nanosleep() of a normal thread is rounded up by 50 us of timer slack; ask for 1 us once per thread
inline void relax_timer_slack() {
    static thread_local bool done = false;
    if (!done) { (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0); done = true; }
}
}  // namespace plade

namespace plade {
// look-back words and tile ticket of the single-launch scans (prims.hip: never reset, see scan_ticket)
struct ScanWork { DBuf<uint64_t> state; DBuf<uint32_t> ticket; uint32_t base = 0, gen = 0; };
// scratch of voxel_whole_batch (voxel.h): keys / values of the concatenated clouds, run heads, coordinates in voxel order
struct VoxBatchWork { DBuf<uint32_t> keys, keys2, vals, vals2, heads, seg_group, offs; DBuf<float> sorted_xyz, obb_coef; };
}  // namespace plade

namespace plade {
// Device -> host hand-over of up to COPY_OUT_RANGES small arrays in ONE kernel (prims.hip): the ranges are written into a
// page-locked arena of the context, then -- once every workgroup's stores are visible to the host -- the word `flag` (in the
// same arena) receives `seq`.  The host polls that word in memory; no copy command per array, no question to the runtime.
constexpr int COPY_OUT_RANGES = 8;
struct CopyOutArgs { const uint32_t *src[COPY_OUT_RANGES]; uint32_t off_words[COPY_OUT_RANGES], words[COPY_OUT_RANGES], n; };
void copy_out_launch(hipStream_t st, const CopyOutArgs &a, uint32_t *arena_words, uint32_t *counter, uint32_t *flag, uint32_t seq);
}  // namespace plade

struct plade_cloud {
    plade::CloudDev dev;
    std::vector<float> host_copy;  // pos_nrm kept for the small host-side gathers
};

namespace plade { struct RegistrationWork; struct RansacWork; }
namespace plade { void comm_all_gather_dev(plade_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t stream); }

struct plade_ctx {
    int device = 0;
    plade::RegistrationWork *reg_work = nullptr;
    plade::RansacWork *ransac_work = nullptr;
    plade_ctx *peers[PLADE_GROUP_MAX - 1] = {};   // the contexts of pairs 1.. of a group (plade_registration_pairs): stream, aux, work areas
    hipEvent_t ev_group = nullptr;   // end of a group's joint plane extraction on `stream` (the peers' streams wait for it)
    bool in_group = false;      // this context carries one pair of a group of several (register_group)
    const char *stage_name = "";       // the innermost StageTimer alive on this context (trace lines)
    plade::Combiner *comb = nullptr;   // ... whose launches, small copies and waits are merged with those of the other pairs (launch.h)
    int comb_slot = -1;
    plade_ctx *aux = nullptr;   // second stream + work areas: stages of the source cloud that are independent of the
                                // target's run concurrently with them
    hipStream_t stream = nullptr;
    // device copies of the clouds the host-pointer entry points register (grow-only, reused from call to call: a
    // hipMalloc / hipFree pair per call costs more than the upload itself, and hipFree synchronises the device)
    plade::CloudDev up_tgt, up_src;
    // batch mode (plade_registration_next): the NEXT pair's clouds are uploaded on a stream of their own while the current
    // pair registers; the next call finds them here and swaps them in
    struct Prefetch {
        hipStream_t stream = nullptr;
        plade::CloudDev cl[2 * PLADE_GROUP_MAX];   // target, source of pair 0; target, source of pair 1; ...
        const float *ptr[2 * PLADE_GROUP_MAX] = {};
        uint32_t n[2 * PLADE_GROUP_MAX] = {};
        int count = 0;                         // clouds in flight (2 per pair)
        bool valid = false;
        plade::HBuf<int> h;      // page-locked: [0..8) init pattern, then 8 ints per bounding box read back
        plade::DBuf<int> d;      // 8 ints per box being reduced
    } pf;
    plade_params params;
    struct CandidateShard { uint32_t rank = 0, world = 1, min_candidates = 0; plade_exchange_fn exchange = nullptr; void *user = nullptr;
                            plade_comm *comm = nullptr; /* RCCL form (comm.hip): the counts never leave the device before the all-gather */ } shard;
    // number of completed host waits on `stream` (sync(), the extraction's flag waits): a stage that left results in host-mapped
    // memory behind kernels queued earlier remembers the count at enqueue time and knows from it whether anything has waited since
    uint64_t wait_epoch = 0;
    std::string last_error;
    std::map<std::string, std::vector<char>> dump;
    plade::Stats stats;
    // optional per-kernel timing with HIP events on this ctx's stream (params.dump & 2)
    struct EvRec { hipEvent_t a = nullptr, b = nullptr; std::string tag; double bytes = 0; int clk = -1; double clk_secs = -1.0; };
    std::vector<EvRec> evs;
    // Under load the two events of a record also see whatever other streams run on the same hardware queue in between;
    // the kernels that matter for the roofline therefore also take the device's wall clock themselves (first wavefront
    // in, last wavefront out: what rocprofv3's kernel trace reports): one (min start, max end) pair per launch.
    static constexpr int CLK_SLOTS = 1024, CLK_WAYS = 64;   // launches per collect window, (start, end) pairs per launch
    plade::DBuf<unsigned long long> clk_dev;
    int clk_used = 0;
    double clk_hz = 0;
    // params.dump & 4 (with & 2): the profiled launches stay inside the captured graph of the plane extraction's iteration,
    // as in the timed path.  Kernel arguments of a graph are fixed, so launch j of an iteration always stamps clock pair j
    // (the graph is captured with these pointers; the plain graph has nullptr there), the pairs are read and reset behind
    // every iteration (ev_graph_clocks), and there are no HIP events -- they cannot be placed between the nodes of a
    // graph launch: a record's `seconds` is then the kernel's own clock too.
    bool graph_clocks() const { return (params.dump & 6) == 6; }
    bool capturing = false;                 // between hipStreamBeginCapture and EndCapture of such a graph
    int cap_slot = 0;
    std::vector<std::string> cap_tags;      // tags of the stamped launches of the graph being captured, in launch order
    std::vector<unsigned long long> clk_host, clk_init;
    void ensure_clk() {
        if (clk_dev.p) return;
        {
            clk_dev.ensure(2 * (size_t)CLK_SLOTS * CLK_WAYS);
            std::vector<unsigned long long> init(2 * (size_t)CLK_SLOTS * CLK_WAYS);
            for (size_t i = 0; i < init.size() / 2; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0ull; }
            (void)hipMemcpy(clk_dev.p, init.data(), init.size() * 8, hipMemcpyHostToDevice);
            int khz = 0;
            (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device);
            clk_hz = khz > 0 ? khz * 1e3 : 1e8;
        }
    }
    unsigned long long *ev_clock() {   // the clock pair of the record opened last (nullptr outside profiled runs)
        if (graph_clocks()) {
            if (!capturing || cap_slot >= CLK_SLOTS) return nullptr;
            ensure_clk();
            return clk_dev.p + 2 * (size_t)CLK_WAYS * (cap_slot++);
        }
        if (!profiling() || evs.empty() || clk_used >= CLK_SLOTS) return nullptr;
        ensure_clk();
        evs.back().clk = clk_used;
        return clk_dev.p + 2 * (size_t)CLK_WAYS * (clk_used++);
    }
    // graph path: one record per stamped launch of the graph just queued ...
    void ev_graph_launched(const std::vector<std::string> &tags) {
        for (size_t j = 0; j < tags.size(); ++j) { EvRec r; r.tag = tags[j]; r.clk = (int)j; evs.push_back(r); }
    }
    // ... and, with the iteration finished (the caller has waited for the stream), their clock pairs: read, reset
    void ev_graph_clocks(size_t ev0) {
        size_t used = 0;
        for (size_t e = ev0; e < evs.size(); ++e) if (!evs[e].a && evs[e].clk >= 0) used = std::max(used, (size_t)evs[e].clk + 1);
        if (!used) return;
        clk_host.resize(2 * used * CLK_WAYS);
        if (clk_init.size() < clk_host.size()) {
            clk_init.resize(clk_host.size());
            for (size_t i = 0; i < clk_init.size() / 2; ++i) { clk_init[2 * i] = ~0ull; clk_init[2 * i + 1] = 0ull; }
        }
        HIP_TRY(hipMemcpyAsync(clk_host.data(), clk_dev.p, clk_host.size() * 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(clk_dev.p, clk_init.data(), clk_host.size() * 8, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (size_t e = ev0; e < evs.size(); ++e) {
            EvRec &r = evs[e];
            if (r.a || r.clk < 0) continue;
            unsigned long long lo = ~0ull, hi = 0ull;
            for (int w = 0; w < CLK_WAYS; ++w) {
                lo = std::min(lo, clk_host[2 * ((size_t)r.clk * CLK_WAYS + w)]);
                hi = std::max(hi, clk_host[2 * ((size_t)r.clk * CLK_WAYS + w) + 1]);
            }
            r.clk_secs = hi > lo ? (double)(hi - lo) / clk_hz : -1.0;
            r.clk = -1;
        }
    }
    bool profiling() const { return (params.dump & 2) != 0; }
    void ev_begin(const char *tag, double bytes) {
        if (!profiling()) return;
        if (graph_clocks()) { if (capturing) cap_tags.push_back(tag); return; }
        EvRec r;
        r.tag = tag; r.bytes = bytes;
        (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
        (void)hipEventRecord(r.a, stream);
        evs.push_back(r);
    }
    void ev_end() {
        if (!profiling() || graph_clocks() || evs.empty()) return;
        (void)hipEventRecord(evs.back().b, stream);
    }
    void ev_collect() {
        if (evs.empty()) return;
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> clk(2 * (size_t)clk_used);   // per launch: min start, max end
        if (clk_used) {
            std::vector<unsigned long long> raw(2 * (size_t)clk_used * CLK_WAYS);
            (void)hipMemcpy(raw.data(), clk_dev.p, raw.size() * 8, hipMemcpyDeviceToHost);
            for (int i = 0; i < clk_used; ++i) {
                unsigned long long lo = ~0ull, hi = 0ull;
                for (int w = 0; w < CLK_WAYS; ++w) {
                    lo = std::min(lo, raw[2 * ((size_t)i * CLK_WAYS + w)]);
                    hi = std::max(hi, raw[2 * ((size_t)i * CLK_WAYS + w) + 1]);
                }
                clk[2 * i] = lo; clk[2 * i + 1] = hi;
            }
            for (size_t i = 0; i < raw.size() / 2; ++i) { raw[2 * i] = ~0ull; raw[2 * i + 1] = 0ull; }
            (void)hipMemcpy(clk_dev.p, raw.data(), raw.size() * 8, hipMemcpyHostToDevice);
            clk_used = 0;
        }
        for (auto &r : evs) {
            float ms = 0.f;
            if (!r.a) {   // a launch inside a graph: its own clock is all there is
                if (r.bytes >= 0 && r.clk_secs >= 0) {
                    stats.add("k_" + r.tag + "_seconds", r.clk_secs);
                    stats.add("k_" + r.tag + "_launches", 1.0);
                    stats.add("k_" + r.tag + "_bytes", r.bytes);
                    stats.add("k_" + r.tag + "_clock_seconds", r.clk_secs);
                    stats.add("k_" + r.tag + "_clock_launches", 1.0);
                    stats.add("k_" + r.tag + "_clock_bytes", r.bytes);
                } else stats.add("k_" + r.tag + "_idle_launches", 1.0);
                continue;
            }
            (void)hipEventElapsedTime(&ms, r.a, r.b);
            if (r.bytes >= 0) {   // a negative byte count marks a launch that had nothing to do (device-side early exit)
                stats.add("k_" + r.tag + "_seconds", ms * 1e-3);
                stats.add("k_" + r.tag + "_launches", 1.0);
                stats.add("k_" + r.tag + "_bytes", r.bytes);
                if (r.clk >= 0 && clk[2 * r.clk + 1] > clk[2 * r.clk]) {
                    stats.add("k_" + r.tag + "_clock_seconds", (double)(clk[2 * r.clk + 1] - clk[2 * r.clk]) / clk_hz);
                    stats.add("k_" + r.tag + "_clock_launches", 1.0);
                    stats.add("k_" + r.tag + "_clock_bytes", r.bytes);
                }
            } else stats.add("k_" + r.tag + "_idle_launches", 1.0);   // queued, found nothing to do, returned at once
            (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
        }
        evs.clear();
    }
    // Wait until everything queued on `s` (default: this ctx's stream) has finished.  Every wait of the HIP runtime
    // spins (one CPU per waiting thread, whatever the event flags: tools/wait_cost.hip); with several contexts in
    // flight per GPU that exhausts a container's CPU quota long before the GPU is full, so params.host_wait = 1
    // polls the stream and sleeps in between.
    void sync(hipStream_t s = nullptr) {
        if (!s) s = stream;
        if (comb && s == stream) { comb->wait(this); return; }
        if (s == stream && !pending_reads.empty()) { sync_with_reads(); return; }
        if (params.host_wait == 0) { HIP_TRY(hipStreamSynchronize(s)); }
        else {
            plade::relax_timer_slack();
            const double c0 = plade::thread_cpu_seconds();
            struct Acc { plade::Stats &st; double c0; ~Acc() { st.add("cpu_sync_polls", plade::thread_cpu_seconds() - c0); } } acc{stats, c0};
            for (int polls = 0;; ++polls) {
                const hipError_t e = hipStreamQuery(s);
                if (e == hipSuccess) break;
                if (e != hipErrorNotReady) throw plade::Err{-2, std::string("hipStreamQuery: ") + hipGetErrorString(e)};
                plade::poll_sleep(polls, in_group);
            }
        }
        if (s == stream) { finish_reads(); write_arena_used = 0; ++wait_epoch; }
    }
    // Device -> host readback on this ctx's stream; `dst` is valid after the next sync(), and `src` must not be overwritten
    // before it: the small readbacks (there are ~20 per registration) are only NOTED here; sync() hands all of them over
    // with one kernel that writes them into a page-locked arena and then raises a flag word in the same arena, which the
    // host polls in memory (plade::copy_out_launch).  Against one copy command per array and a stream query per poll
    // that is fewer commands in the stream and no work for the runtime's event thread (profiles/r3_experiments.md).
    // Large arrays, and whatever does not fit the arena, are copied directly.
    struct PendingRead { void *dst; const void *src; size_t off, bytes; std::vector<char> eager; };
    // The contract of d2h() -- `src` must keep its contents until the next sync() -- is checked, not just stated, under
    // PLADE_DEBUG_READS=1: the range is also copied at once (a stream wait + a blocking copy: debug only) and compared with what
    // the deferred hand-over delivers; a difference fails the call (the GPU test-suite passes in this mode).
    bool debug_reads = getenv("PLADE_DEBUG_READS") != nullptr;
    std::vector<PendingRead> pending_reads;
    plade::HBuf<char> read_arena;
    char *read_arena_dev = nullptr;       // the arena as the device addresses it
    plade::DBuf<uint32_t> read_counter;
    uint32_t read_seq = 0;
    size_t read_arena_used = 0;
    static constexpr size_t READ_ARENA_BYTES = 8u << 20, READ_DIRECT_BYTES = 2u << 20;
    void d2h(void *dst, const void *src, size_t bytes) {
        if (!bytes) return;
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (bytes > READ_DIRECT_BYTES || (bytes & 3) || (reinterpret_cast<uintptr_t>(src) & 3) || read_arena_used + need > READ_ARENA_BYTES) {
            if (comb) {    // at its place in the pair's order, on the group's stream
                plade::QEntry &e = comb->push(this);
                e.kind = plade::QEntry::FUNC;
                e.fn = [dst, src, bytes](hipStream_t st) { HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st)); };
                return;
            }
            if (params.host_wait != 0) sync();   // drain with sleeping polls first: only the copy itself is waited for actively
            HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
            return;
        }
        pending_reads.push_back(PendingRead{dst, src, read_arena_used, bytes, {}});
        if (debug_reads) {
            HIP_TRY(hipStreamSynchronize(stream));
            pending_reads.back().eager.resize(bytes);
            HIP_TRY(hipMemcpy(pending_reads.back().eager.data(), src, bytes, hipMemcpyDeviceToHost));
        }
        read_arena_used += need;
    }
    char *ensure_read_arena() {
        // host-mapped and COHERENT: the kernel's stores must reach the host while the stream keeps running
        char *arena = read_arena.ensure(READ_ARENA_BYTES + 256, hipHostMallocMapped | hipHostMallocCoherent);
        if (!read_arena_dev) {
            HIP_TRY(hipHostGetDevicePointer((void **)&read_arena_dev, arena, 0));
            *reinterpret_cast<volatile uint32_t *>(arena + READ_ARENA_BYTES) = 0u;   // sequence numbers start at 1
        }
        if (!read_counter.p) { read_counter.ensure(4); HIP_TRY(hipMemsetAsync(read_counter.p, 0, 16, stream)); }
        return arena;
    }
    void sync_with_reads() {
        char *arena = ensure_read_arena();
        volatile uint32_t *flag = reinterpret_cast<volatile uint32_t *>(arena + READ_ARENA_BYTES);
        const uint32_t seq = ++read_seq ? read_seq : ++read_seq;   // never 0
        for (size_t i = 0; i < pending_reads.size(); i += plade::COPY_OUT_RANGES) {
            plade::CopyOutArgs a;
            a.n = (uint32_t)std::min<size_t>(plade::COPY_OUT_RANGES, pending_reads.size() - i);
            for (uint32_t q = 0; q < a.n; ++q) {
                const PendingRead &r = pending_reads[i + q];
                a.src[q] = static_cast<const uint32_t *>(r.src); a.off_words[q] = (uint32_t)(r.off / 4); a.words[q] = (uint32_t)(r.bytes / 4);
            }
            const bool last = i + plade::COPY_OUT_RANGES >= pending_reads.size();
            plade::copy_out_launch(stream, a, reinterpret_cast<uint32_t *>(read_arena_dev), read_counter.p,
                                   last ? reinterpret_cast<uint32_t *>(read_arena_dev + READ_ARENA_BYTES) : nullptr, seq);
        }
        const double c0 = plade::thread_cpu_seconds();
        struct Acc { plade::Stats &st; double c0; ~Acc() { st.add("cpu_sync_polls", plade::thread_cpu_seconds() - c0); } } acc{stats, c0};
        if (params.host_wait != 0) plade::relax_timer_slack();
        for (uint32_t polls = 0; *flag != seq; ++polls) {
            // the stream query only catches a stream that died without raising the flag
            if ((polls & (params.host_wait != 0 ? 63u : 0xfffffu)) == (params.host_wait != 0 ? 63u : 0xfffffu)) {
                const hipError_t e = hipStreamQuery(stream);
                if (e != hipSuccess && e != hipErrorNotReady) throw plade::Err{-2, std::string("hipStreamQuery: ") + hipGetErrorString(e)};
                if (e == hipSuccess && *flag != seq) throw plade::Err{-2, "device -> host hand-over: the stream finished without the flag"};
            }
            if (params.host_wait != 0) plade::poll_sleep((int)polls, in_group);
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        finish_reads();
        write_arena_used = 0;
        ++wait_epoch;
    }
    // Host -> device upload of a small table: staged in the pinned arena so that the copy is asynchronous (a pageable
    // source makes the runtime copy it to its own staging buffer and, for some sizes, wait for the transfer).
    plade::HBuf<char> write_arena;
    size_t write_arena_used = 0;
    // true: `src` has been copied into the arena and may be released at once; false: the copy reads `src` itself, which
    // must stay valid until the next sync() of this stream
    char *write_arena_dev = nullptr;      // the staging arena as the device addresses it (the group's copy kernel reads it there)
    bool h2d(void *dst, const void *src, size_t bytes) {
        if (!bytes) return true;
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (bytes > READ_DIRECT_BYTES || write_arena_used + need > READ_ARENA_BYTES) {
            if (comb) {
                plade::QEntry &e = comb->push(this);
                e.kind = plade::QEntry::FUNC;
                e.fn = [dst, src, bytes](hipStream_t st) { HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st)); };
                return false;
            }
            HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
            return false;
        }
        char *a = write_arena.ensure(READ_ARENA_BYTES, hipHostMallocMapped) + write_arena_used;
        memcpy(a, src, bytes);
        if (comb && !(bytes & 3) && !(reinterpret_cast<uintptr_t>(dst) & 3)) {
            // merged with the other pairs' uploads into one copy kernel that reads the staging arenas over PCIe
            if (!write_arena_dev) HIP_TRY(hipHostGetDevicePointer((void **)&write_arena_dev, write_arena.p, 0));
            plade::QEntry &e = comb->push(this);
            e.kind = plade::QEntry::COPY_IN;
            e.dst = dst; e.src = write_arena_dev + write_arena_used; e.words = (uint32_t)(bytes / 4);
        } else if (comb) {
            plade::QEntry &e = comb->push(this);
            e.kind = plade::QEntry::FUNC;
            e.fn = [dst, a, bytes](hipStream_t st) { HIP_TRY(hipMemcpyAsync(dst, a, bytes, hipMemcpyHostToDevice, st)); };
        } else HIP_TRY(hipMemcpyAsync(dst, a, bytes, hipMemcpyHostToDevice, stream));
        write_arena_used += need;
        return true;
    }
    // fill / device-to-device copy on this context's stream (in a group: at their place in the pair's order on the group's)
    void fill_async(void *dst, int byte_value, size_t bytes) {
        if (!bytes) return;
        if (comb && bytes <= (1u << 20) && !(bytes & 3) && !(reinterpret_cast<uintptr_t>(dst) & 3)) {
            plade::QEntry &e = comb->push(this);
            e.kind = plade::QEntry::FILL;
            const uint32_t b = (uint32_t)(byte_value & 0xff);
            e.dst = dst; e.words = (uint32_t)(bytes / 4); e.value = b | (b << 8) | (b << 16) | (b << 24);
        } else if (comb) {
            plade::QEntry &e = comb->push(this);
            e.kind = plade::QEntry::FUNC;
            e.fn = [dst, byte_value, bytes](hipStream_t st) { HIP_TRY(hipMemsetAsync(dst, byte_value, bytes, st)); };
        } else HIP_TRY(hipMemsetAsync(dst, byte_value, bytes, stream));
    }
    void copy_dd_async(void *dst, const void *src, size_t bytes) {
        if (!bytes) return;
        if (comb) {
            plade::QEntry &e = comb->push(this);
            e.kind = plade::QEntry::FUNC;
            e.fn = [dst, src, bytes](hipStream_t st) { HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st)); };
        } else HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
    }
    // a kernel that has not been given its launch.h form: at its place in the pair's order, on its own
    template <class F> void raw_launch(F f) {
        if (comb) { plade::QEntry &e = comb->push(this); e.kind = plade::QEntry::FUNC; e.fn = f; }
        else f(stream);
    }
    // forget the queued hand-overs (their destinations may be gone): after an error, and before every call
    void drop_reads() {
        if (!pending_reads.empty()) (void)hipStreamSynchronize(stream);   // the copies themselves still target the arena
        else if (write_arena_used) (void)hipStreamSynchronize(stream);
        pending_reads.clear();
        read_arena_used = 0;
        write_arena_used = 0;
        if (aux) aux->drop_reads();
        for (plade_ctx *p : peers) if (p) p->drop_reads();
    }
    void finish_reads() {
        bool changed = false;
        for (const PendingRead &r : pending_reads) {
            memcpy(r.dst, read_arena.p + r.off, r.bytes);
            if (!r.eager.empty() && memcmp(r.eager.data(), read_arena.p + r.off, r.bytes) != 0) changed = true;
        }
        pending_reads.clear();
        read_arena_used = 0;
        if (changed) throw plade::Err{-2, "PLADE_DEBUG_READS: a range noted by d2h() changed before the wait that delivers it"};
    }
    plade::ScanWork scan;
    plade::VoxBatchWork vox_batch;   // the whole-cloud voxel grids of a group's clouds (voxel_whole_batch)
    // the radix sort's two global digit histograms (radix_sort.hip) and the number of sorts issued on this context
    plade::DBuf<uint32_t> sort_ghist;
    uint32_t sort_seq = 0;
    uint32_t sort_segs[2] = {0, 0};   // segments whose histogram words the last sort on each buffer left non-zero
    // generic scratch
    plade::DBuf<char> scratch[8];
    plade::HBuf<char> pinned[4];

    template <class T>
    void put(const std::string &name, const T *data, size_t count) {
        if (!params.dump) return;
        std::vector<char> &b = dump[name];
        b.resize(count * sizeof(T));
        if (count) memcpy(b.data(), data, count * sizeof(T));
    }
    template <class T>
    void put1(const std::string &name, T v) { put(name, &v, 1); }
    // copy a device array into the dump (sync)
    template <class T>
    void put_dev(const std::string &name, const T *dptr, size_t count) {
        if (!params.dump) return;
        std::vector<T> h(count);
        if (count) d2h(h.data(), dptr, count * sizeof(T));
        sync();
        put(name, h.data(), count);
    }
};

#ifdef __HIPCC__
#include <tuple>
namespace plade {
// An ordinary __global__ kernel on the context's stream; in a group (launch.h) at its place in the pair's order on the group's
// stream, on its own.  The arguments are evaluated NOW, as a launch would.
template <class... KArgs, class... Args>
void launch_raw(plade_ctx *ctx, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t smem, Args... args) {
    if (!ctx->comb) { hipLaunchKernelGGL(kernel, grid, block, smem, ctx->stream, static_cast<KArgs>(args)...); return; }
    std::tuple<KArgs...> t(static_cast<KArgs>(args)...);
    ctx->raw_launch([kernel, grid, block, smem, t](hipStream_t st) {
        std::apply([&](auto... a) { hipLaunchKernelGGL(kernel, grid, block, smem, st, a...); }, t);
    });
}
}  // namespace plade
#endif

namespace plade {

// run `body`, translate exceptions into C error codes
template <class F>
int guarded(plade_ctx *ctx, F body) {
    if (!ctx) return PLADE_EINVAL;
    try {
        // HIP's current device is per host thread (default 0): a context may be driven from any thread, so every
        // entry point binds the calling thread to the context's GPU first (allocations follow the current device)
        HIP_TRY(hipSetDevice(ctx->device));
        ctx->drop_reads();   // nothing may be pending from a call that ended in an error
        return body();
    } catch (const Err &e) {
        ctx->drop_reads();
        ctx->last_error = e.msg;
        return e.code;
    } catch (const std::exception &e) {
        ctx->drop_reads();
        ctx->last_error = e.what();
        return PLADE_EDEVICE;
    }
}

struct StageTimer {
    plade_ctx *ctx;
    const char *name;
    Clock::time_point t0;
    double cpu0;      // CPU time of the calling thread (helper threads account for themselves)
    const char *outer;
    StageTimer(plade_ctx *c, const char *n) : ctx(c), name(n), t0(Clock::now()), cpu0(plade::thread_cpu_seconds()), outer(c->stage_name) { c->stage_name = n; }
    ~StageTimer() {
        ctx->stage_name = outer;
        ctx->stats.add(name, secs_since(t0));
        ctx->stats.add(std::string("cpu") + (name + 1), plade::thread_cpu_seconds() - cpu0);   // "t_x" -> "cpu_x"
    }
};

// ---- kernels / stages implemented across the .hip files ----------------------------------
// cloud upload: AoS N x 6 (host) -> SoA on device
void cloud_upload(plade_ctx *ctx, const float *pos_nrm, uint32_t n, CloudDev &out);
void cloud_upload_pair(plade_ctx *ctx, const float *tgt, uint32_t n_t, CloudDev &out_t, const float *src, uint32_t n_s, CloudDev &out_s);
// `count` (<= 2 * PLADE_GROUP_MAX) clouds in one go: all copies first, one wait, then conversion + bounding boxes of all of them, one wait
void cloud_upload_many(plade_ctx *ctx, int count, const float *const ptr[], const uint32_t n[], CloudDev *const out[]);
// batch mode: queue the upload of the NEXT call's clouds (count = 2 per pair: target, source[, target, source ...]) on the context's
// prefetch stream / take a finished prefetch over into out[] (false: nothing usable was prefetched)
void cloud_prefetch(plade_ctx *ctx, int count, const float *const ptr[], const uint32_t n[]);
bool cloud_take_prefetched(plade_ctx *ctx, int count, const float *const ptr[], const uint32_t n[], CloudDev *const out[]);
// a pending prefetch is waited for and dropped (every entry point that does not consume it: the caller may release or reuse
// the buffers after the call that announced them, and a reused address must not be mistaken for the announced pair)
void cloud_drop_prefetch(plade_ctx *ctx);

}  // namespace plade
