// plade_amd/csrc/api.hip -- context management and instrumentation entry points of the C ABI.
#include "ctx.h"
#include "exact_sort.h"
#include "pipeline.h"
#include "ransac.h"

using namespace plade;

extern "C" void plade_default_params(plade_params *p) {
    if (!p) return;
    p->max_planes = 40;         // code/PLADE/plade.cpp:604
    p->min_planes = 10;         // code/PLADE/plade.cpp:603
    p->max_candidates = 200;    // code/PLADE/plade.cpp:54
    p->init_min_support = 10000;  // code/PLADE/plade.cpp:602
    p->orient_normals = 0;      // the reference's behaviour (plane_extraction.cpp:43-58 never flips)
    p->dump = 0;
    p->ransac_seed = 0x9E3779B97F4A7C15ull;
    p->host_wait = 0;
    p->unoriented_normals = 0;
    p->ransac_topup = 1;
    p->match_window = 0;
    p->match_cell_budget = 0;
    p->group_max_points = 48000000u;
    p->prepare_sides = 0;
    p->closest_point_mode = 1;  // the reference's fp32 SVD solves (k_svd.h); 0 = the fp64 closed form (opt-in deviation)
    // pure: no environment look-ups here -- programs that cannot pass plade_params (the CLI, the C++ registration()
    // overloads) read their opt-in switches themselves (plade_host.cpp: context())
}

// ---- device memory (common.h) --------------------------------------------------------------------------------------------------
namespace plade {
namespace {
struct Slab { char *base = nullptr; size_t used = 0; uint32_t live = 0; int device = 0; };
struct SmallPool {
    static constexpr size_t SLAB = 32u << 20, SMALL = 1u << 20, GRAN = 256;
    std::mutex m;
    std::vector<Slab> slabs;            // (a few dozen per process: linear searches)
    std::vector<size_t> empty;          // slabs whose blocks have all come back (not the device's current one)
    std::map<int, size_t> current;      // device -> the slab being filled
    bool enabled = getenv("PLADE_EXP_NO_POOL") == nullptr;   // A/B timing hook
    void *alloc(size_t bytes) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        const size_t need = (bytes + GRAN - 1) & ~(GRAN - 1);
        std::lock_guard<std::mutex> lk(m);
        auto it = current.find(dev);
        if (it == current.end() || slabs[it->second].used + need > SLAB) {
            size_t pick = (size_t)-1;
            if (it != current.end() && slabs[it->second].live == 0) pick = it->second;      // nothing of it is in use: start it over
            for (size_t q = 0; q < empty.size() && pick == (size_t)-1; ++q)
                if (slabs[empty[q]].device == dev) { pick = empty[q]; empty.erase(empty.begin() + (long)q); }
            if (pick == (size_t)-1) {
                Slab s;
                s.device = dev;
                HIP_TRY(hipMalloc((void **)&s.base, SLAB));
                slabs.push_back(s);
                pick = slabs.size() - 1;
            } else {
                // its blocks have all been returned, but kernels queued earlier may still be using them (hipFree would have waited too)
                HIP_TRY(hipDeviceSynchronize());
                slabs[pick].used = 0;
            }
            if (it != current.end() && it->second != pick && slabs[it->second].live == 0) empty.push_back(it->second);
            current[dev] = pick;
            it = current.find(dev);
        }
        Slab &s = slabs[it->second];
        void *p = s.base + s.used;
        s.used += need;
        ++s.live;
        return p;
    }
    bool release(void *p) {     // false: not a pool block
        std::lock_guard<std::mutex> lk(m);
        for (size_t q = 0; q < slabs.size(); ++q) {
            Slab &s = slabs[q];
            if ((char *)p >= s.base && (char *)p < s.base + SLAB) {
                if (--s.live == 0) {
                    auto it = current.find(s.device);
                    if (it == current.end() || it->second != q) empty.push_back(q);
                }
                return true;
            }
        }
        return false;
    }
};
SmallPool &small_pool() { static SmallPool *p = new SmallPool; return *p; }   // (never destroyed: frees may arrive during process exit)
}  // namespace

void *dev_alloc(size_t bytes) {
    const auto t0 = std::chrono::steady_clock::now();
    void *p = nullptr;
    SmallPool &sp = small_pool();
    if (sp.enabled && bytes <= SmallPool::SMALL) p = sp.alloc(bytes);
    else HIP_TRY(hipMalloc(&p, bytes));
    AllocStats &as = alloc_stats();
    as.calls += 1; as.bytes += bytes;
    as.nanos += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return p;
}
void dev_free(void *p) {
    if (!p) return;
    if (small_pool().release(p)) return;
    (void)hipFree(p);
}
}  // namespace plade

extern "C" const char *plade_version(void) { return "plade-hip 0.1 (gfx950)"; }

extern "C" int plade_ctx_create(int device, plade_ctx **out) {
    if (!out) return PLADE_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return PLADE_EDEVICE;
    if (hipSetDevice(device) != hipSuccess) return PLADE_EDEVICE;
    plade_ctx *c = new plade_ctx;
    c->device = device;
    plade_default_params(&c->params);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return PLADE_EDEVICE; }
    *out = c;
    return PLADE_OK;
}

extern "C" void plade_ctx_destroy(plade_ctx *ctx) {
    if (!ctx) return;
    if (ctx->aux) { plade_ctx_destroy(ctx->aux); ctx->aux = nullptr; }
    for (plade_ctx *&p : ctx->peers) if (p) { plade_ctx_destroy(p); p = nullptr; }
    if (ctx->ev_group) { (void)hipEventDestroy(ctx->ev_group); ctx->ev_group = nullptr; }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->reg_work) plade::registration_work_destroy(ctx->reg_work);
    if (ctx->ransac_work) plade::ransac_work_destroy(ctx->ransac_work);
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->pf.stream) { (void)hipStreamSynchronize(ctx->pf.stream); (void)hipStreamDestroy(ctx->pf.stream); }
    delete ctx;
    if (getenv("PLADE_DEBUG_ALLOC")) {
        plade::AllocStats &as = plade::alloc_stats();
        fprintf(stderr, "[plade] device allocations so far: %llu calls, %.1f MB, %.1f ms in hipMalloc\n", (unsigned long long)as.calls.load(),
                as.bytes.load() / 1e6, as.nanos.load() / 1e6);
    }
}

// waits for everything queued on `device` by this process (the timed region of a benchmark is bracketed with it)
extern "C" int plade_device_synchronize(int device) {
    if (hipSetDevice(device) != hipSuccess) return PLADE_EDEVICE;
    return hipDeviceSynchronize() == hipSuccess ? PLADE_OK : PLADE_EDEVICE;
}

extern "C" int plade_device_count(void) {
    int count = 0;
    return hipGetDeviceCount(&count) == hipSuccess && count > 0 ? count : 0;
}

extern "C" const char *plade_last_error(const plade_ctx *ctx) { return ctx ? ctx->last_error.c_str() : "null ctx"; }

extern "C" int plade_set_params(plade_ctx *ctx, const plade_params *p) {
    if (!ctx || !p) return PLADE_EINVAL;
    if (p->max_planes < 1 || p->min_planes < 0 || p->max_candidates < 1 || p->init_min_support < 1 || p->match_window < -1 || p->match_window > 1 ||
        p->closest_point_mode < 0 || p->closest_point_mode > 1) return PLADE_EINVAL;
    ctx->params = *p;
    return PLADE_OK;
}

extern "C" int plade_set_candidate_shard(plade_ctx *ctx, uint32_t rank, uint32_t world, uint32_t min_candidates, plade_exchange_fn exchange,
                                         void *user) {
    if (!ctx || (world > 1 && (rank >= world || !exchange))) return PLADE_EINVAL;
    ctx->shard.rank = world > 1 ? rank : 0; ctx->shard.world = world > 1 ? world : 1; ctx->shard.min_candidates = min_candidates;
    ctx->shard.exchange = world > 1 ? exchange : nullptr; ctx->shard.user = user;
    return PLADE_OK;
}

// Diagnostic (tools/exp_interference.py): `count` launches on this context's stream of a kernel with `blocks` workgroups of 256
// lanes that either return at once (mbytes = 0) or read `mbytes` MB of scratch memory, then a wait.  Used to measure what
// foreign kernel boundaries / workgroup dispatches / memory traffic cost the registrations running beside them.
namespace plade {
__global__ void k_diag_load(const float4 *__restrict__ buf, size_t n16, float *__restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const float4 v = buf[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) *sink = acc;   // never true: keeps the loads
}
}  // namespace plade
extern "C" int plade_diag_launches(plade_ctx *ctx, uint32_t count, uint32_t blocks, uint32_t mbytes) {
    return guarded(ctx, [&]() -> int {
        PLADE_REQUIRE(blocks >= 1 && blocks <= (1u << 20) && mbytes <= 4096, PLADE_EINVAL, "plade_diag_launches: bad argument");
        const size_t n16 = (size_t)mbytes << 16;
        float4 *buf = reinterpret_cast<float4 *>(ctx->scratch[0].ensure(n16 * 16 + 256));
        for (uint32_t i = 0; i < count; ++i)
            hipLaunchKernelGGL(plade::k_diag_load, dim3(blocks), dim3(256), 0, ctx->stream, buf, n16, reinterpret_cast<float *>(buf));
        HIP_TRY(hipGetLastError());
        ctx->sync();
        return PLADE_OK;
    });
}

extern "C" int plade_diag_cluster_order(const float *sizes, uint32_t n, int32_t mode, int32_t depth_limit, int32_t *order) {
    if ((n && (!sizes || !order)) || mode < 0 || mode > 3) return PLADE_EINVAL;
    try {
        std::vector<plade::exact_sort::Item> v(n);
        for (uint32_t i = 0; i < n; ++i) { v[i].index = (int)i; v[i].length = sizes[i]; }
        if (mode == 0) plade::exact_sort::sort_descending(v.data(), v.data() + n);
        else if (mode == 1) std::sort(v.begin(), v.end(), plade::exact_sort::greater);
        else if (mode == 2) plade::exact_sort::sort_descending<true>(v.data(), v.data() + n, depth_limit);
        else plade::exact_sort::sort_descending<false>(v.data(), v.data() + n, depth_limit);
        for (uint32_t i = 0; i < n; ++i) order[i] = v[i].index;
    } catch (...) { return PLADE_ELIMIT; }
    return PLADE_OK;
}

extern "C" int plade_dump_get(plade_ctx *ctx, const char *name, const void **ptr, int64_t *nbytes) {
    if (!ctx || !name || !ptr || !nbytes) return PLADE_EINVAL;
    auto it = ctx->dump.find(name);
    if (it == ctx->dump.end()) return PLADE_EINVAL;
    *ptr = it->second.data();
    *nbytes = (int64_t)it->second.size();
    return PLADE_OK;
}

extern "C" int plade_stats_get(plade_ctx *ctx, const char **names, const double **values, int32_t *count) {
    if (!ctx || !names || !values || !count) return PLADE_EINVAL;
    ctx->stats.joined.clear();
    for (auto &n : ctx->stats.names) { ctx->stats.joined += n; ctx->stats.joined += ";"; }
    *names = ctx->stats.joined.c_str();
    *values = ctx->stats.values.data();
    *count = (int32_t)ctx->stats.values.size();
    return PLADE_OK;
}
