// plade_amd/csrc/cloud.hip -- host cloud (the PLY vertex layout, N x 6 fp32: x y z nx ny nz) -> device:
// the AoS copy as it came + the SoA planes x|y|z|nx|ny|nz every scan kernel reads with 16 B/lane loads, and
// the bounding box (which also refuses non-finite coordinates).  Replaces the copy loop of
// PlaneExtraction::detect (code/PLADE/plane_extraction.cpp:61-90).
#include "ctx.h"
#include "voxel.h"
#include <utility>

namespace plade {

// A large H2D copy runs on an SDMA engine; a kernel queued BEHIND it on the same stream waits in its hardware queue
// behind a barrier packet until the engine signals -- and with it every other stream that shares that hardware queue
// (the runtime maps all streams of a process onto four of them).  Measured: two 24 MB uploads per registration with
// their conversion kernels queued directly behind cost 11 % of the throughput of eight registrations in flight, the
// same DMA traffic without dependent kernels costs nothing.  Hence: copies first, the HOST waits for them (sleeping
// polls in throughput mode), and only then are the kernels that read the data queued.
namespace {
void shape_cloud(CloudDev &out, uint32_t n) {
    out.n = n;
    out.pitch = ((size_t)n + 3) & ~(size_t)3;
    out.soa.ensure(6 * out.pitch + 4);
    out.aos.ensure((size_t)n * 6 + 8);
}
// SoA conversion AND bounding box (+ the refusal of non-finite coordinates) of up to 2 * PLADE_GROUP_MAX uploaded clouds in
// ONE launch: every point is read once (r3: a conversion kernel and a min/max kernel per cloud, each reading the 24 B/point,
// with an 32-byte copy in front of and behind every min/max kernel: 4 commands per cloud).  grid (workgroups, clouds).
struct FinishArgs {
    const float *aos[2 * PLADE_GROUP_MAX];
    float *soa[2 * PLADE_GROUP_MAX];
    uint32_t n[2 * PLADE_GROUP_MAX];
    uint32_t pitch[2 * PLADE_GROUP_MAX];
};
__global__ __launch_bounds__(256) void k_finish_uploads(const FinishArgs A, int *__restrict__ slots /* 8 ints per cloud, initialised */) {
    __shared__ float s_lds[6][8];
    const int c = blockIdx.y;
    const uint32_t n = A.n[c];
    const float *__restrict__ aos = A.aos[c];
    float *__restrict__ soa = A.soa[c];
    const size_t pitch = A.pitch[c];
    int *out = slots + 8 * c;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float2 *p = reinterpret_cast<const float2 *>(aos + 6 * (size_t)i);
        const float2 a = p[0], b = p[1], d = p[2];
        soa[i] = a.x; soa[pitch + i] = a.y; soa[2 * pitch + i] = b.x;
        soa[3 * pitch + i] = b.y; soa[4 * pitch + i] = d.x; soa[5 * pitch + i] = d.y;
        const float v[3] = {a.x, a.y, b.x};
        for (int k = 0; k < 3; ++k) {
            bad = bad || !(fabsf(v[k]) <= FLT_MAX);   // NaN or infinity (fminf / fmaxf would hide a NaN)
            mn[k] = fminf(mn[k], v[k]);
            mx[k] = fmaxf(mx[k], v[k]);
        }
    }
    if (bad) out[6] = 1;
    block_minmax_commit<3>(mn, mx, out, s_lds);
}

// SoA conversion + bounding boxes of clouds whose AoS copy is complete: one small upload (the slots' initial pattern), one
// kernel, one read-back, ONE wait for all of them
void finish_uploads(plade_ctx *ctx, CloudDev *const clouds[], int count) {
    plade_ctx::Prefetch &P = ctx->pf;
    constexpr int MAXC = 2 * PLADE_GROUP_MAX;
    if (!P.h.p) {
        int *h = P.h.ensure(8 * MAXC + 8 * MAXC);   // [0, 8 MAXC): the init pattern of every slot; then the slots read back
        for (int i = 0; i < MAXC; ++i) bbox_init_pattern(h + 8 * i);
        P.d.ensure(8 * MAXC);
    }
    PLADE_REQUIRE(count >= 1 && count <= MAXC, PLADE_EINVAL, "finish_uploads: too many clouds");
    FinishArgs A;
    memset(&A, 0, sizeof(A));
    uint32_t n_max = 0;
    for (int i = 0; i < count; ++i) {
        A.aos[i] = clouds[i]->aos.p; A.soa[i] = clouds[i]->soa.p; A.n[i] = clouds[i]->n; A.pitch[i] = (uint32_t)clouds[i]->pitch;
        n_max = std::max(n_max, clouds[i]->n);
    }
    int *h_out = P.h.p + 8 * MAXC;
    HIP_TRY(hipMemcpyAsync(P.d.p, P.h.p, 32 * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    if (n_max) hipLaunchKernelGGL(k_finish_uploads, dim3(std::min(cdiv(n_max, 1024), 1024u), count), dim3(256), 0, ctx->stream, A, P.d.p);
    HIP_TRY(hipMemcpyAsync(h_out, P.d.p, 32 * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipGetLastError());
    ctx->sync();
    for (int i = 0; i < count; ++i) bbox_decode(h_out + 8 * i, clouds[i]->bbmin, clouds[i]->bbmax);
}
}  // namespace

void cloud_upload(plade_ctx *ctx, const float *pos_nrm, uint32_t n, CloudDev &out) {
    shape_cloud(out, n);
    if (n == 0) return;
    HIP_TRY(hipMemcpyAsync(out.aos.p, pos_nrm, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    ctx->sync();
    CloudDev *cl[1] = {&out};
    finish_uploads(ctx, cl, 1);
}

void cloud_upload_many(plade_ctx *ctx, int count, const float *const ptr[], const uint32_t n[], CloudDev *const out[]) {
    PLADE_REQUIRE(count >= 1 && count <= 2 * PLADE_GROUP_MAX, PLADE_EINVAL, "cloud_upload_many: too many clouds");
    for (int i = 0; i < count; ++i) {
        shape_cloud(*out[i], n[i]);
        if (n[i]) HIP_TRY(hipMemcpyAsync(out[i]->aos.p, ptr[i], (size_t)n[i] * 24, hipMemcpyHostToDevice, ctx->stream));
    }
    ctx->sync();
    CloudDev *cl[2 * PLADE_GROUP_MAX];
    for (int i = 0; i < count; ++i) cl[i] = out[i];
    finish_uploads(ctx, cl, count);
}

void cloud_upload_pair(plade_ctx *ctx, const float *tgt, uint32_t n_t, CloudDev &out_t, const float *src, uint32_t n_s, CloudDev &out_s) {
    const float *ptr[2] = {tgt, src};
    const uint32_t n[2] = {n_t, n_s};
    CloudDev *out[2] = {&out_t, &out_s};
    cloud_upload_many(ctx, 2, ptr, n, out);
}

namespace {
void swap_clouds(CloudDev &a, CloudDev &b) {
    std::swap(a.n, b.n);
    std::swap(a.pitch, b.pitch);
    a.soa.swap(b.soa);
    a.aos.swap(b.aos);
    for (int k = 0; k < 3; ++k) { std::swap(a.bbmin[k], b.bbmin[k]); std::swap(a.bbmax[k], b.bbmax[k]); }
}
}  // namespace

void cloud_drop_prefetch(plade_ctx *ctx) {
    plade_ctx::Prefetch &P = ctx->pf;
    if (!P.valid) return;
    P.valid = false;
    if (P.stream) ctx->sync(P.stream);
}

// The upload of the clouds the NEXT batch-mode call will be handed: the H2D copies (asynchronous DMA when the caller's
// buffers are page-locked, plade_host_pin) on the prefetch stream and nothing else -- no kernel, no event waits behind them
// in a hardware queue (see above); the call that takes the clouds over converts them.
void cloud_prefetch(plade_ctx *ctx, int count, const float *const ptr[], const uint32_t n[]) {
    plade_ctx::Prefetch &P = ctx->pf;
    cloud_drop_prefetch(ctx);   // an earlier prefetch nobody took is still writing into the buffers reshaped below
    if (count < 2 || count > 2 * PLADE_GROUP_MAX || (count & 1)) return;
    for (int i = 0; i < count; ++i) if (!ptr[i] || !n[i]) return;
    if (!P.stream) HIP_TRY(hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking));
    for (int i = 0; i < count; ++i) {
        shape_cloud(P.cl[i], n[i]);
        HIP_TRY(hipMemcpyAsync(P.cl[i].aos.p, ptr[i], (size_t)n[i] * 24, hipMemcpyHostToDevice, P.stream));
        P.ptr[i] = ptr[i]; P.n[i] = n[i];
    }
    P.count = count;
    P.valid = true;
}

// true: the prefetched clouds are exactly these; they are now out[] (converted, boxes decoded, non-finite coordinates refused as
// cloud_upload does).  false: nothing (usable) was prefetched.
bool cloud_take_prefetched(plade_ctx *ctx, int count, const float *const ptr[], const uint32_t n[], CloudDev *const out[]) {
    plade_ctx::Prefetch &P = ctx->pf;
    if (!P.valid) return false;
    P.valid = false;
    ctx->sync(P.stream);   // whatever happens next, the prefetch stream must have finished with the buffers
    if (P.count != count) return false;
    for (int i = 0; i < count; ++i) if (P.ptr[i] != ptr[i] || P.n[i] != n[i]) return false;
    CloudDev *cl[2 * PLADE_GROUP_MAX];
    for (int i = 0; i < count; ++i) { swap_clouds(*out[i], P.cl[i]); cl[i] = out[i]; }
    finish_uploads(ctx, cl, count);
    return true;
}

}  // namespace plade
