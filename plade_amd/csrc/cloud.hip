// plade_amd/csrc/cloud.hip -- host cloud (the PLY vertex layout, N x 6 fp32: x y z nx ny nz) -> device:
// the AoS copy as it came + the SoA planes x|y|z|nx|ny|nz every scan kernel reads with 16 B/lane loads, and
// the bounding box (which also refuses non-finite coordinates).  Replaces the copy loop of
// PlaneExtraction::detect (code/PLADE/plane_extraction.cpp:61-90).
#include "ctx.h"
#include "voxel.h"
#include <utility>

namespace plade {

__global__ void k_aos_to_soa(const float *__restrict__ aos, uint32_t n, size_t pitch, float *__restrict__ soa) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 *p = reinterpret_cast<const float2 *>(aos + 6 * (size_t)i);
    float2 a = p[0], b = p[1], c = p[2];
    soa[i] = a.x; soa[pitch + i] = a.y; soa[2 * pitch + i] = b.x;
    soa[3 * pitch + i] = b.y; soa[4 * pitch + i] = c.x; soa[5 * pitch + i] = c.y;
}

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
void convert_on(hipStream_t st, CloudDev &c) {
    if (c.n) hipLaunchKernelGGL(k_aos_to_soa, dim3(cdiv(c.n, 256)), dim3(256), 0, st, c.aos.p, c.n, c.pitch, c.soa.p);
}
// SoA conversion + bounding boxes of clouds whose AoS copy is complete: ONE wait for all of them
void finish_uploads(plade_ctx *ctx, CloudDev *const clouds[], int count) {
    plade_ctx::Prefetch &P = ctx->pf;
    constexpr int MAXC = 2 * PLADE_GROUP_MAX;
    if (!P.h.p) { bbox_init_pattern(P.h.ensure(8 + 8 * MAXC)); P.d.ensure(8 * MAXC); }
    PLADE_REQUIRE(count <= MAXC, PLADE_EINVAL, "finish_uploads: too many clouds");
    for (int i = 0; i < count; ++i) {
        convert_on(ctx->stream, *clouds[i]);
        bbox_async(ctx->stream, clouds[i]->aos.p, clouds[i]->n, 6, P.d.p + 8 * i, P.h.p, P.h.p + 8 + 8 * i);
    }
    HIP_TRY(hipGetLastError());
    ctx->sync();
    for (int i = 0; i < count; ++i) bbox_decode(P.h.p + 8 + 8 * i, clouds[i]->bbmin, clouds[i]->bbmax);
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
