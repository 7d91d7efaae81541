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
    if (!P.h.p) { bbox_init_pattern(P.h.ensure(8 + 8 * 4)); P.d.ensure(8 * 4); }
    PLADE_REQUIRE(count <= 4, PLADE_EINVAL, "finish_uploads: at most four clouds");
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

void cloud_upload_pair(plade_ctx *ctx, const float *tgt, uint32_t n_t, CloudDev &out_t, const float *src, uint32_t n_s, CloudDev &out_s) {
    shape_cloud(out_t, n_t);
    shape_cloud(out_s, n_s);
    if (n_t) HIP_TRY(hipMemcpyAsync(out_t.aos.p, tgt, (size_t)n_t * 24, hipMemcpyHostToDevice, ctx->stream));
    if (n_s) HIP_TRY(hipMemcpyAsync(out_s.aos.p, src, (size_t)n_s * 24, hipMemcpyHostToDevice, ctx->stream));
    ctx->sync();
    CloudDev *cl[2] = {&out_t, &out_s};
    finish_uploads(ctx, cl, 2);
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

// The upload of the pair the NEXT plade_registration_next call will be handed: the two H2D copies (asynchronous DMA when
// the caller's buffers are page-locked, plade_host_pin) on the prefetch stream and nothing else -- no kernel, no event
// waits behind them in a hardware queue (see above); the call that takes the pair over converts it.
void cloud_prefetch_pair(plade_ctx *ctx, const float *tgt, uint32_t n_t, const float *src, uint32_t n_s) {
    plade_ctx::Prefetch &P = ctx->pf;
    P.valid = false;
    if (!tgt || !src || !n_t || !n_s) return;
    if (!P.stream) HIP_TRY(hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking));
    shape_cloud(P.tgt, n_t);
    shape_cloud(P.src, n_s);
    HIP_TRY(hipMemcpyAsync(P.tgt.aos.p, tgt, (size_t)n_t * 24, hipMemcpyHostToDevice, P.stream));
    HIP_TRY(hipMemcpyAsync(P.src.aos.p, src, (size_t)n_s * 24, hipMemcpyHostToDevice, P.stream));
    P.ptr_t = tgt; P.ptr_s = src; P.n_t = n_t; P.n_s = n_s;
    P.valid = true;
}

// true: the prefetched clouds are exactly this pair; they are now ctx->up_tgt / ctx->up_src (converted, boxes decoded,
// non-finite coordinates refused as cloud_upload does).  false: nothing (usable) was prefetched.
bool cloud_take_prefetched(plade_ctx *ctx, const float *tgt, uint32_t n_t, const float *src, uint32_t n_s) {
    plade_ctx::Prefetch &P = ctx->pf;
    if (!P.valid) return false;
    P.valid = false;
    ctx->sync(P.stream);   // whatever happens next, the prefetch stream must have finished with the buffers
    if (P.ptr_t != tgt || P.ptr_s != src || P.n_t != n_t || P.n_s != n_s) return false;
    swap_clouds(ctx->up_tgt, P.tgt);
    swap_clouds(ctx->up_src, P.src);
    CloudDev *cl[2] = {&ctx->up_tgt, &ctx->up_src};
    finish_uploads(ctx, cl, 2);
    return true;
}

}  // namespace plade
