// plade_amd/csrc/cloud.hip -- host cloud (the PLY vertex layout, N x 6 fp32: x y z nx ny nz) -> device:
// the AoS copy as it came + the SoA planes x|y|z|nx|ny|nz every scan kernel reads with 16 B/lane loads, and
// the bounding box (which also refuses non-finite coordinates).  Replaces the copy loop of
// PlaneExtraction::detect (code/PLADE/plane_extraction.cpp:61-90).
#include "ctx.h"
#include "voxel.h"

namespace plade {

__global__ void k_aos_to_soa(const float *__restrict__ aos, uint32_t n, size_t pitch, float *__restrict__ soa) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 *p = reinterpret_cast<const float2 *>(aos + 6 * (size_t)i);
    float2 a = p[0], b = p[1], c = p[2];
    soa[i] = a.x; soa[pitch + i] = a.y; soa[2 * pitch + i] = b.x;
    soa[3 * pitch + i] = b.y; soa[4 * pitch + i] = c.x; soa[5 * pitch + i] = c.y;
}

void cloud_upload(plade_ctx *ctx, const float *pos_nrm, uint32_t n, CloudDev &out) {
    out.n = n;
    out.pitch = ((size_t)n + 3) & ~(size_t)3;
    out.soa.ensure(6 * out.pitch + 4);
    if (n == 0) return;
    float *stage = out.aos.ensure((size_t)n * 6 + 8);
    ctx->h2d(stage, pos_nrm, (size_t)n * 24);
    hipLaunchKernelGGL(k_aos_to_soa, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, stage, n, out.pitch, out.soa.p);
    HIP_TRY(hipGetLastError());
    bbox_host(ctx, stage, n, 6, out.bbmin, out.bbmax);
}

}  // namespace plade
