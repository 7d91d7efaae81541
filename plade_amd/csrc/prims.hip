// plade_amd/csrc/prims.hip -- rocPRIM-backed sort/scan wrappers (plumbing, not a hot op).
#include "prims.h"
#include <hipcub/hipcub.hpp>

namespace plade {

static void *temp(plade_ctx *ctx, size_t bytes) { return ctx->scratch[7].ensure(bytes + 256); }

void sort_pairs_u32(plade_ctx *ctx, const uint32_t *ki, uint32_t *ko, const uint32_t *vi, uint32_t *vo, size_t n,
                    int bits) {
    if (!n) return;
    size_t tb = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, ki, ko, vi, vo, (int)n, 0, bits, ctx->stream));
    void *t = temp(ctx, tb);
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(t, tb, ki, ko, vi, vo, (int)n, 0, bits, ctx->stream));
}

void sort_pairs_u64(plade_ctx *ctx, const uint64_t *ki, uint64_t *ko, const uint32_t *vi, uint32_t *vo, size_t n,
                    int bits) {
    if (!n) return;
    size_t tb = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, ki, ko, vi, vo, (int)n, 0, bits, ctx->stream));
    void *t = temp(ctx, tb);
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(t, tb, ki, ko, vi, vo, (int)n, 0, bits, ctx->stream));
}

void exclusive_scan_u32(plade_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n) {
    if (!n) return;
    size_t tb = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, (int)n, ctx->stream));
    void *t = temp(ctx, tb);
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(t, tb, in, out, (int)n, ctx->stream));
}

void exclusive_scan_u64(plade_ctx *ctx, const uint64_t *in, uint64_t *out, size_t n) {
    if (!n) return;
    size_t tb = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, (int)n, ctx->stream));
    void *t = temp(ctx, tb);
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(t, tb, in, out, (int)n, ctx->stream));
}

}  // namespace plade
