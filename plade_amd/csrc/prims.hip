// plade_amd/csrc/prims.hip -- rocPRIM-backed sort/scan wrappers (plumbing, not a hot op).
#include "prims.h"
#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

namespace plade {

static void *temp(plade_ctx *ctx, size_t bytes) { return ctx->scratch[7].ensure(bytes + 256); }

// rocPRIM's default switches from its merge sort to the onesweep radix sort above 1 Mi items; with the
// bit range known (our keys are packed cell / Morton / index fields) onesweep needs ceil(bits/8)
// passes and wins far earlier on gfx950, so the switch point is lowered.
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 16384>;

template <class K>
static void sort_pairs(plade_ctx *ctx, const K *ki, K *ko, const uint32_t *vi, uint32_t *vo, size_t n, int bits) {
    if (!n) return;
    PLADE_REQUIRE(n < (1ull << 31), PLADE_ELIMIT, "sort: too many items");
    size_t tb = 0;
    HIP_TRY(rocprim::radix_sort_pairs<SortConfig>(nullptr, tb, ki, ko, vi, vo, (unsigned int)n, 0u, (unsigned int)bits, ctx->stream));
    void *t = temp(ctx, tb);
    HIP_TRY(rocprim::radix_sort_pairs<SortConfig>(t, tb, ki, ko, vi, vo, (unsigned int)n, 0u, (unsigned int)bits, ctx->stream));
}

// below this size rocPRIM's single-launch block / merge sort is the cheaper choice
constexpr size_t RS_MIN_ITEMS = 16384;
static bool use_rocprim() { static const bool v = getenv("PLADE_SORT_ROCPRIM") != nullptr; return v; }

void sort_pairs_u32(plade_ctx *ctx, const uint32_t *ki, uint32_t *ko, const uint32_t *vi, uint32_t *vo, size_t n,
                    int bits) {
    if (n > RS_MIN_ITEMS && !use_rocprim()) radix_sort_pairs_u32(ctx, ki, ko, vi, vo, n, bits);
    else sort_pairs<uint32_t>(ctx, ki, ko, vi, vo, n, bits);
}

void sort_pairs_u64(plade_ctx *ctx, const uint64_t *ki, uint64_t *ko, const uint32_t *vi, uint32_t *vo, size_t n,
                    int bits) {
    if (n > RS_MIN_ITEMS && !use_rocprim()) radix_sort_pairs_u64(ctx, ki, ko, vi, vo, n, bits);
    else sort_pairs<uint64_t>(ctx, ki, ko, vi, vo, n, bits);
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
