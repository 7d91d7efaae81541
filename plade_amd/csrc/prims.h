// plade_amd/csrc/prims.h -- device-wide sort/scan plumbing: radix_sort.hip for the big sorts, rocPRIM (via
// hipCUB) for scans and small sorts; the only translation unit that includes the heavy templates is prims.hip.
#pragma once
#include "ctx.h"

namespace plade {

// stable LSD radix sorts (ascending) of key/value pairs; bits = number of significant key bits
void sort_pairs_u32(plade_ctx *ctx, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                    uint32_t *vals_out, size_t n, int bits = 32);
void sort_pairs_u64(plade_ctx *ctx, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in,
                    uint32_t *vals_out, size_t n, int bits = 64);
// radix_sort.hip: the hand-written onesweep sort behind sort_pairs_* for n above a few thousand
void radix_sort_pairs_u32(plade_ctx *ctx, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                          uint32_t *vals_out, size_t n, int bits);
void radix_sort_pairs_u64(plade_ctx *ctx, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in,
                          uint32_t *vals_out, size_t n, int bits);
// exclusive prefix sums
void exclusive_scan_u32(plade_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n);
void exclusive_scan_u64(plade_ctx *ctx, const uint64_t *in, uint64_t *out, size_t n);

}  // namespace plade
