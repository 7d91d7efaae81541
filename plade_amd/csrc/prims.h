// plade_amd/csrc/prims.h -- device-wide sort/scan plumbing: radix_sort.hip for the sorts, a single-launch scan.
#pragma once
#include "ctx.h"

namespace plade {

// stable LSD radix sorts (ascending) of key/value pairs; bits = number of significant key bits
void sort_pairs_u32(plade_ctx *ctx, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                    uint32_t *vals_out, size_t n, int bits = 32);
void sort_pairs_u64(plade_ctx *ctx, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in,
                    uint32_t *vals_out, size_t n, int bits = 64);
// radix_sort.hip: the hand-written onesweep sort behind sort_pairs_*
void radix_sort_pairs_u32(plade_ctx *ctx, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                          uint32_t *vals_out, size_t n, int bits);
void radix_sort_pairs_u64(plade_ctx *ctx, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in,
                          uint32_t *vals_out, size_t n, int bits);
// up to 16 arrays in ONE launch sequence: segment s = items [seg_off[s], seg_off[s + 1]) of the same buffers, each sorted on
// its own (stable) and left in its own range
void radix_sort_segments_u32(plade_ctx *ctx, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                             uint32_t *vals_out, const uint32_t *seg_off, int nseg, int bits);
// exclusive prefix sum, one launch (decoupled look-back)
void exclusive_scan_u32(plade_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n);
// Look-back words + tile ticket for ONE launch of a decoupled look-back kernel over `n` items in tiles of `tile_items`
// (the scan above, the voxel-run kernel): a tile takes ticket = atomicAdd(ticket, 1) - base, its word is state[tile],
// tagged with gen in bits 34..63, status in bits 32..33 (1 aggregate, 2 inclusive prefix), value in bits 0..31.
struct ScanTicket { uint64_t *state; uint32_t *ticket; uint32_t base, gen, tiles; };
ScanTicket scan_ticket(plade_ctx *ctx, size_t n, uint32_t tile_items);

}  // namespace plade
