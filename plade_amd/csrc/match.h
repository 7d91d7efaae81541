// plade_amd/csrc/match.h -- K5 descriptor radius match.
#pragma once
#include "ctx.h"

namespace plade {

struct MatchResult {
    uint64_t total = 0;
    DBuf<int64_t> offsets;            // dq + 1
    DBuf<uint32_t> t_idx;             // total, sorted by (query, dist2, target)
    DBuf<double> dist2;
    const uint32_t *q_idx_sorted = nullptr;  // total: query id per sorted entry
    // scratch
    DBuf<uint32_t> cnt, offs, t_raw, q_raw, v32a, v32b, k32a, k32b, info;
    DBuf<double> d2_raw;
    DBuf<uint64_t> k64a, k64b;
    // windowed path (large tables): both tables sorted by the first descriptor component
    DBuf<float> q_sorted, t_sorted;           // dq x 8, dt x 8
    DBuf<uint32_t> q_perm, t_perm, wk_a, wk_b, wv_a, w_lo, w_cnt, row_tot, row_off;
    uint64_t run_windowed(plade_ctx *ctx, const float *d_qry, uint32_t dq, const float *d_tgt, uint32_t dt, float radius);
    uint64_t run(plade_ctx *ctx, const float *d_qry, uint32_t dq, const float *d_tgt, uint32_t dt, float radius);
};

}  // namespace plade
