// plade_amd/csrc/pipeline.h -- registration pipeline entry points shared by the C ABI wrappers.
#pragma once
#include "ctx.h"
#include "stages.h"
#include "match.h"
#include "overlap.h"
#include "voxel.h"

namespace plade {

// Extracted planes of one cloud: coef P x 4 = (unit n, d), offsets P+1, support point indices
// (host arrays; d_idx optionally the same index list already resident on the device).
struct PlaneSetView {
    const float *coef = nullptr;
    const int32_t *offsets = nullptr;
    const int32_t *idx = nullptr;
    const uint32_t *d_idx = nullptr;
    uint32_t P = 0;
};

struct RegistrationWork;
RegistrationWork *registration_work_create();
void registration_work_destroy(RegistrationWork *w);

// registration(T, target, source, target_planes, source_planes) (code/PLADE/plade.cpp:31-580).
// Returns false where the reference returns false.
double ctx_stat(plade_ctx *ctx, const char *name);

bool run_registration(plade_ctx *ctx, RegistrationWork &W, const CloudDev &tgt, const CloudDev &src,
                      const PlaneSetView &tp, const PlaneSetView &sp, float *T16_out, const float *spacing_or_null);
// average point spacing of the source cloud (plade.cpp:41); may run ahead of run_registration on another ctx
float source_spacing(plade_ctx *ctx, RegistrationWork &W, const CloudDev &src);

}  // namespace plade
