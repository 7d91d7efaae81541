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
    // the lists as ascending positions in a Morton-ordered SoA copy of the cloud (PlaneSetOut::d_pos), or null
    const uint32_t *d_pos = nullptr;
    const float *m_x = nullptr, *m_y = nullptr, *m_z = nullptr;
    uint32_t P = 0;
    // unoriented-normals mode: planes [P/2, P) are planes [0, P/2) with (n, d) negated and the same supports; d_idx then
    // holds the supports of the first half only (offsets[P/2] items)
    bool mirrored = false;
};

// The "unoriented normals" mode the reference's README describes (README.md:109-110): "allowing a plane (a group of 3D
// points) to have two opposite orientations.  This way, more descriptors (considering both orientations for each plane)
// will be generated and matched."  Every plane of the TARGET takes part twice, as (n, d) and as (-n, -d), with the same
// support: the target's descriptor table then holds every sign pattern of every pair of intersection lines, so a source
// pair finds its counterpart whatever signs the source planes came out with, and the plane-consistency count finds an
// aligned copy of every matching target plane.  (Mirroring the source as well would only repeat each candidate.)
struct MirroredPlanes {
    std::vector<float> coef;
    std::vector<int32_t> offsets, idx;
    // from a plane set (idx may be null when the lists live on the device only)
    void build(const float *coef_in, const int32_t *offsets_in, const int32_t *idx_in, uint32_t P);
};

struct RegistrationWork;
RegistrationWork *registration_work_create();
// where the whole-cloud voxel grid of the pair's next registration goes when the caller builds it for several pairs at once
// (voxel_whole_batch): the side's VoxelWork, its SoA output (>= 3 n floats) and the flag prepare_side consumes
struct VoxelWork;
struct ObbWork;
struct WholeVoxelSlot { VoxelWork *work; float *out_soa; bool *ready; VoxelWork *planes; bool *planes_ready; ObbWork *obb; bool *obb_ready; };
WholeVoxelSlot whole_voxel_slot(RegistrationWork &W, bool target, uint32_t n);
void registration_work_destroy(RegistrationWork *w);

// registration(T, target, source, target_planes, source_planes) (code/PLADE/plade.cpp:31-580).
// Returns false where the reference returns false.
double ctx_stat(plade_ctx *ctx, const char *name);

bool run_registration(plade_ctx *ctx, RegistrationWork &W, const CloudDev &tgt, const CloudDev &src,
                      const PlaneSetView &tp, const PlaneSetView &sp, float *T16_out, const float *spacing_or_null);
// average point spacing of the source cloud (plade.cpp:41); may run ahead of run_registration on another ctx
float source_spacing(plade_ctx *ctx, RegistrationWork &W, const CloudDev &src);

}  // namespace plade
