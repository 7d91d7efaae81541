// plade_amd/csrc/pipeline.hip -- host orchestration of
//   registration(T, target, source, target_planes, source_planes)   code/PLADE/plade.cpp:31-580
// on one MI355X: every data-parallel stage is a HIP kernel (K1..K9, see the k_*.hip files); the host
// keeps what the reference's host keeps -- control flow, the per-plane PCA boxes of the (small)
// voxel-downsampled clouds, plane/plane intersection lines (P^2/2 of them) and the two std::sort
// calls whose tie order is part of the result.
#include "pipeline.h"
#include "hostgeom.h"
#include "prims.h"
#include "exact_sort.h"
#include <algorithm>
#include <numeric>
#include <thread>

namespace plade {

namespace {

struct LengthIndex { float length; int index; };   // util.h:347-350
inline bool cmp_greater(const LengthIndex &a, const LengthIndex &b) { return a.length > b.length; }  // util.h:360-365

__device__ void k_gather_rt(const VB &vb, const float4 *__restrict__ rt, const uint32_t *__restrict__ ids, uint32_t n,
                            float *__restrict__ out /* n x 12 */) {
    const uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = ids[i];
    const float4 r0 = rt[4 * (size_t)k], r1 = rt[4 * (size_t)k + 1], r2 = rt[4 * (size_t)k + 2];
    float *o = out + 12 * (size_t)i;
    o[0] = r0.x; o[1] = r0.y; o[2] = r0.z; o[3] = r1.x; o[4] = r1.y; o[5] = r1.z; o[6] = r2.x; o[7] = r2.y; o[8] = r2.z;
    o[9] = r0.w; o[10] = r1.w; o[11] = r2.w;
}

// per-cloud preparation: code/PLADE/plade.cpp:75-172 (target) / :290-381 (source)
struct Side {
    std::vector<float> ds;       // host copy of the whole-cloud downsample, n_ds x 3
    DBuf<float> d_ds;            // device AoS
    DBuf<float> d_ds_soa;        // device SoA x|y|z
    uint32_t n_ds = 0;
    f3 bcenter;
    double radius = 0;
    PlaneGeomHost geom;
    PlaneCloudsDev pcl;
    std::vector<float> plane_ds; // host copy
    std::vector<float> normals;  // P x 3
    LineTableHost lines;
    VoxelWork vox_all, vox_planes;
    bool vox_all_ready = false;  // the whole-cloud grid of the coming registration was queued with the group's (voxel_whole_batch)
    bool vox_planes_ready = false;   // ... and the per-plane grids
    bool obb_ready = false;          // ... and the boxes (obb_units_batch)
    ObbWork obb;
    DBuf<uint32_t> d_items, d_offs;
};

void make_line_table(const float *coef, uint32_t P, f3 bcenter, double radius, LineTableHost &lt) {
    lt = LineTableHost();
    for (uint32_t i = 0; i < P; ++i)
        for (uint32_t j = i + 1; j < P; ++j) {
            f3 vec, pt;
            if (!plane_plane_line(coef + 4 * (size_t)i, coef + 4 * (size_t)j, vec, pt)) continue;  // plade.cpp:133-136
            // plade.cpp:137-142: drop lines farther than `radius` from the bounding-box centre
            const f3 tv = pt - bcenter;
            const double dt = dot_e(tv, vec);
            const double distance = std::sqrt((double)sqn_e(tv) - dt * dt);
            if (distance > radius) continue;
            // two ComputeMeanDistanceOfLine2Plane calls (plade.cpp:146-154 -> util.h:396) re-normalise the vector
            f3 v3 = normalized_e(normalized_e(vec));
            // iterates 3, 4, ... until the sequence repeats
            std::vector<f3> its;
            its.push_back(v3);
            int start = -1;
            for (int it = 0; it < 64 && start < 0; ++it) {
                f3 nx = normalized_e(its.back());
                for (size_t s = 0; s < its.size(); ++s)
                    if (its[s].x == nx.x && its[s].y == nx.y && its[s].z == nx.z) { start = (int)s; break; }
                if (start < 0) its.push_back(nx);
            }
            PLADE_REQUIRE(start >= 0, PLADE_ELIMIT, "line direction normalisation did not settle");
            lt.it_off.push_back((int32_t)(lt.iter.size() / 3));
            lt.it_len.push_back((int32_t)its.size());
            lt.it_start.push_back(start);
            lt.it_period.push_back((int32_t)its.size() - start);
            for (auto &v : its) { lt.iter.push_back(v.x); lt.iter.push_back(v.y); lt.iter.push_back(v.z); }
            lt.pt.push_back(pt.x); lt.pt.push_back(pt.y); lt.pt.push_back(pt.z);
            lt.sp.push_back((int32_t)i); lt.sp.push_back((int32_t)j);
            ++lt.L;
        }
}

bool prepare_side(plade_ctx *ctx, const char *tag, const CloudDev &cloud, const PlaneSetView &pl, float leaf, float pen_cell,
                  float desc_scale, bool is_target, PairTableDev &pairs, Side &S, bool finish = true) {
    const uint32_t P = pl.P;
    Clock::time_point tp0 = Clock::now();
    // whole cloud: DownSamplePointCloud (plade.cpp:77-79 / :292-294) and the per-plane clouds in one pass
    // (plade.cpp:93-105 / :308-319): both voxel-grid runs are queued before the host waits for either
    S.d_ds_soa.ensure(3 * (size_t)cloud.n + 4);   // the centroid kernel writes the SoA copy too (pitch = voxel count)
    if (S.vox_all_ready) { S.vox_all_ready = false; S.vox_all.adopt_batch(ctx); }   // queued with the group's grids (registration.hip)
    else
    S.vox_all.enqueue(ctx, cloud.aos.p, 6, cloud.x(), cloud.y(), cloud.z(), nullptr, nullptr, cloud.n, 1, leaf, cloud.bbmin, cloud.bbmax,
                      false, false, S.d_ds_soa.p);
    const uint32_t n_items = (uint32_t)pl.offsets[P];
    S.d_items.ensure((size_t)n_items + 4); S.d_offs.ensure(5 * (size_t)P + 4);
    // Planes that come from the GPU extraction are read through their Morton positions from the extraction's Morton-ordered
    // copy (the lists are ascending positions: near-sequential reads) instead of through their point indices from the
    // cloud in input order (random 12-byte reads of 128-byte lines); same coordinates, same order, same sums
    const bool by_pos = pl.d_pos && pl.m_x;
    const uint32_t *dev_list = by_pos ? pl.d_pos : pl.d_idx;
    const uint32_t *items = S.d_items.p;
    if (dev_list && pl.mirrored) {   // the mirrored half shares the supports of the first half
        const size_t half = (size_t)pl.offsets[P / 2];
        ctx->copy_dd_async(S.d_items.p, dev_list, 4 * half);
        ctx->copy_dd_async(S.d_items.p + half, dev_list, 4 * half);
    } else if (dev_list) items = dev_list;   // read where the extraction left it (valid until this cloud slot's next detect)
    else ctx->h2d(S.d_items.p, pl.idx, 4 * (size_t)n_items);
    const bool batched = S.vox_planes_ready && S.obb_ready && by_pos && !pl.mirrored;   // grids and boxes came with the group's
    if (!batched) {   // one upload: the planes' item offsets | their coefficients (for the per-plane boxes below)
        std::vector<uint32_t> up(5 * (size_t)P + 1);
        memcpy(up.data(), pl.offsets, 4 * ((size_t)P + 1));
        memcpy(up.data() + P + 1, pl.coef, 16 * (size_t)P);
        ctx->h2d(S.d_offs.p, up.data(), 4 * up.size());
    }
    if (S.vox_planes_ready && by_pos && !pl.mirrored) { S.vox_planes_ready = false; S.vox_planes.adopt_batch(ctx); }
    else
    S.vox_planes.enqueue(ctx, cloud.aos.p, 6, by_pos ? pl.m_x : nullptr, by_pos ? pl.m_y : nullptr, by_pos ? pl.m_z : nullptr, items,
                         S.d_offs.p, n_items, P, leaf, cloud.bbmin, cloud.bbmax, by_pos, true);   // groups = the planes' item ranges
    // ComputeBoundingBox of the whole downsampled cloud (plade.cpp:81-84 / :295-299) and per plane (plade.cpp:106-117 /
    // :320-330) on the device, reading the voxel grids' results where they lie; ONE wait for the grids' sizes, the
    // per-plane offsets and the boxes
    if (S.obb_ready && !pl.mirrored) { S.obb_ready = false; obb_adopt_batch(ctx, S.obb, P); }   // queued with the group's boxes
    else
    obb_units(ctx, S.obb, S.vox_all.out_xyz.p, S.vox_all.count.p, cloud.n, S.vox_planes.out_xyz.p, S.vox_planes.group_offsets.p, n_items, P,
              pl.coef, reinterpret_cast<const float *>(S.d_offs.p + P + 1));
    S.pcl.off.resize((size_t)P + 1);
    ctx->d2h(S.pcl.off.data(), S.vox_planes.group_offsets.p, 4 * ((size_t)P + 1));
    S.n_ds = S.vox_all.finish(ctx);
    const uint32_t n_pds = S.vox_planes.finish(ctx);
    if (n_items == 0) std::fill(S.pcl.off.begin(), S.pcl.off.end(), 0u);
    ctx->stats.add(std::string("t_prep_voxel_") + tag, secs_since(tp0));
    tp0 = Clock::now();
    if (S.n_ds == 0) return false;
    S.d_ds.swap(S.vox_all.out_xyz);   // the result becomes d_ds, the work area takes last call's buffer back
    S.pcl.xyz.swap(S.vox_planes.out_xyz);
    S.pcl.d_off.swap(S.vox_planes.group_offsets);
    const float *ob = S.obb.host.data();
    if (ob[OBB_OUT_WHOLE - 1] == 0.f) return false;
    S.bcenter = f3(ob[0], ob[1], ob[2]);
    memcpy(&S.radius, ob + 4, 8);
    S.geom.P = P;
    S.geom.coef.assign(pl.coef, pl.coef + 4 * (size_t)P);
    S.geom.center.assign(3 * (size_t)P, 0.f);
    S.geom.radius.assign(P, 0.f);
    S.geom.four.assign(12 * (size_t)P, 0.f);
    S.normals.resize(3 * (size_t)P);
    for (uint32_t i = 0; i < P; ++i) {
        for (int k = 0; k < 3; ++k) S.normals[3 * i + k] = pl.coef[4 * (size_t)i + k];
        const float *o = ob + OBB_OUT_WHOLE + (size_t)i * OBB_OUT_PLANE;
        for (int k = 0; k < 12; ++k) S.geom.four[12 * (size_t)i + k] = o[k];
        for (int k = 0; k < 3; ++k) S.geom.center[3 * i + k] = o[12 + k];
        S.geom.radius[i] = o[15];
    }
    if (ctx->params.dump) {   // host copies of the downsampled clouds only for the dump
        S.ds.resize(3 * (size_t)S.n_ds);
        S.plane_ds.resize(3 * (size_t)n_pds);
        ctx->d2h(S.ds.data(), S.d_ds.p, 12 * (size_t)S.n_ds);
        ctx->d2h(S.plane_ds.data(), S.pcl.xyz.p, 12 * (size_t)n_pds);
        ctx->sync();
    }
    ctx->stats.add(std::string("t_prep_obb_") + tag, secs_since(tp0));
    tp0 = Clock::now();
    S.pcl.grid_cell = 0.f;
    build_pen_grid(ctx, S.pcl, S.geom, pen_cell);   // in-plane grids for the penetration walk (A11)
    make_line_table(pl.coef, P, S.bcenter, S.radius, S.lines);
    // line-pair descriptors of this side (K4); both sides build theirs concurrently
    build_pair_table(ctx, S.lines, S.normals.data(), P, desc_scale, is_target, pairs);
    // everything this side produced is consumed on the OTHER stream (match, transforms, penetration run on the
    // main stream, the source side is prepared on the auxiliary one): finish it before handing over.  (One side after the other
    // on ONE stream: the first side's last read-back -- the size of its descriptor table -- rides on the second side's waits.)
    if (finish || ctx->params.dump) ctx->sync();
    ctx->stats.add(std::string("t_prep_lines_") + tag, secs_since(tp0));
    if (ctx->params.dump) {
        const std::string t(tag);
        ctx->put(t + "_ds", S.ds.data(), S.ds.size());
        ctx->put(t + "_bcenter", &S.bcenter.x, 3);
        ctx->put1(t + "_radius", S.radius);
        std::vector<float> pcr(4 * (size_t)P);
        for (uint32_t i = 0; i < P; ++i) { for (int k = 0; k < 3; ++k) pcr[4 * i + k] = S.geom.center[3 * i + k]; pcr[4 * i + 3] = S.geom.radius[i]; }
        ctx->put(t + "_plane_center_radius", pcr.data(), pcr.size());
        ctx->put(t + "_plane_four", S.geom.four.data(), S.geom.four.size());
        std::vector<int32_t> off(S.pcl.off.begin(), S.pcl.off.end());
        ctx->put(t + "_plane_ds_offsets", off.data(), off.size());
        ctx->put(t + "_plane_ds", S.plane_ds.data(), S.plane_ds.size());
        std::vector<float> ld(8 * (size_t)S.lines.L);
        for (uint32_t l = 0; l < S.lines.L; ++l) {
            const float *v = &S.lines.iter[3 * (size_t)S.lines.it_off[l]];
            ld[8 * l] = v[0]; ld[8 * l + 1] = v[1]; ld[8 * l + 2] = v[2];
            ld[8 * l + 3] = S.lines.pt[3 * l]; ld[8 * l + 4] = S.lines.pt[3 * l + 1]; ld[8 * l + 5] = S.lines.pt[3 * l + 2];
            ld[8 * l + 6] = (float)S.lines.sp[2 * l]; ld[8 * l + 7] = (float)S.lines.sp[2 * l + 1];
        }
        ctx->put(t + "_lines", ld.data(), ld.size());
    }
    return true;
}

}  // namespace

struct RegistrationWork {
    Side M, C;  // main (target), current (source)
    PairTableDev tgt_pairs, src_pairs;
    MatchResult match;
    CandidateSet cand;
    TargetGrid grid, sp_grid;
    DBuf<float> d_rt12, d_T16;      // d_T16: transforms | centres | overlap counts | sphere flags of the verified candidates
    DBuf<int32_t> d_shard_all;      // candidate shard over RCCL: every rank's (counts | flags) block after the all-gather
    OverlapWork ov_work;
    // host side of the cluster stage: ~30 000 clusters per registration, five arrays of them -- kept from call to call (a fresh
    // std::vector of 100-200 KB per call is an mmap, a page fault per 4 KB and an munmap: ~0.3 ms of system time per registration)
    struct ClusterHost {
        std::vector<uint32_t> seeds, sizes;
        std::vector<int32_t> pcounts;
        std::vector<int> cand_cluster, match_counts;
        std::vector<exact_sort::Item> sort_vec;
    } ch;
    hipEvent_t ev_grid = nullptr;   // target grid of the verification built on the auxiliary stream
    hipEvent_t ev_main = nullptr;   // everything queued on the main stream before the two sides are prepared
    ~RegistrationWork() { if (ev_grid) (void)hipEventDestroy(ev_grid); if (ev_main) (void)hipEventDestroy(ev_main); }
};

void MirroredPlanes::build(const float *coef_in, const int32_t *offsets_in, const int32_t *idx_in, uint32_t P) {
    coef.assign(coef_in, coef_in + 4 * (size_t)P);
    for (size_t i = 0; i < 4 * (size_t)P; ++i) coef.push_back(-coef_in[i]);
    offsets.assign(offsets_in, offsets_in + P + 1);
    const int32_t total = offsets_in[P];
    for (uint32_t i = 1; i <= P; ++i) offsets.push_back(total + offsets_in[i]);
    idx.clear();
    if (idx_in) {
        idx.assign(idx_in, idx_in + total);
        idx.insert(idx.end(), idx_in, idx_in + total);
    }
}

RegistrationWork *registration_work_create() { return new RegistrationWork; }
WholeVoxelSlot whole_voxel_slot(RegistrationWork &W, bool target, uint32_t n) {
    Side &S = target ? W.M : W.C;
    S.d_ds_soa.ensure(3 * (size_t)n + 4);
    return WholeVoxelSlot{&S.vox_all, S.d_ds_soa.p, &S.vox_all_ready, &S.vox_planes, &S.vox_planes_ready, &S.obb, &S.obb_ready};
}
void registration_work_destroy(RegistrationWork *w) { delete w; }

float source_spacing(plade_ctx *ctx, RegistrationWork &W, const CloudDev &src) {
    StageTimer t(ctx, "t_spacing");
    return average_spacing_dev(ctx, src.aos.p, 6, src.n, src.bbmin, src.bbmax, 6, 10000, W.sp_grid);
}

bool run_registration(plade_ctx *ctx, RegistrationWork &W, const CloudDev &tgt, const CloudDev &src,
                      const PlaneSetView &tp, const PlaneSetView &sp, float *T16_out, const float *spacing_or_null) {
    for (int i = 0; i < 16; ++i) T16_out[i] = (i % 5 == 0) ? 1.f : 0.f;
    const int max_candidates = ctx->params.max_candidates;
    Side &M = W.M, &C = W.C;
    // without planes there are no intersection lines, no descriptors and no candidate: the reference
    // ends in "no matched result found" (plade.cpp:537-540)
    if (tp.P == 0 || sp.P == 0) return false;

    // plade.cpp:41
    const float average_space = spacing_or_null ? *spacing_or_null : source_spacing(ctx, W, src);
    ctx->put1("average_spacing", average_space);
    // plade.cpp:46-56
    const float downSampleDistance = average_space * 4;
    // The verification kernel probes the target's occupancy index once per (candidate, source point); with the source points
    // in a spatially blocked order neighbouring lanes probe neighbouring cells.  That order costs a sort of the downsampled
    // source (4 passes + 5 kernels, ~110 us) and pays from a few hundred candidates on (BASELINE configs[4]: 10^4); for the
    // default <= 201 candidates, of which ~20 survive the penetration filter, the voxel order the downsampled cloud already
    // has serves as well (k_overlap 169 / 198 us with the sort, 156 / 226 us without on the two bench pairs; same counts).
    const bool sort_source = max_candidates > 1000;
    const float lengthThreshold = average_space * 5;
    const float angleThreshold = 5.0 / 180 * M_PI;
    const float cosAngleThreshold = cos(angleThreshold);
    const float scale = lengthThreshold / cos(M_PI_2 - angleThreshold);
    ctx->put1("scale", scale);

    {
        // the two sides are independent until the descriptor match: the source side runs on the
        // auxiliary stream / a second host thread
        StageTimer t(ctx, "t_prepare");
        if (!ctx->aux) {
            plade_ctx *a = nullptr;
            PLADE_REQUIRE(plade_ctx_create(ctx->device, &a) == PLADE_OK, PLADE_EDEVICE, "cannot create the auxiliary stream");
            ctx->aux = a;
        }
        plade_ctx *aux = ctx->aux;
        aux->params = ctx->params;
        aux->dump.clear();
        aux->stats.clear();
        bool ok_c = false, ok_m = false;
        Err aux_err{0, ""}, main_err{0, ""};
        // The source side reads what the plane extraction left on the device (the support lists in the extraction's work
        // area), and the extraction ran on the MAIN stream: its host loop returns as soon as the device reports the end
        // through host-mapped memory, with the last kernels of the iteration possibly still running.  The auxiliary
        // stream therefore waits for the main one here.
        if (!W.ev_main) HIP_TRY(hipEventCreateWithFlags(&W.ev_main, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(W.ev_main, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(aux->stream, W.ev_main, 0));
        // Alone on the GPU the two sides run side by side (the source on the auxiliary stream and a second host thread: a
        // single registration is 0.5 ms shorter); with other registrations in flight that only doubles the streams that share
        // the four hardware queues and adds a host thread per pair -- one side after the other on this stream: +1.5 % of the
        // batch throughput, 0.6 ms less host CPU per registration (profiles/r4_experiments.md 2c).  plade_params.prepare_sides.
        const int ps = ctx->params.prepare_sides;
        // (under a Combiner always: the pair's launches travel on the lead's stream, and an auxiliary stream of the pair's own
        //  could not be ordered against them by events recorded on this context's idle stream; advisor r5)
        const bool serial_sides = ps == 2 || ctx->comb || (ps == 0 && (ctx->in_group || ctx->params.host_wait != 0));
        if (serial_sides) {
            ok_m = prepare_side(ctx, "tgt", tgt, tp, downSampleDistance, pen_grid_cell(lengthThreshold), scale, true, W.tgt_pairs, M, false);
            ok_c = prepare_side(ctx, "src", src, sp, downSampleDistance, pen_grid_cell(lengthThreshold), scale, false, W.src_pairs, C);
            if (!ok_c) ctx->sync();   // the source side gave up before its last wait: the target side's read-back is still out
        }
        std::thread th([&]() {
            if (serial_sides) return;
            const double cpu0 = thread_cpu_seconds();
            struct Acc { Stats &st; double c0; ~Acc() { st.add("cpu_prepare_helper", thread_cpu_seconds() - c0); } } acc{aux->stats, cpu0};
            (void)hipSetDevice(ctx->device);
            try { ok_c = prepare_side(aux, "src", src, sp, downSampleDistance, pen_grid_cell(lengthThreshold), scale, false, W.src_pairs, C); }
            catch (const Err &e) { aux_err = e; }
            catch (const std::exception &e) { aux_err = Err{PLADE_EDEVICE, e.what()}; }
        });
        try { if (!serial_sides) ok_m = prepare_side(ctx, "tgt", tgt, tp, downSampleDistance, pen_grid_cell(lengthThreshold), scale, true, W.tgt_pairs, M); }
        catch (const Err &e) { main_err = e; }
        catch (const std::exception &e) { main_err = Err{PLADE_EDEVICE, e.what()}; }   // the helper thread is still joinable here
        th.join();
        for (auto &kv : aux->dump) ctx->dump[kv.first] = kv.second;
        ctx->stats.merge(aux->stats);
        if (main_err.code) throw main_err;
        if (aux_err.code) throw aux_err;
        if (!ok_m || !ok_c) return false;
        // the verification's target grid (plade.cpp:545-564 builds a kd-tree per candidate) only needs the
        // downsampled target: built now on the idle auxiliary stream, consumed five stages later
        if (!W.ev_grid) HIP_TRY(hipEventCreateWithFlags(&W.ev_grid, hipEventDisableTiming));
        // (in a crowd -- serial_sides -- on this stream too: no second stream per pair at all)
        plade_ctx *gctx = serial_sides ? ctx : aux;
        W.grid.build(gctx, M.d_ds.p, M.n_ds, 3, downSampleDistance, tgt.bbmin, tgt.bbmax, true);
        // ... and the source in a spatially blocked order for the same kernel
        if (sort_source)
        overlap_sort_source(gctx, W.ov_work, C.d_ds_soa.p, C.d_ds_soa.p + C.n_ds, C.d_ds_soa.p + 2 * (size_t)C.n_ds, C.n_ds,
                            1.f / W.grid.gp.inv);
        // (under a Combiner the sides are serial and builder and consumer sit in the same pair queue, in order: no event needed --
        //  and a FUNC entry per pair would hold that pair's queue back while the others' next kernels merge without it)
        if (!gctx->comb) HIP_TRY(hipEventRecord(W.ev_grid, gctx->stream));
    }
    {
        StageTimer t(ctx, "t_descriptors");   // built by prepare_side; only the optional dump is left here
        if (ctx->params.dump) {
            ctx->put_dev("tgt_desc", W.tgt_pairs.desc.p, 8 * (size_t)W.tgt_pairs.count);
            ctx->put_dev("src_desc", W.src_pairs.desc.p, 8 * (size_t)W.src_pairs.count);
        }
    }
    uint64_t n_match;
    {
        StageTimer t(ctx, "t_match");
        n_match = W.match.run(ctx, W.src_pairs.desc.p, W.src_pairs.count, W.tgt_pairs.desc.p, W.tgt_pairs.count, 0.04f);
        if (ctx->params.dump) {
            ctx->put_dev("match_offsets", W.match.offsets.p, (size_t)W.src_pairs.count + 1);
            ctx->put_dev("match_nbr", (const int32_t *)W.match.t_idx.p, n_match);
            ctx->put_dev("match_dist2", W.match.dist2.p, n_match);
        }
    }
    ctx->stats.add("n_descriptors_tgt", W.tgt_pairs.count);
    ctx->stats.add("n_descriptors_src", W.src_pairs.count);
    ctx->stats.add("n_matches", (double)n_match);
    PLADE_REQUIRE(n_match < (1ull << 31), PLADE_ELIMIT, "too many descriptor matches");
    {
        StageTimer t(ctx, "t_transforms");
        build_transforms(ctx, W.src_pairs, W.tgt_pairs, W.match.q_idx_sorted, W.match.t_idx.p, (uint32_t)n_match, W.cand);
        if (ctx->params.dump && n_match) {
            std::vector<float4> h(4 * (size_t)n_match);
            // on the context's (non-blocking) stream: a null-stream copy would not wait for build_transforms
            ctx->d2h(h.data(), W.cand.rt.p, 64 * (size_t)n_match);
            ctx->sync();
            std::vector<float> rt(12 * (size_t)n_match);
            for (size_t i = 0; i < n_match; ++i) {
                for (int r = 0; r < 3; ++r) {
                    rt[12 * i + 3 * r] = h[4 * i + r].x; rt[12 * i + 3 * r + 1] = h[4 * i + r].y; rt[12 * i + 3 * r + 2] = h[4 * i + r].z;
                    rt[12 * i + 9 + r] = h[4 * i + r].w;
                }
            }
            ctx->put("initial_RT", rt.data(), rt.size());
        }
    }
    std::vector<uint32_t> &seeds = W.ch.seeds, &sizes = W.ch.sizes;
    std::vector<int32_t> &pcounts = W.ch.pcounts;
    {
        StageTimer t(ctx, "t_cluster");
        // util.cpp:331: ClusterTransformation(Rs, Ts, lengthThreshold / 2, angleThreshold / 2) with the
        // Parameter fields held as double
        const float distanceThreshold = (float)((double)lengthThreshold / 2);
        const float g_angle = (float)((double)angleThreshold / 2);
        cluster_transforms(ctx, W.cand, distanceThreshold, g_angle);
    }
    const uint32_t nC = W.cand.n_clusters;
    ctx->stats.add("n_clusters", nC);
    // util.cpp:335-401: clusters by size (std::sort with myCompareGreater), centre gate, plane counts.
    std::vector<int> &cand_cluster = W.ch.cand_cluster;   // cluster index per entry of `matches`
    std::vector<int> &match_counts = W.ch.match_counts;
    cand_cluster.clear(); match_counts.clear();
    {
        StageTimer t(ctx, "t_plane_consistency");
        seeds.resize(nC); sizes.resize(nC); pcounts.resize(nC);
        const float sbc[3] = {C.bcenter.x, C.bcenter.y, C.bcenter.z}, tbc[3] = {M.bcenter.x, M.bcenter.y, M.bcenter.z};
        plane_consistency(ctx, W.cand, C.geom, M.geom, sbc, tbc, (float)M.radius, (float)(double)cosAngleThreshold,
                          lengthThreshold);
        if (nC) {   // ONE wait for sizes, seeds and counts (a wait of its own for the sizes, so that the sort could run beside the
                    // 70 us kernel, cost more than it hid)
            ctx->d2h(sizes.data(), W.cand.sizes.p, 4 * (size_t)nC);
            ctx->d2h(seeds.data(), W.cand.seeds.p, 4 * (size_t)nC);
            ctx->d2h(pcounts.data(), W.cand.plane_counts.p, 4 * (size_t)nC);
        }
        ctx->sync();
        const double c_c = thread_cpu_seconds();
        std::vector<exact_sort::Item> &sortVec = W.ch.sort_vec;
        sortVec.resize(nC);
        for (uint32_t i = 0; i < nC; ++i) { sortVec[i].index = (int)i; sortVec[i].length = (float)sizes[i]; }
        // std::sort(sortVec.begin(), sortVec.end(), myCompareGreater) -- its permutation exactly, ties included, at a third
        // of the host time (exact_sort.h)
        exact_sort::sort_descending(sortVec.data(), sortVec.data() + nC);
        const double c_d = thread_cpu_seconds();
        ctx->stats.add("cpu_cluster_order", c_d - c_c);
        cand_cluster.reserve(nC);
        match_counts.reserve(nC);
        for (uint32_t i = 0; i < nC; ++i) {
            const int c = sortVec[i].index;
            if (pcounts[c] < 0) continue;
            cand_cluster.push_back(c);
            match_counts.push_back(pcounts[c]);
        }
    }
    if (ctx->params.dump) {
        std::vector<int32_t> a(sizes.begin(), sizes.end()), b(seeds.begin(), seeds.end());
        ctx->put("cluster_sizes", a.data(), a.size());
        ctx->put("cluster_seeds", b.data(), b.size());
    }
    ctx->put("plane_match_counts", match_counts.data(), match_counts.size());
    // util.cpp:403-445
    size_t maxMatchNum = 0;
    for (int v : match_counts) maxMatchNum = std::max(maxMatchNum, (size_t)v);
    std::vector<std::vector<int>> matchedPlanes;
    if (maxMatchNum > 0) {
        int matchedCount = 0;
        for (size_t i = maxMatchNum; i >= 2; i--) {
            std::vector<int> tmp;
            for (size_t j = 0; j < match_counts.size(); ++j)
                if (i == (size_t)match_counts[j]) { tmp.push_back((int)j); matchedCount++; }
            matchedPlanes.push_back(tmp);
            if (matchedCount >= max_candidates) break;
        }
    }
    // util.cpp:449-459: at most max_candidates + 1 candidates reach the penetration test
    std::vector<int> tested;
    {
        int count = 0;
        bool stop = false;
        for (size_t m = 0; m < matchedPlanes.size() && !stop; ++m)
            for (size_t i = 0; i < matchedPlanes[m].size(); ++i) {
                if (count++ > max_candidates) { stop = true; break; }
                tested.push_back(matchedPlanes[m][i]);
            }
    }
    const uint32_t K = (uint32_t)tested.size();
    ctx->put("pen_tested", tested.data(), tested.size());
    ctx->stats.add("n_candidates_tested", K);
    if (K == 0) return false;
    // fetch the (R, T) of the tested candidates
    std::vector<float> rt12(12 * (size_t)K);
    {
        W.d_rt12.ensure(12 * (size_t)K);
    }
    std::vector<int32_t> penflags;
    {
        StageTimer t(ctx, "t_penetration");
        // the ids of the tested candidates travel with the filter's table upload; the gather of their (R, T) is queued behind it
        std::vector<uint32_t> ids(K);
        for (uint32_t i = 0; i < K; ++i) ids[i] = seeds[cand_cluster[tested[i]]];
        PenGather pg{ids.data(), [&](const uint32_t *d_ids) {
            launch<k_gather_rt, 64>(ctx, dim3(cdiv(K, 64)), 0, W.cand.rt.p, d_ids, K, W.d_rt12.p);
            ctx->d2h(rt12.data(), W.d_rt12.p, 48 * (size_t)K);   // valid after the wait at the end of the penetration filter
        }};
        penetration_filter(ctx, nullptr, K, C.geom, M.geom, C.pcl, M.pcl, lengthThreshold, angleThreshold, penflags, W.d_rt12.p, &pg);
    }
    ctx->put("pen_flags", penflags.data(), penflags.size());
    // survivors = MatchedResult list (util.cpp:513-517)
    std::vector<uint32_t> surv;
    for (uint32_t i = 0; i < K; ++i) if (!penflags[i]) surv.push_back(i);
    const uint32_t Kv = (uint32_t)surv.size();
    ctx->stats.add("n_candidates_verified", Kv);
    if (Kv == 0) return false;  // plade.cpp:537-540
    // plade.cpp:545-575 verification
    std::vector<float> T16(16 * (size_t)Kv), centers(3 * (size_t)Kv);
    for (uint32_t i = 0; i < Kv; ++i) {
        const float *r = &rt12[12 * (size_t)surv[i]];
        float *T = &T16[16 * (size_t)i];
        m3 R;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { R.m[a][b] = r[3 * a + b]; T[4 * a + b] = r[3 * a + b]; }
        T[3] = r[9]; T[7] = r[10]; T[11] = r[11];
        T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
        const f3 cc = mul_e(R, C.bcenter) + f3(r[9], r[10], r[11]);  // plade.cpp:555
        centers[3 * i] = cc.x; centers[3 * i + 1] = cc.y; centers[3 * i + 2] = cc.z;
    }
    std::vector<int32_t> counts(Kv);
    {
        StageTimer t(ctx, "t_verify");
        // one upload (transforms | centres) and one read-back (counts | sphere flags): every command of a stream costs ~6 us
        // of queue time and ~10 us of host time, whatever its size.  With a candidate shard set (plade_set_candidate_shard)
        // only this rank's candidates k % world == rank are scored here and the counts of the others arrive through the
        // caller's exchange.
        const plade_ctx::CandidateShard &sh = ctx->shard;
        const bool sharded = ((sh.world > 1 && sh.exchange) || sh.comm) && Kv >= sh.min_candidates;   // a communicator of ONE rank still exchanges (the one-GPU test of this path)
        const bool by_rccl = sharded && sh.comm;
        std::vector<uint32_t> mine;
        for (uint32_t i = 0; i < Kv; ++i) if (!sharded || i % sh.world == sh.rank) mine.push_back(i);
        const uint32_t Km = (uint32_t)mine.size();
        // slots per rank: the RCCL form all-gathers equally sized blocks (counts | sphere flags of ceil(Kv / world) candidates)
        const uint32_t per = by_rccl ? cdiv(Kv, sh.world) : Km;
        // transforms | centres | counts | sphere flags in one block: the zeros of the last two travel with the upload (a memset of
        // a few hundred bytes is one or two fill commands of its own)
        W.d_T16.ensure(19 * (size_t)Km + 2 * (size_t)per + 4);
        float *d_centers = W.d_T16.p + 16 * (size_t)Km;
        int32_t *d_counts = reinterpret_cast<int32_t *>(W.d_T16.p + 19 * (size_t)Km);
        uint32_t *d_any = reinterpret_cast<uint32_t *>(d_counts) + per;
        std::vector<float> up(19 * (size_t)Km + 2 * (size_t)per, 0.f);
        for (uint32_t q = 0; q < Km; ++q) {
            memcpy(up.data() + 16 * (size_t)q, T16.data() + 16 * (size_t)mine[q], 64);
            memcpy(up.data() + 16 * (size_t)Km + 3 * (size_t)q, centers.data() + 3 * (size_t)mine[q], 12);
        }
        std::vector<int32_t> back(2 * (size_t)Kv, 0);
        if (Km || by_rccl) {
            ctx->h2d(W.d_T16.p, up.data(), 4 * up.size());
            if (Km) {
                if (!ctx->comb) HIP_TRY(hipStreamWaitEvent(ctx->stream, W.ev_grid, 0));   // (under a Combiner: same queue, see the record)
                const float *vsx = sort_source ? W.ov_work.sorted.p : C.d_ds_soa.p;
                overlap_counts(ctx, W.ov_work, vsx, vsx + C.n_ds, vsx + 2 * (size_t)C.n_ds, C.n_ds,
                               W.grid, W.d_T16.p, d_centers, Km, (float)C.radius, downSampleDistance, d_counts, d_any, true);
            }
            if (by_rccl) {
                // ONE ncclAllGather on this stream, behind the kernel that wrote the counts: they never visit the host before
                // every rank holds all of them (rank r's slot q is candidate q * world + r)
                int32_t *d_all = W.d_shard_all.ensure(2 * (size_t)per * sh.world + 4);
                // (through raw_launch: when this pair's launches are collected by a Combiner -- a one-pair part of a group on a peer
                //  context -- the upload and the kernels above reach the lead's stream only at the next flush, and the collective
                //  has to take its place BEHIND them in the same order, on that stream; advisor r5)
                {
                    plade_comm *comm = sh.comm;
                    const size_t bytes = 8 * (size_t)per;
                    ctx->raw_launch([comm, d_counts, d_all, bytes](hipStream_t st) { comm_all_gather_dev(comm, d_counts, d_all, bytes, st); });
                }
                std::vector<int32_t> all(2 * (size_t)per * sh.world);
                ctx->d2h(all.data(), d_all, 4 * all.size());
                ctx->sync();
                for (uint32_t r = 0; r < sh.world; ++r)
                    for (uint32_t q = 0; q < per; ++q) {
                        const uint32_t i = q * sh.world + r;
                        if (i < Kv) { back[i] = all[2 * (size_t)per * r + q]; back[Kv + i] = all[2 * (size_t)per * r + per + q]; }
                    }
            } else {
                std::vector<int32_t> part(2 * (size_t)Km);
                ctx->d2h(part.data(), d_counts, 8 * (size_t)Km);
                ctx->sync();
                for (uint32_t q = 0; q < Km; ++q) { back[mine[q]] = part[q]; back[Kv + mine[q]] = part[Km + q]; }
            }
        }
        if (sharded) ctx->stats.add("n_candidates_scored_here", Km);
        if (sharded && !by_rccl)
            PLADE_REQUIRE(sh.exchange(sh.user, back.data(), 2 * Kv, sh.rank, sh.world) == 0, PLADE_EDEVICE, "candidate shard: the exchange failed");
        for (uint32_t i = 0; i < Kv; ++i) counts[i] = back[Kv + i] ? back[i] : -1;
    }
    std::vector<LengthIndex> ov(Kv);
    std::vector<float> scores(Kv);
    const size_t denom = std::min<size_t>(C.n_ds, M.n_ds);
    for (uint32_t i = 0; i < Kv; ++i) {
        float ratio = 0.f;
        if (counts[i] >= 0) ratio = (float)(double(counts[i]) / denom);                     // util.h:644
        const int matched = match_counts[tested[surv[i]]];
        ov[i].index = (int)i;
        ov[i].length = (float)(0.2 * (matched / double(sp.P)) + 0.8 * ratio);               // plade.cpp:561-562
        scores[i] = ov[i].length;
    }
    ctx->put("candidates", T16.data(), T16.size());
    ctx->put("candidate_centers", centers.data(), centers.size());
    ctx->put("overlap_counts", counts.data(), counts.size());
    ctx->put("scores", scores.data(), scores.size());
    std::sort(ov.begin(), ov.end(), cmp_greater);                                             // plade.cpp:565
    const int best = ov[0].index;
    ctx->put1("best_index", (int32_t)best);
    memcpy(T16_out, &T16[16 * (size_t)best], 64);
    // roofline bookkeeping: algorithmic bytes of the verify stage
    ctx->stats.add("bytes_verify", (double)Kv * C.n_ds * 12.0 + (double)M.n_ds * 12.0);
    return true;
}

}  // namespace plade

using namespace plade;

static void validate_planes(const float *coef, const int32_t *off, const int32_t *idx, uint32_t P, uint32_t n, const char *who) {
    PLADE_REQUIRE(coef && off && (idx || off[P] == 0), PLADE_EINVAL, std::string(who) + ": null plane arrays");
    PLADE_REQUIRE(off[0] == 0, PLADE_EINVAL, std::string(who) + ": offsets[0] must be 0");
    for (uint32_t i = 0; i < P; ++i) PLADE_REQUIRE(off[i + 1] >= off[i], PLADE_EINVAL, std::string(who) + ": offsets must be non-decreasing");
    for (int32_t i = 0; i < off[P]; ++i) PLADE_REQUIRE(idx[i] >= 0 && (uint32_t)idx[i] < n, PLADE_EINVAL, std::string(who) + ": point index out of range");
}

// ---- C ABI: plade.h:74-79 ---------------------------------------------------------------------
extern "C" int plade_registration_planes(plade_ctx *ctx, const float *tgt_pos_nrm, uint32_t n_t, const float *src_pos_nrm,
                                         uint32_t n_s, const float *tgt_planes, const int32_t *tgt_offsets,
                                         const int32_t *tgt_idx, uint32_t p_t, const float *src_planes,
                                         const int32_t *src_offsets, const int32_t *src_idx, uint32_t p_s, float *T16) {
    return guarded(ctx, [&]() -> int {
        PLADE_REQUIRE(tgt_pos_nrm && src_pos_nrm && T16 && n_t && n_s, PLADE_EINVAL, "plade_registration_planes: bad cloud arguments");
        validate_planes(tgt_planes, tgt_offsets, tgt_idx, p_t, n_t, "target planes");
        validate_planes(src_planes, src_offsets, src_idx, p_s, n_s, "source planes");
        ctx->stats.clear();
        ctx->dump.clear();
        if (!ctx->reg_work) ctx->reg_work = registration_work_create();
        cloud_drop_prefetch(ctx);
        CloudDev &tgt = ctx->up_tgt, &src = ctx->up_src;
        {
            StageTimer t(ctx, "t_upload");
            cloud_upload_pair(ctx, tgt_pos_nrm, n_t, tgt, src_pos_nrm, n_s, src);
        }
        PlaneSetView tp, sp;
        tp.coef = tgt_planes; tp.offsets = tgt_offsets; tp.idx = tgt_idx; tp.P = p_t;
        sp.coef = src_planes; sp.offsets = src_offsets; sp.idx = src_idx; sp.P = p_s;
        MirroredPlanes mirror;
        if (ctx->params.unoriented_normals) {   // README.md:109-110, see MirroredPlanes
            mirror.build(tgt_planes, tgt_offsets, tgt_idx, p_t);
            tp.coef = mirror.coef.data(); tp.offsets = mirror.offsets.data(); tp.idx = mirror.idx.data(); tp.P = 2 * p_t;
            tp.mirrored = true;
        }
        Clock::time_point t0 = Clock::now();
        const bool ok = run_registration(ctx, *ctx->reg_work, tgt, src, tp, sp, T16, nullptr);
        ctx->stats.add("t_registration", secs_since(t0));
        ctx->ev_collect();
        if (!ok) { ctx->last_error = "registration failed: no matched result found"; return PLADE_EFAIL; }
        return PLADE_OK;
    });
}
