// plade_amd/csrc/plade_host.cpp -- the four registration() overloads of code/PLADE/plade.h (and
// PlaneExtraction::detect, load_ply_cloud) on top of the C ABI of libplade_hip.so.  Console messages,
// return values and the swap/invert rule follow code/PLADE/plade.cpp:583-706.
#include "plade.h"
#include "plade_hip.h"
#include "ply_reader.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <mutex>
#include <thread>

namespace {

thread_local int g_device = 0;
thread_local plade_ctx *g_ctx = nullptr;
thread_local int g_ctx_device = -1;
// console of the calling thread: std::cout / std::cerr unless the caller collects the messages itself (the CLI's batch
// workers do, so that concurrent registrations do not interleave their lines)
thread_local std::ostream *g_out = nullptr, *g_err = nullptr;
std::ostream &con_out() { return g_out ? *g_out : std::cout; }
std::ostream &con_err() { return g_err ? *g_err : std::cerr; }

// PLADE_TRACE_CLI=1: what the process spends where, on std::cerr (seconds since the first call)
double trace_now() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
bool trace_on() { static const bool v = getenv("PLADE_TRACE_CLI") != nullptr; return v; }
void trace(const char *what) {
    if (!trace_on()) return;
    char b[160];
    snprintf(b, sizeof(b), "[plade %8.3f s] %s\n", trace_now(), what);
    std::cerr << b << std::flush;
}

plade_ctx *context() {
    if (g_ctx && g_ctx_device == g_device) return g_ctx;
    if (g_ctx) { plade_ctx_destroy(g_ctx); g_ctx = nullptr; }
    trace("context: creating");
    if (plade_ctx_create(g_device, &g_ctx) != PLADE_OK) {
        con_err() << "PLADE: cannot create a GPU context on device " << g_device
                  << " (libplade_hip.so needs a gfx950 GPU; there is no CPU fallback)" << std::endl;
        return nullptr;
    }
    g_ctx_device = g_device;
    // opt-in switches of the C++ API / CLI, which have no parameter to carry them (the C ABI's defaults are the
    // reference's values and do not look at the environment): see include/plade_hip.h, plade_params
    plade_params prm;
    plade_default_params(&prm);
    bool changed = false;
    if (const char *w = getenv("PLADE_HOST_WAIT")) { prm.host_wait = (w[0] == 's' && w[1] == 'l') ? 1 : 0; changed = true; }
    if (const char *w = getenv("PLADE_ORIENT_NORMALS")) { prm.orient_normals = atoi(w) != 0; changed = true; }
    if (const char *w = getenv("PLADE_UNORIENTED_NORMALS")) { prm.unoriented_normals = atoi(w) != 0; changed = true; }
    if (const char *w = getenv("PLADE_RANSAC_TOPUP")) { prm.ransac_topup = atoi(w) != 0; changed = true; }
    if (const char *w = getenv("PLADE_CLOSEST_POINT_MODE")) { prm.closest_point_mode = (!strcmp(w, "closed_form") || !strcmp(w, "0")) ? 0 : 1; changed = true; }
    if (changed) (void)plade_set_params(g_ctx, &prm);
    trace("context: ready");
    return g_ctx;
}

// The staging arrays of a worker thread, page-locked: the library then uploads them by asynchronous DMA (~50 GB/s) instead of
// through the runtime's bounce buffer (pageable memory: ~6-10 GB/s, and the call blocks meanwhile) -- at 48 MB per pair the
// difference is most of a long list's wall time.  An array is registered once and again only when its allocation has moved.
struct PinnedVec {
    const float *ptr = nullptr; size_t bytes = 0; plade_ctx *owner = nullptr;
    void release() { if (ptr && owner && owner == g_ctx) (void)plade_host_unpin(owner, ptr); ptr = nullptr; bytes = 0; owner = nullptr; }
    void cover(plade_ctx *ctx, const std::vector<float> &v) {
        const size_t want = v.capacity() * sizeof(float);
        if (ptr == v.data() && bytes == want && owner == ctx) return;
        release();
        if (!ctx || !want) return;
        if (plade_host_pin(ctx, v.data(), want) == PLADE_OK) { ptr = v.data(); bytes = want; owner = ctx; }
    }
};
thread_local PinnedVec g_pins[2 * PLADE_GROUP_MAX];

std::vector<float> flatten(const pcl::PointCloud<pcl::PointNormal> &c) {
    std::vector<float> a(6 * c.size());
    for (size_t i = 0; i < c.size(); ++i) {
        const pcl::PointNormal &p = c.points[i];
        float *o = &a[6 * i];
        o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = p.normal_x; o[4] = p.normal_y; o[5] = p.normal_z;
    }
    return a;
}

void to_matrix(const float *T16, Eigen::Matrix<float, 4, 4> &m) {
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m(r, c) = T16[4 * r + c];
}

void pack_planes(const std::vector<PLANE> &planes, std::vector<float> &coef, std::vector<int32_t> &off, std::vector<int32_t> &idx) {
    coef.clear(); off.assign(1, 0); idx.clear();
    for (const PLANE &p : planes) {
        coef.push_back(p.normal.x()); coef.push_back(p.normal.y()); coef.push_back(p.normal.z()); coef.push_back(p.d);
        idx.insert(idx.end(), p.begin(), p.end());
        off.push_back((int32_t)idx.size());
    }
}

struct Watch {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::string str() const {
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        char buf[64];
        snprintf(buf, sizeof(buf), "%.3f sec", s);
        return buf;
    }
};

std::string extension(const std::string &file_name) {  // util.cpp:525-531
    std::string::size_type dot = file_name.find_last_of('.');
    std::string::size_type slash = file_name.find_last_of("/\\");
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return std::string("");
    return std::string(file_name.begin() + dot + 1, file_name.end());
}

}  // namespace

void plade_cli_trace(const char *what) { trace(what); }
void plade_select_device(int device) { g_device = device; }
int plade_gpu_count() { return plade_device_count(); }

void plade_set_thread_console(std::ostream *out, std::ostream *err) { g_out = out; g_err = err; }

void plade_release_thread_context() {
    for (PinnedVec &p : g_pins) p.release();     // before the arrays themselves go with the thread
    if (g_ctx) { plade_ctx_destroy(g_ctx); g_ctx = nullptr; g_ctx_device = -1; }
}

std::vector<PLANE> PlaneExtraction::detect(const pcl::PointCloud<pcl::PointNormal> &cloud, unsigned int min_support,
                                           float dist_thresh, float bitmap_reso, float normal_thresh, float overlook_prob) {
    std::vector<PLANE> out;
    if (cloud.size() < 3) {  // plane_extraction.cpp:181-184
        con_err() << "point set has less than 3 points" << std::endl;
        return out;
    }
    plade_ctx *ctx = context();
    if (!ctx) return out;
    std::vector<float> a = flatten(cloud);
    const uint32_t max_planes = 4096;
    std::vector<float> coef(4 * max_planes);
    std::vector<int32_t> off(max_planes + 1), idx(cloud.size());
    uint32_t P = 0;
    int rc = plade_extract_planes(ctx, a.data(), (uint32_t)cloud.size(), min_support, dist_thresh, bitmap_reso, normal_thresh,
                                  overlook_prob, coef.data(), off.data(), idx.data(), max_planes, &P);
    if (rc != PLADE_OK) { con_err() << "plane extraction failed: " << plade_last_error(ctx) << std::endl; return out; }
    for (uint32_t i = 0; i < P; ++i) {
        PLANE pl(idx.begin() + off[i], idx.begin() + off[i + 1]);
        pl.normal = Eigen::Vector3f(coef[4 * i], coef[4 * i + 1], coef[4 * i + 2]);
        pl.d = coef[4 * i + 3];
        out.push_back(pl);
    }
    return out;
}

// plade.h:74-79 / plade.cpp:31-580
bool registration(Eigen::Matrix<float, 4, 4> &transformation, pcl::PointCloud<pcl::PointNormal>::Ptr target_cloud,
                  pcl::PointCloud<pcl::PointNormal>::Ptr source_cloud, const std::vector<PLANE> &target_planes,
                  const std::vector<PLANE> &source_planes) {
    con_out() << "#point in target point cloud: " << target_cloud->size() << std::endl;
    con_out() << "#point in source point cloud: " << source_cloud->size() << std::endl;
    con_out() << "#planes in target point cloud: " << target_planes.size() << std::endl;
    con_out() << "#planes in source point cloud: " << source_planes.size() << std::endl;
    plade_ctx *ctx = context();
    if (!ctx) return false;
    std::vector<float> tg = flatten(*target_cloud), sr = flatten(*source_cloud), tc, sc;
    std::vector<int32_t> to, ti, so, si;
    pack_planes(target_planes, tc, to, ti);
    pack_planes(source_planes, sc, so, si);
    float T16[16];
    Watch w;
    con_out() << "registration..." << std::endl;
    int rc = plade_registration_planes(ctx, tg.data(), (uint32_t)target_cloud->size(), sr.data(), (uint32_t)source_cloud->size(),
                                       tc.data(), to.data(), ti.data(), (uint32_t)target_planes.size(), sc.data(), so.data(),
                                       si.data(), (uint32_t)source_planes.size(), T16);
    if (rc != PLADE_OK) {
        con_err() << (rc == PLADE_EFAIL ? "registration failed: no matched result found" : plade_last_error(ctx)) << std::endl;
        return false;
    }
    to_matrix(T16, transformation);
    con_out() << "done. time: " << w.str() << std::endl;
    return true;
}

// plade.h:91-96 / plade.cpp:583-599
bool registration(Eigen::Matrix<float, 4, 4> &transformation, pcl::PointCloud<pcl::PointNormal>::Ptr target_cloud,
                  pcl::PointCloud<pcl::PointNormal>::Ptr source_cloud, int ransac_min_support_target,
                  int ransac_min_support_source) {
    plade_ctx *ctx = context();
    if (!ctx) return false;
    std::vector<float> tg = flatten(*target_cloud), sr = flatten(*source_cloud);
    float T16[16];
    Watch w;
    con_out() << "extracting planes and registering...\n";
    int rc = plade_registration_minsupport(ctx, tg.data(), (uint32_t)target_cloud->size(), sr.data(), (uint32_t)source_cloud->size(),
                                           ransac_min_support_target, ransac_min_support_source, T16);
    if (rc != PLADE_OK) {
        con_err() << (rc == PLADE_EFAIL ? plade_last_error(ctx) : plade_last_error(ctx)) << std::endl;
        return false;
    }
    to_matrix(T16, transformation);
    con_out() << "done. time: " << w.str() << std::endl;
    return true;
}

namespace {

// body of registration(T, target, source) (plade.cpp:638-662) on packed x y z nx ny nz arrays
bool register_packed(Eigen::Matrix<float, 4, 4> &transformation, const float *tg, size_t n_t, const float *sr, size_t n_s) {
    plade_ctx *ctx = context();
    if (!ctx) return false;
    con_out() << "extracting planes for both point clouds...\n";
    float T16[16];
    Watch w;
    int rc = plade_registration(ctx, tg, (uint32_t)n_t, sr, (uint32_t)n_s, T16);
    if (rc != PLADE_OK) {
        con_err() << plade_last_error(ctx) << std::endl;
        return false;
    }
    to_matrix(T16, transformation);
    con_out() << "done. time: " << w.str() << std::endl;
    return true;
}

// a PLY file straight into a packed array (no pcl::PointCloud in between: at GPU speeds the 48 B/point
// intermediate and its page faults cost several registrations); `buf` is reused from call to call
bool load_packed(const std::string &file_name, std::vector<float> &buf) {
    std::string err;
    std::vector<std::string> warnings;
    if (!plade::read_ply_pos_nrm(file_name, buf, err, &warnings)) {
        if (!err.empty()) con_err() << err << std::endl;
        return false;
    }
    for (auto &w : warnings) con_out() << w << std::endl;
    return !buf.empty();
}

}  // namespace

// plade.h:58-61 / plade.cpp:638-662
bool registration(Eigen::Matrix<float, 4, 4> &transformation, pcl::PointCloud<pcl::PointNormal>::Ptr target_cloud,
                  pcl::PointCloud<pcl::PointNormal>::Ptr source_cloud) {
    std::vector<float> tg = flatten(*target_cloud), sr = flatten(*source_cloud);
    return register_packed(transformation, tg.data(), target_cloud->size(), sr.data(), source_cloud->size());
}

// plade.h:44-47 / plade.cpp:665-706
bool registration(Eigen::Matrix<float, 4, 4> &transformation, const std::string &target_cloud_file,
                  const std::string &source_cloud_file) {
    con_out() << "target file: " << target_cloud_file << std::endl;
    con_out() << "source file: " << source_cloud_file << std::endl;
    if (extension(target_cloud_file) != "ply" || extension(source_cloud_file) != "ply") {
        con_err() << "only PLY format is accepted" << std::endl;
        return false;
    }
    thread_local std::vector<float> target_buf, source_buf;   // one pair of staging arrays per worker thread
    trace("pair: reading the two files");
    if (!load_packed(target_cloud_file, target_buf)) {
        con_err() << "loading target point cloud failed" << std::endl;
        return false;
    }
    if (!load_packed(source_cloud_file, source_buf)) {
        con_err() << "loading source point cloud failed" << std::endl;
        return false;
    }
    const float *tg = target_buf.data(), *sr = source_buf.data();
    size_t n_t = target_buf.size() / 6, n_s = source_buf.size() / 6;
    bool switched = false;
    if (n_s >= n_t * 1.2f) {
        std::swap(tg, sr);
        std::swap(n_t, n_s);
        switched = true;
        con_out() << "---->>> ATTENTION: target and source have been switched for efficiency <<<----" << std::endl;
    }
    transformation.setIdentity();
    trace("pair: files read");
    bool status = register_packed(transformation, tg, n_t, sr, n_s);
    trace("pair: registered");
    if (!status) {
        con_err() << "registration failed" << std::endl;
        return false;
    }
    if (switched) transformation = transformation.inverse();
    return true;
}

// Batch extension: see plade.h.  The per-pair part of the file overload above (messages, extension check, loading, the
// target/source switch, plade.cpp:665-706) runs pair by pair; the pairs that got that far are registered as one group.
void registration_group(size_t count, Eigen::Matrix<float, 4, 4> *transformations, const std::string *target_cloud_files,
                        const std::string *source_cloud_files, bool *ok, std::ostream *const *out, std::ostream *const *err) {
    constexpr size_t GMAX = PLADE_GROUP_MAX;
    static_assert(GMAX == registration_group_max, "plade.h states the group limit of include/plade_hip.h");
    if (count > GMAX) count = GMAX;
    thread_local std::vector<float> bufs[2 * GMAX];   // staging arrays of this worker thread, reused from group to group
    struct Item { size_t pair; const float *tg, *sr; size_t n_t, n_s; bool switched; };
    std::vector<Item> items;
    // The files of the group are read side by side (a 1M-point binary PLY is ~10 ms of parsing and copying; sixteen of them one
    // after the other would take longer than the group's registration); what the loads have to say is kept and printed below,
    // pair by pair, where the sequential run prints it.
    struct Loaded { bool tried = false, ok = false; std::string err; std::vector<std::string> warnings; };
    Loaded loaded[2 * GMAX];
    {
        std::vector<float> *const b = bufs;
        std::vector<std::thread> readers;
        for (size_t i = 0; i < count; ++i) {
            if (extension(target_cloud_files[i]) != "ply" || extension(source_cloud_files[i]) != "ply") continue;
            for (int side = 0; side < 2; ++side) {
                const std::string *file = side ? &source_cloud_files[i] : &target_cloud_files[i];
                Loaded *l = &loaded[2 * i + side];
                std::vector<float> *buf = &b[2 * i + side];
                l->tried = true;
                // (an array that has to grow moves: its page-locked registration is released BEFORE the old block is freed -- a
                //  registration left on freed memory fails the next one of whatever block takes the address; advisor r5)
                PinnedVec *pin = &g_pins[2 * i + side];
                readers.emplace_back([file, l, buf, pin]() {
                    const std::function<void()> before_grow = [pin]() { pin->release(); };
                    l->ok = plade::read_ply_pos_nrm(*file, *buf, l->err, &l->warnings, &before_grow) && !buf->empty();
                });
            }
        }
        // the worker's GPU context (HIP start-up on first use, streams, the first work areas) is set up while the files load
        trace("group: reading");
        (void)context();
        for (auto &t : readers) t.join();
        trace("group: files read");
    }
    auto report = [&](const Loaded &l) {   // load_packed's messages
        if (!l.ok && !l.err.empty()) con_err() << l.err << std::endl;
        if (l.ok || l.err.empty()) for (auto &w : l.warnings) con_out() << w << std::endl;
        return l.ok;
    };
    for (size_t i = 0; i < count; ++i) {
        ok[i] = false;
        transformations[i].setIdentity();
        plade_set_thread_console(out ? out[i] : nullptr, err ? err[i] : nullptr);
        con_out() << "target file: " << target_cloud_files[i] << std::endl;
        con_out() << "source file: " << source_cloud_files[i] << std::endl;
        if (extension(target_cloud_files[i]) != "ply" || extension(source_cloud_files[i]) != "ply") {
            con_err() << "only PLY format is accepted" << std::endl;
            continue;
        }
        if (!report(loaded[2 * i])) { con_err() << "loading target point cloud failed" << std::endl; continue; }
        if (!report(loaded[2 * i + 1])) { con_err() << "loading source point cloud failed" << std::endl; continue; }
        Item it{i, bufs[2 * i].data(), bufs[2 * i + 1].data(), bufs[2 * i].size() / 6, bufs[2 * i + 1].size() / 6, false};
        if (it.n_s >= it.n_t * 1.2f) {
            std::swap(it.tg, it.sr);
            std::swap(it.n_t, it.n_s);
            it.switched = true;
            con_out() << "---->>> ATTENTION: target and source have been switched for efficiency <<<----" << std::endl;
        }
        con_out() << "extracting planes for both point clouds...\n";
        items.push_back(it);
    }
    plade_set_thread_console(nullptr, nullptr);
    if (items.empty()) return;
    plade_ctx *ctx = context();
    for (size_t q = 0; q < 2 * GMAX; ++q) if (!bufs[q].empty()) g_pins[q].cover(ctx, bufs[q]);
    trace("group: staging arrays page-locked");
    const uint32_t k = (uint32_t)items.size();
    const float *tg[GMAX], *sr[GMAX];
    uint32_t n_t[GMAX], n_s[GMAX];
    float T16[16 * GMAX];
    int32_t status[GMAX];
    for (uint32_t q = 0; q < k; ++q) { tg[q] = items[q].tg; sr[q] = items[q].sr; n_t[q] = (uint32_t)items[q].n_t; n_s[q] = (uint32_t)items[q].n_s; status[q] = PLADE_EDEVICE; }
    Watch w;
    int rc = ctx ? plade_registration_pairs(ctx, k, tg, n_t, sr, n_s, 0, nullptr, nullptr, nullptr, nullptr, T16, status) : PLADE_EDEVICE;
    trace("group: registered");
    for (uint32_t q = 0; q < k; ++q) {
        const Item &it = items[q];
        plade_set_thread_console(out ? out[it.pair] : nullptr, err ? err[it.pair] : nullptr);
        if (rc != PLADE_OK || status[q] != PLADE_OK) {
            const plade_ctx *pc = ctx ? (rc != PLADE_OK ? ctx : plade_pair_ctx(ctx, q)) : nullptr;
            if (pc) con_err() << plade_last_error(pc) << std::endl;
            con_err() << "registration failed" << std::endl;
            continue;
        }
        to_matrix(T16 + 16 * q, transformations[it.pair]);
        con_out() << "done. time: " << w.str() << std::endl;
        if (it.switched) transformations[it.pair] = transformations[it.pair].inverse();
        ok[it.pair] = true;
    }
    plade_set_thread_console(nullptr, nullptr);
}

bool load_ply_cloud(const std::string &file_name, pcl::PointCloud<pcl::PointNormal> &cloud) {
    std::vector<float> pos_nrm;
    std::string err;
    std::vector<std::string> warnings;
    if (!plade::read_ply_pos_nrm(file_name, pos_nrm, err, &warnings)) {
        if (!err.empty()) con_err() << err << std::endl;
        return false;
    }
    for (auto &w : warnings) con_out() << w << std::endl;
    const size_t n = pos_nrm.size() / 6;
    cloud.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const float *p = &pos_nrm[6 * i];
        cloud.at(i) = pcl::PointNormal(p[0], p[1], p[2], p[3], p[4], p[5]);
    }
    return cloud.size() > 0;
}
