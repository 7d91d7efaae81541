"""plade_amd -- MI355X (gfx950) implementation of PLADE's registration hot path.

This package is a thin ctypes binding over the C ABI of ``libplade_hip.so`` (declared in
``include/plade_hip.h``).  There is no CPU fallback: importing the package is cheap, but creating
a :class:`Context` fails loudly when the HIP library or a GPU is missing.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libplade_hip.so")

PLADE_OK = 0
PLADE_EINVAL, PLADE_EDEVICE, PLADE_ECAP, PLADE_EFAIL, PLADE_ELIMIT = -1, -2, -3, -4, -5

# every symbol include/plade_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "plade_ctx_create", "plade_ctx_destroy", "plade_last_error", "plade_version", "plade_default_params",
    "plade_set_params", "plade_score_planes", "plade_extract_planes", "plade_match_descriptors",
    "plade_overlap_counts", "plade_average_spacing", "plade_voxel_downsample", "plade_registration_planes",
    "plade_registration", "plade_registration_minsupport", "plade_cloud_upload", "plade_cloud_free",
    "plade_registration_dev", "plade_dump_get", "plade_stats_get", "plade_kernel_time", "plade_plane_component",
    "plade_sort_pairs", "plade_host_pin", "plade_host_unpin", "plade_score_planes_subset", "plade_registration_next", "plade_cluster_transforms", "plade_device_synchronize", "plade_selftest_readback",
    "plade_registration_pairs", "plade_registration_pairs_dev", "plade_pair_ctx", "plade_set_candidate_shard", "plade_diag_launches", "plade_diag_cluster_order", "plade_diag_line_solver_host", "plade_sort_segments",
    "plade_closest_points", "plade_lines_meet", "plade_ply_read", "plade_ply_free",
    "plade_device_count", "plade_comm_unique_id", "plade_comm_create", "plade_comm_all_gather", "plade_comm_destroy", "plade_comm_last_error",
    "plade_set_candidate_shard_comm",
]


class PladeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libplade_hip error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("max_planes", C.c_int32), ("min_planes", C.c_int32), ("max_candidates", C.c_int32),
                ("init_min_support", C.c_int32), ("orient_normals", C.c_int32), ("dump", C.c_int32),
                ("ransac_seed", C.c_uint64), ("host_wait", C.c_int32), ("unoriented_normals", C.c_int32),
                ("ransac_topup", C.c_int32), ("match_window", C.c_int32), ("match_cell_budget", C.c_uint32),
                ("group_max_points", C.c_uint32), ("prepare_sides", C.c_int32), ("closest_point_mode", C.c_int32)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.c_uint32)

_lib = None


def load_library(path=LIB_PATH):
    """Load libplade_hip.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: build it with `make lib` or __graft_entry__.build(); "
                                "plade_amd has no CPU fallback")
    L = C.CDLL(path)
    p, f, i32, u32, u64 = C.c_void_p, C.c_float, C.c_int32, C.c_uint32, C.c_uint64
    def sig(name, argtypes=None, restype=None):
        # a symbol missing from the library is reported by Context/_need, never silently replaced
        fn = getattr(L, name, None)
        if fn is None:
            return
        if argtypes is not None:
            fn.argtypes = argtypes
        if restype is not None:
            fn.restype = restype

    sig("plade_version", restype=C.c_char_p)
    sig("plade_last_error", restype=C.c_char_p)
    sig("plade_last_error", argtypes=[p])
    sig("plade_device_synchronize", argtypes=[C.c_int])
    sig("plade_ctx_create", argtypes=[C.c_int, C.POINTER(p)])
    sig("plade_ctx_destroy", argtypes=[p])
    sig("plade_default_params", argtypes=[C.POINTER(Params)])
    sig("plade_set_params", argtypes=[p, C.POINTER(Params)])
    sig("plade_score_planes", argtypes=[p, p, p, u32, p, u32, f, f, p, p, u32])
    sig("plade_score_planes_subset", argtypes=[p, p, p, u32, p, u32, p, u32, f, f, p, p])
    sig("plade_extract_planes", argtypes=[p, p, u32, u32, f, f, f, f, p, p, p, u32, p])
    sig("plade_match_descriptors", argtypes=[p, p, u32, p, u32, f, p, p, p, u64, p])
    sig("plade_cluster_transforms", argtypes=[p, p, p, u32, f, f, p, p])
    sig("plade_overlap_counts", argtypes=[p, p, u32, p, u32, p, u32, p, f, f, p])
    sig("plade_average_spacing", argtypes=[p, p, u32, u32, u32, u32, p])
    sig("plade_voxel_downsample", argtypes=[p, p, u32, u32, f, p, p])
    sig("plade_registration_planes", argtypes=[p, p, u32, p, u32, p, p, p, u32, p, p, p, u32, p])
    sig("plade_registration", argtypes=[p, p, u32, p, u32, p])
    sig("plade_registration_next", argtypes=[p, p, u32, p, u32, p, u32, p, u32, p])
    sig("plade_registration_pairs", argtypes=[p, u32, p, p, p, p, u32, p, p, p, p, p, p])
    sig("plade_registration_pairs_dev", argtypes=[p, u32, p, p, p, p])
    sig("plade_pair_ctx", argtypes=[p, u32], restype=p)
    sig("plade_closest_points", argtypes=[p, i32, p, p, p, p, u32, p, p, p, p])
    sig("plade_lines_meet", argtypes=[p, i32, p, p, p, p, u32, p, p])
    sig("plade_diag_launches", argtypes=[p, u32, u32, u32])
    sig("plade_diag_cluster_order", argtypes=[p, u32, i32, i32, p])
    sig("plade_diag_line_solver_host", argtypes=[i32, p, p, p, p, u32, p, p, p])
    sig("plade_ply_read", argtypes=[C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(u64), C.c_char_p, C.c_size_t])
    sig("plade_ply_free", argtypes=[C.POINTER(C.c_float)], restype=None)
    sig("plade_sort_segments", argtypes=[p, p, p, p, u32, C.c_int, p, p])
    sig("plade_set_candidate_shard", argtypes=[p, u32, u32, u32, EXCHANGE_FN, p])
    sig("plade_registration_minsupport", argtypes=[p, p, u32, p, u32, i32, i32, p])
    sig("plade_cloud_upload", argtypes=[p, p, u32, C.POINTER(p)])
    sig("plade_cloud_free", argtypes=[p, p])
    sig("plade_registration_dev", argtypes=[p, p, p, p])
    sig("plade_dump_get", argtypes=[p, C.c_char_p, C.POINTER(p), C.POINTER(C.c_int64)])
    sig("plade_stats_get", argtypes=[p, C.POINTER(C.c_char_p), C.POINTER(C.POINTER(C.c_double)), C.POINTER(i32)])
    sig("plade_kernel_time", argtypes=[p, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)])
    sig("plade_plane_component", argtypes=[p, p, u32, p, p, p, u32, f, C.c_int, f, p, p, p, p])
    sig("plade_sort_pairs", argtypes=[p, p, p, u32, C.c_int, C.c_int, p, p])
    sig("plade_selftest_readback", argtypes=[p, u32, u32, C.POINTER(u32)])
    sig("plade_host_pin", argtypes=[p, p, C.c_size_t])
    sig("plade_host_unpin", argtypes=[p, p])
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# name -> dtype of the intermediates plade_dump_get can return (same names as the oracle's dump)
DUMP_FIELDS = {
    "average_spacing": np.float32, "scale": np.float32,
    "tgt_ds": np.float32, "src_ds": np.float32, "tgt_bcenter": np.float32, "src_bcenter": np.float32,
    "tgt_radius": np.float64, "src_radius": np.float64,
    "tgt_plane_center_radius": np.float32, "src_plane_center_radius": np.float32,
    "tgt_plane_four": np.float32, "src_plane_four": np.float32,
    "tgt_plane_ds_offsets": np.int32, "src_plane_ds_offsets": np.int32,
    "tgt_plane_ds": np.float32, "src_plane_ds": np.float32,
    "tgt_lines": np.float32, "src_lines": np.float32,
    "tgt_desc": np.float32, "src_desc": np.float32,
    "match_offsets": np.int64, "match_nbr": np.int32, "match_dist2": np.float64,
    "initial_RT": np.float32, "cluster_sizes": np.int32, "cluster_seeds": np.int32,
    "plane_match_counts": np.int32, "pen_tested": np.int32, "pen_flags": np.int32,
    "candidates": np.float32, "candidate_centers": np.float32, "overlap_counts": np.int32,
    "scores": np.float32, "best_index": np.int32,
    "tgt_planes": np.float32, "tgt_plane_offsets": np.int32, "tgt_plane_idx": np.int32,
    "src_planes": np.float32, "src_plane_offsets": np.int32, "src_plane_idx": np.int32,
}


def device_synchronize(device=0):
    """hipDeviceSynchronize on `device` through the library (no torch needed in a single-GPU process)."""
    rc = load_library().plade_device_synchronize(int(device))
    if rc != 0:
        raise PladeError(rc, "plade_device_synchronize failed")


def default_params():
    """plade_default_params (pure: needs no GPU): the library's shipped defaults = the reference's behaviour."""
    prm = Params()
    load_library().plade_default_params(C.byref(prm))
    return prm


def read_ply(path):
    """plade_ply_read (no GPU): the CLI's PLY ingest -> (N, 6) float32 array x y z nx ny nz; raises PladeError with the
    reader's message where the reference's load_ply_cloud (code/PLADE/util.cpp:1505-1546) returns false."""
    L = load_library()
    ptr, n = C.POINTER(C.c_float)(), C.c_uint64(0)
    err = C.create_string_buffer(512)
    rc = L.plade_ply_read(os.fsencode(path), C.byref(ptr), C.byref(n), err, len(err))
    if rc != 0:
        raise PladeError(rc, err.value.decode(errors="replace"))
    try:
        return np.ctypeslib.as_array(ptr, shape=(n.value, 6)).copy()
    finally:
        L.plade_ply_free(ptr)


def line_solver_host(kind, a, b, c, d):
    """Host seam (no GPU): the register form of the reference's SVD solver as the kernels inline it -- kind 0: closest points of
    n line pairs -> (q1, q2, ok), kind 1: meeting points -> (point, ok); ok 1 solved / 0 rank-deficient / -1 guard fired."""
    a, b, c, d = (_f32(x).reshape(-1, 3) for x in (a, b, c, d))
    n = len(a)
    o1, o2, ok = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
    rc = load_library().plade_diag_line_solver_host(int(kind), _ptr(a), _ptr(b), _ptr(c), _ptr(d), n, _ptr(o1), _ptr(o2), _ptr(ok))
    if rc != 0:
        raise PladeError(rc, "plade_diag_line_solver_host failed")
    return (o1, o2, ok) if kind == 0 else (o1, ok)


def cluster_order(sizes, mode=0, depth_limit=-1):
    """Host seam: the order std::sort(..., myCompareGreater) (util.cpp:335-345) gives clusters of these sizes -- mode 0: the
    library's implementation, 1: std::sort, 2 / 3: block-wise / sequential partition at a given recursion depth limit."""
    a = _f32(sizes).reshape(-1)
    out = np.zeros(len(a), np.int32)
    rc = load_library().plade_diag_cluster_order(a.ctypes.data_as(C.c_void_p), len(a), int(mode), int(depth_limit),
                                                 out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise PladeError(rc, "plade_diag_cluster_order failed")
    return out


class Cloud:
    """Device-resident oriented point cloud (plade_cloud)."""

    def __init__(self, ctx, pos_nrm):
        self.ctx = ctx
        a = _f32(pos_nrm)
        if a.ndim != 2 or a.shape[1] != 6:
            raise ValueError(f"Cloud: an (N, 6) array x y z nx ny nz is required, got shape {a.shape}")
        self.n = len(a)
        self.h = C.c_void_p()
        ctx._check(ctx.L.plade_cloud_upload(ctx.h, _ptr(a), self.n, C.byref(self.h)))

    def free(self):
        if self.h:
            self.ctx.L.plade_cloud_free(self.ctx.h, self.h)
            self.h = C.c_void_p()


class Context:
    """One HIP stream + scratch pools on one GPU (plade_ctx)."""

    def __init__(self, device=0, **params):
        self.L = load_library()
        self.h = C.c_void_p()
        rc = self.L.plade_ctx_create(device, C.byref(self.h))
        if rc != 0:
            raise PladeError(rc, "plade_ctx_create failed (no gfx950 device visible?)")
        self.params = Params()
        self.L.plade_default_params(C.byref(self.params))
        if params:
            self.set_params(**params)

    def close(self):
        if self.h:
            self.L.plade_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise PladeError(rc, self.L.plade_last_error(self.h).decode(errors="replace"))
        return rc

    def set_params(self, **kw):
        for k, v in kw.items():
            setattr(self.params, k, v)
        self._check(self.L.plade_set_params(self.h, C.byref(self.params)))

    def closest_points(self, u1, p1, u2, p2, mode=0):
        """Seam of ComputeNearstTwoPointsOfTwo3DLine (util.cpp:1167-1229) for n line pairs; mode 0 closed form, 1 / "svd_fp32"
        the reference's 9 x 9 float SVD solve.  Returns q1, q2 (n x 3), len (n, float64), ok (n, int32)."""
        u1, p1, u2, p2 = (_f32(a).reshape(-1, 3) for a in (u1, p1, u2, p2))
        n = len(u1)
        if not (len(p1) == len(u2) == len(p2) == n):
            raise ValueError("closest_points: the four arrays must hold the same number of rows")
        q1, q2 = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        ln, ok = np.zeros(n, np.float64), np.zeros(n, np.int32)
        self._check(self.L.plade_closest_points(self.h, 1 if mode in (1, "svd_fp32") else 0, _ptr(u1), _ptr(p1), _ptr(u2), _ptr(p2), n,
                                                _ptr(q1), _ptr(q2), _ptr(ln), _ptr(ok)))
        return q1, q2, ln, ok

    def lines_meet(self, v1, p1, v2, p2, mode=0):
        """Seam of ComputeIntersectionPointOf23DLine (util.cpp:1461-1500); returns points (n x 3), ok (n)."""
        v1, p1, v2, p2 = (_f32(a).reshape(-1, 3) for a in (v1, p1, v2, p2))
        n = len(v1)
        if not (len(p1) == len(v2) == len(p2) == n):
            raise ValueError("lines_meet: the four arrays must hold the same number of rows")
        out, ok = np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
        self._check(self.L.plade_lines_meet(self.h, 1 if mode in (1, "svd_fp32") else 0, _ptr(v1), _ptr(p1), _ptr(v2), _ptr(p2), n, _ptr(out), _ptr(ok)))
        return out, ok

    def sort_pairs(self, keys, vals, bits=None):
        """Diagnostic seam: the device-wide stable radix sort (radix_sort.hip).  keys uint32 or uint64."""
        k = np.ascontiguousarray(keys)
        assert k.dtype in (np.uint32, np.uint64)
        v = np.ascontiguousarray(vals, np.uint32)
        kb = k.dtype.itemsize
        ko, vo = np.empty_like(k), np.empty_like(v)
        self._check(self.L.plade_sort_pairs(self.h, k.ctypes.data_as(C.c_void_p), _ptr(v), len(k), kb,
                                            int(bits if bits is not None else 8 * kb), ko.ctypes.data_as(C.c_void_p), _ptr(vo)))
        return ko, vo

    def sort_segments(self, keys, vals, seg_off, bits=32):
        """The radix sort over up to 16 independent arrays in one launch sequence: segment s = [seg_off[s], seg_off[s + 1])."""
        k = np.ascontiguousarray(keys, np.uint32)
        v = np.ascontiguousarray(vals, np.uint32)
        so = np.ascontiguousarray(seg_off, np.uint32)
        ko, vo = np.empty_like(k), np.empty_like(v)
        self._check(self.L.plade_sort_segments(self.h, _ptr(k), _ptr(v), _ptr(so), len(so) - 1, int(bits), _ptr(ko), _ptr(vo)))
        return ko, vo

    def selftest_readback(self, n_ranges, words):
        """Test seam: n_ranges device arrays of `words` words through the library's device -> host hand-over; returns the number
        of words that arrived wrong."""
        bad = C.c_uint32(0)
        self._check(self.L.plade_selftest_readback(self.h, int(n_ranges), int(words), C.byref(bad)))
        return int(bad.value)

    # ---- seams -----------------------------------------------------------------------------
    def score_planes(self, pos_nrm, shape_index, planes, eps, cos_thresh, want_indices=False):
        pn = _f32(pos_nrm)
        pl = _f32(planes).reshape(-1, 4)
        si = _i32(shape_index) if shape_index is not None else None
        n, h = len(pn), len(pl)
        counts = np.zeros(h, np.uint32)
        idx = np.zeros((h, n), np.uint32) if want_indices else None
        self._check(self.L.plade_score_planes(self.h, _ptr(pn), _ptr(si), n, _ptr(pl), h, eps, cos_thresh,
                                              _ptr(counts), _ptr(idx), n if want_indices else 0))
        if want_indices:
            return counts, [idx[j, : counts[j]].astype(np.int32) for j in range(h)]
        return counts

    def score_planes_subset(self, pos_nrm, shape_index, sub_index, planes, eps, cos_thresh):
        """Seam S1a on a subset (k_r_score_sub): (counts per hypothesis, unassigned subset points)."""
        pn = _f32(pos_nrm)
        pl = _f32(planes).reshape(-1, 4)
        si = _i32(shape_index) if shape_index is not None else None
        sub = np.ascontiguousarray(sub_index, dtype=np.uint32)
        counts = np.zeros(len(pl), np.uint32)
        un = C.c_uint32()
        self._check(self.L.plade_score_planes_subset(self.h, _ptr(pn), _ptr(si), len(pn), _ptr(sub), len(sub), _ptr(pl), len(pl),
                                                     eps, cos_thresh, _ptr(counts), C.byref(un)))
        return counts, un.value

    def plane_component(self, pos_nrm, normal, point, idx, bitmap_eps, closing_filter, w_eps):
        """Seam S1c: (kept indices, LS fit[7], weighted score) of one plane candidate's score list."""
        pn, nn, pp, ii = _f32(pos_nrm), _f32(normal), _f32(point), _i32(idx)
        kept = np.zeros(max(len(ii), 1), np.int32)
        nk = C.c_uint32()
        fit = np.zeros(7, np.float32)
        ws = C.c_double()
        self._check(self.L.plade_plane_component(self.h, _ptr(pn), len(pn), _ptr(nn), _ptr(pp), _ptr(ii), len(ii),
                                                 C.c_float(bitmap_eps), int(closing_filter), C.c_float(w_eps), _ptr(kept),
                                                 C.byref(nk), _ptr(fit), C.byref(ws)))
        return kept[: nk.value].copy(), fit, ws.value

    def extract_planes(self, pos_nrm, min_support, dist_rel=0.005, bitmap_rel=0.02, cos_thresh=0.8,
                       overlook=0.001, max_planes=256):
        pn = _f32(pos_nrm)
        n = len(pn)
        planes = np.zeros((max_planes, 4), np.float32)
        offs = np.zeros(max_planes + 1, np.int32)
        idx = np.zeros(n, np.int32)
        npl = C.c_uint32()
        self._check(self.L.plade_extract_planes(self.h, _ptr(pn), n, min_support, dist_rel, bitmap_rel, cos_thresh,
                                                overlook, _ptr(planes), _ptr(offs), _ptr(idx), max_planes,
                                                C.byref(npl)))
        p = npl.value
        return planes[:p].copy(), offs[: p + 1].copy(), idx[: offs[p]].copy()

    def match_descriptors(self, qry, tgt, radius=0.04):
        q = _f32(qry).reshape(-1, 8)
        t = _f32(tgt).reshape(-1, 8)
        off = np.zeros(len(q) + 1, np.int64)
        total = C.c_uint64()
        self._check(self.L.plade_match_descriptors(self.h, _ptr(q), len(q), _ptr(t), len(t), radius, _ptr(off), None,
                                                   None, 0, C.byref(total)))
        m = total.value
        nbr = np.zeros(max(m, 1), np.uint32)
        d2 = np.zeros(max(m, 1), np.float64)
        self._check(self.L.plade_match_descriptors(self.h, _ptr(q), len(q), _ptr(t), len(t), radius, _ptr(off),
                                                   _ptr(nbr), _ptr(d2), m, C.byref(total)))
        return off, nbr[:m].astype(np.int32), d2[:m]

    def cluster_transforms(self, t_xyz, euler, dist_threshold, angle_gate):
        """Seam of the clustering stage (util.cpp:1245-1277): (cluster index per candidate, number of clusters)."""
        t, e = _f32(t_xyz).reshape(-1, 3), _f32(euler).reshape(-1, 3)
        out = np.full(len(t), -1, np.int32)
        n = C.c_uint32()
        self._check(self.L.plade_cluster_transforms(self.h, _ptr(t), _ptr(e), len(t), dist_threshold, angle_gate, _ptr(out), C.byref(n)))
        return out, n.value

    def overlap_counts(self, src_ds, tgt_ds, T, centers, src_radius, inlier_dist):
        s, t = _f32(src_ds), _f32(tgt_ds)
        T = _f32(T).reshape(-1, 16)
        c = _f32(centers).reshape(-1, 3)
        counts = np.zeros(len(T), np.int32)
        self._check(self.L.plade_overlap_counts(self.h, _ptr(s), len(s), _ptr(t), len(t), _ptr(T), len(T), _ptr(c),
                                                src_radius, inlier_dist, _ptr(counts)))
        return counts

    def average_spacing(self, pts, k=6, samples=10000):
        a = _f32(pts)
        out = C.c_float()
        self._check(self.L.plade_average_spacing(self.h, _ptr(a), len(a), a.shape[1], k, samples, C.byref(out)))
        return np.float32(out.value)

    def voxel_downsample(self, pts, leaf):
        a = _f32(pts)
        out = np.zeros((len(a), 3), np.float32)
        n = C.c_uint32()
        self._check(self.L.plade_voxel_downsample(self.h, _ptr(a), len(a), a.shape[1], leaf, _ptr(out), C.byref(n)))
        return out[: n.value].copy()

    # ---- registration() overloads (code/PLADE/plade.h) --------------------------------------
    def registration_planes(self, tgt, src, tgt_planes, src_planes):
        """plade.h:74.  Returns (ok, T 4x4)."""
        tgt, src = _f32(tgt), _f32(src)
        tc, to, ti = _f32(tgt_planes[0]), _i32(tgt_planes[1]), _i32(tgt_planes[2])
        sc, so, si = _f32(src_planes[0]), _i32(src_planes[1]), _i32(src_planes[2])
        T = np.zeros((4, 4), np.float32)
        rc = self._check(self.L.plade_registration_planes(self.h, _ptr(tgt), len(tgt), _ptr(src), len(src), _ptr(tc),
                                                          _ptr(to), _ptr(ti), len(tc), _ptr(sc), _ptr(so), _ptr(si),
                                                          len(sc), _ptr(T)), allow=(PLADE_EFAIL,))
        return rc == 0, T

    def registration(self, tgt, src):
        """plade.h:58.  Returns (ok, T 4x4)."""
        tgt, src = _f32(tgt), _f32(src)
        self._check_cloud(tgt, "registration"); self._check_cloud(src, "registration")
        T = np.zeros((4, 4), np.float32)
        rc = self._check(self.L.plade_registration(self.h, _ptr(tgt), len(tgt), _ptr(src), len(src), _ptr(T)),
                         allow=(PLADE_EFAIL,))
        return rc == 0, T

    def registration_next(self, tgt, src, next_tgt=None, next_src=None):
        """plade.h:58 in batch mode: registers (tgt, src) and starts the upload of the pair the next call will be handed.
        The arrays must be C-contiguous float32 and stay alive and unchanged until that call."""
        for a in (tgt, src, next_tgt, next_src):
            if a is not None:
                self._check_cloud(a, "registration_next")
        T = np.zeros((4, 4), np.float32)
        nt, ns = (next_tgt, next_src) if next_tgt is not None and next_src is not None else (None, None)
        self._announced = [(nt, ns)] if nt is not None else None      # alive until the next call consumes or drops them
        rc = self._check(self.L.plade_registration_next(self.h, _ptr(tgt), len(tgt), _ptr(src), len(src), _ptr(nt),
                                                        len(nt) if nt is not None else 0, _ptr(ns),
                                                        len(ns) if ns is not None else 0, _ptr(T)), allow=(PLADE_EFAIL,))
        return rc == 0, T

    @staticmethod
    def _cloud_table(arrs):
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        ns = (C.c_uint32 * len(arrs))(*[len(a) for a in arrs])
        return ptrs, ns

    @staticmethod
    def _check_cloud(a, what):
        """The library reads len(a) x 6 floats behind the pointer: anything else than a C-contiguous (N, 6) float32 array would
        make it read past the buffer (python -O strips asserts, so this raises)."""
        if not isinstance(a, np.ndarray) or a.dtype != np.float32 or a.ndim != 2 or a.shape[1] != 6 or not a.flags["C_CONTIGUOUS"]:
            raise ValueError(f"{what}: a C-contiguous float32 array of shape (N, 6) is required, got "
                             f"{getattr(a, 'dtype', type(a))} {getattr(a, 'shape', '')}")
        if len(a) == 0:
            raise ValueError(f"{what}: empty cloud")

    def registration_pairs(self, pairs, next_pairs=None, raise_on_error=True):
        """plade.h:58 in batch mode, one GROUP of 1..8 pairs per call (plade_registration_pairs): pairs = [(tgt, src), ...];
        next_pairs = the pairs the next call on this context will be handed (their upload is started now).  The arrays must be
        C-contiguous (N, 6) float32 and stay alive and UNCHANGED until that call (the context keeps a reference to them until
        then; changing their contents meanwhile is the caller's race).  Returns [(ok, T 4x4), ...]; with raise_on_error=False a
        pair the library refused (status other than OK / EFAIL) does not raise: [(status, T), ...] with the PLADE_E* codes."""
        if not 1 <= len(pairs) <= 8 or (next_pairs and len(next_pairs) > 8):
            raise ValueError("registration_pairs: a group holds 1..8 pairs")
        for pr in list(pairs) + list(next_pairs or []):
            for a in pr[:2]:
                self._check_cloud(a, "registration_pairs")
        self._announced = [(p[0], p[1]) for p in next_pairs] if next_pairs else None   # alive until the next call consumes or drops them
        k = len(pairs)
        tp, tn = self._cloud_table([p[0] for p in pairs])
        sp, sn = self._cloud_table([p[1] for p in pairs])
        nk = len(next_pairs) if next_pairs else 0
        if nk:
            ntp, ntn = self._cloud_table([p[0] for p in next_pairs])
            nsp, nsn = self._cloud_table([p[1] for p in next_pairs])
        else:
            ntp = ntn = nsp = nsn = None
        T = np.zeros((k, 4, 4), np.float32)
        st = np.zeros(k, np.int32)
        self._check(self.L.plade_registration_pairs(self.h, k, tp, tn, sp, sn, nk, ntp, ntn, nsp, nsn, _ptr(T), _ptr(st)))
        if not raise_on_error:
            return [(int(st[i]), T[i].copy()) for i in range(k)]
        for i in range(k):
            if st[i] not in (0, PLADE_EFAIL):
                raise PladeError(int(st[i]), self.pair_error(i))
        return [(bool(st[i] == 0), T[i].copy()) for i in range(k)]

    def registration_pairs_dev(self, clouds, raise_on_error=True):
        """The same group call on resident clouds: clouds = [(tgt Cloud, src Cloud), ...]."""
        k = len(clouds)
        if not 1 <= k <= 8:
            raise ValueError("registration_pairs_dev: a group holds 1..8 pairs")
        tp = (C.c_void_p * k)(*[c[0].h.value for c in clouds])
        sp = (C.c_void_p * k)(*[c[1].h.value for c in clouds])
        T = np.zeros((k, 4, 4), np.float32)
        st = np.zeros(k, np.int32)
        self._check(self.L.plade_registration_pairs_dev(self.h, k, tp, sp, _ptr(T), _ptr(st)))
        if not raise_on_error:
            return [(int(st[i]), T[i].copy()) for i in range(k)]
        for i in range(k):
            if st[i] not in (0, PLADE_EFAIL):
                raise PladeError(int(st[i]), self.pair_error(i))
        return [(bool(st[i] == 0), T[i].copy()) for i in range(k)]

    def set_candidate_shard(self, rank, world, exchange=None, min_candidates=0):
        """Second sharding axis (plade_set_candidate_shard): with world > 1 this context scores only candidates k % world == rank
        of a registration's verification and calls exchange(values: int32 numpy view of all words, rank, world), which must fill
        in the other ranks' words (an all-gather).  world <= 1 switches it off."""
        if world <= 1 or exchange is None:
            self._shard_cb = None
            self._check(self.L.plade_set_candidate_shard(self.h, 0, 1, 0, EXCHANGE_FN(0), None))
            return

        def cb(user, values, count, r, w):
            try:
                exchange(np.ctypeslib.as_array(values, shape=(count,)), int(r), int(w))
                return 0
            except Exception:      # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return -1
        self._shard_cb = EXCHANGE_FN(cb)      # keep the trampoline alive
        self._check(self.L.plade_set_candidate_shard(self.h, int(rank), int(world), int(min_candidates), self._shard_cb, None))

    def diag_launches(self, count, blocks=1, mbytes=0):
        """Diagnostic: `count` launches of an empty (or memory-streaming) kernel on this context's stream, then a wait."""
        self._check(self.L.plade_diag_launches(self.h, int(count), int(blocks), int(mbytes)))

    def _pair_handle(self, index):
        h = self.L.plade_pair_ctx(self.h, int(index))
        if not h:
            raise PladeError(PLADE_EINVAL, f"no pair {index} on this context")
        return C.c_void_p(h)

    def pair_error(self, index):
        return self.L.plade_last_error(self._pair_handle(index)).decode(errors="replace")

    def registration_minsupport(self, tgt, src, ms_t, ms_s):
        """plade.h:91."""
        tgt, src = _f32(tgt), _f32(src)
        T = np.zeros((4, 4), np.float32)
        rc = self._check(self.L.plade_registration_minsupport(self.h, _ptr(tgt), len(tgt), _ptr(src), len(src), ms_t,
                                                              ms_s, _ptr(T)), allow=(PLADE_EFAIL,))
        return rc == 0, T

    def upload(self, pos_nrm):
        return Cloud(self, pos_nrm)

    def pin(self, arr):
        """Page-lock a C-contiguous float32 array the caller keeps alive (plade_host_pin); registration() calls that are
        handed this very array then upload it by asynchronous DMA."""
        assert arr.dtype == np.float32 and arr.flags["C_CONTIGUOUS"]
        self._check(self.L.plade_host_pin(self.h, _ptr(arr), arr.nbytes))
        return arr

    def unpin(self, arr):
        self._check(self.L.plade_host_unpin(self.h, _ptr(arr)))

    def registration_dev(self, tgt_cloud, src_cloud):
        T = np.zeros((4, 4), np.float32)
        rc = self._check(self.L.plade_registration_dev(self.h, tgt_cloud.h, src_cloud.h, _ptr(T)), allow=(PLADE_EFAIL,))
        return rc == 0, T

    # ---- instrumentation ---------------------------------------------------------------------
    def dump(self, pair=0):
        out = {}
        h = self._pair_handle(pair) if pair else self.h
        for name, dt in DUMP_FIELDS.items():
            ptr = C.c_void_p()
            nb = C.c_int64()
            if self.L.plade_dump_get(h, name.encode(), C.byref(ptr), C.byref(nb)) != 0:
                continue
            if nb.value == 0 or not ptr.value:
                out[name] = np.zeros(0, dt)
                continue
            buf = (C.c_char * nb.value).from_address(ptr.value)
            out[name] = np.frombuffer(bytes(buf), dtype=dt).copy()
        return out

    def stats(self, pair=0):
        names = C.c_char_p()
        vals = C.POINTER(C.c_double)()
        cnt = C.c_int32()
        self._check(self.L.plade_stats_get(self._pair_handle(pair) if pair else self.h, C.byref(names), C.byref(vals), C.byref(cnt)))
        ns = names.value.decode().strip(";").split(";") if cnt.value else []
        return {ns[i]: vals[i] for i in range(cnt.value)}

    def kernel_time(self, which, iters=20):
        t = C.c_double()
        b = C.c_double()
        self._check(self.L.plade_kernel_time(self.h, which.encode(), iters, C.byref(t), C.byref(b)))
        return t.value, b.value
