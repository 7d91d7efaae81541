"""ctypes bindings for the oracle (oracle/libplade_oracle.so) and, when it has been
built, for the compiled reference pieces (oracle/_ref/libplade_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under plade_amd/ may import this module.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libplade_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libplade_ref.so")

_p = C.c_void_p
_f = C.c_float
_i = C.c_int


def _ptr(a):
    return a.ctypes.data_as(_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def mirror_planes(planes):
    """The "unoriented normals" mode of the reference's README (README.md:109-110): "allowing a plane (a group of 3D
    points) to have two opposite orientations.  This way, more descriptors (considering both orientations for each plane)
    will be generated and matched."  Restated as a transformation of the TARGET plane set handed to registration():
    every plane (n, d) is followed by (-n, -d) with the same support, so the target's descriptor table holds every sign
    pattern of every pair of intersection lines (the reference ships no code for this mode: this is its specification
    here, mirrored by plade_amd/csrc/pipeline.h MirroredPlanes)."""
    coef, off, idx = _f32(planes[0]).reshape(-1, 4), _i32(planes[1]), _i32(planes[2])
    total = int(off[-1])
    return (np.concatenate([coef, -coef]).astype(np.float32),
            np.concatenate([off, total + off[1:]]).astype(np.int32),
            np.concatenate([idx[:total], idx[:total]]).astype(np.int32))


class Oracle:
    """CPU restatement of the reference's algorithm (oracle/plade_oracle.cpp)."""

    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make oracle` (or __graft_entry__.build())")
        L = self.L = C.CDLL(path)
        L.orc_average_spacing.restype = _f
        L.orc_average_spacing.argtypes = [_p, _i, _i, _i, _i]
        L.orc_weighted_score.restype = _f
        L.orc_weighted_score.argtypes = [_p, _i, _p, _p, _p, _i, _f]
        L.orc_cloud_scale.restype = _f
        L.orc_cloud_scale.argtypes = [_p, _i]
        L.orc_score_plane.argtypes = [_p, _p, _i, _p, _f, _f, _p, _p]
        L.orc_plane_from_points.argtypes = [_p, _p]
        L.orc_connected_component.argtypes = [_p, _i, _p, _p, _p, _i, _f, _i, _p]
        L.orc_ls_fit.argtypes = [_p, _i, _p, _i, _p]
        L.orc_voxel_downsample.argtypes = [_p, _i, _i, _f, _i, _p, _p]
        L.orc_bounding_box.argtypes = [_p, _i, _p, _p, _p]
        L.orc_bounding_box_mode.argtypes = [_p, _i, _i, _p, _p, _p]
        L.orc_intersection_line.argtypes = [_p, _p, _p, _p]
        L.orc_closest_points.argtypes = [_p, _p, _p, _p, _p, _p, _p]
        L.orc_match_descriptors.restype = C.c_int64
        L.orc_match_descriptors.argtypes = [_p, _i, _p, _i, _f, _p, _p, _p, C.c_int64]
        L.orc_umeyama3.argtypes = [_p, _p, _p]
        L.orc_selfadjoint_eig3.argtypes = [_p, _p, _p]
        L.orc_overlap_count.argtypes = [_p, _i, _p, _i, _p, _p, _f, _f]
        L.orc_reg_create.restype = _p
        L.orc_reg_destroy.argtypes = [_p]
        L.orc_registration.argtypes = [_p, _p, _i, _p, _i, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _p]
        L.orc_registration_sampled.argtypes = [_p, _p, _i, _p, _i, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p]
        L.orc_dump_get.argtypes = [_p, C.c_char_p, _p, _p]
        L.orc_set_closest_point_mode.argtypes = [_i]
        L.orc_intersection_point.argtypes = [_p, _p, _p, _p, _p]
        L.orc_solve_svd_f32.argtypes = [_p, _p, _p, _i, _i]
        L.orc_hypot.restype = C.c_double
        L.orc_hypot.argtypes = [C.c_double, C.c_double]
        L.orc_libm_hypot.restype = C.c_double
        L.orc_libm_hypot.argtypes = [C.c_double, C.c_double]
        L.orc_cluster_transforms.argtypes = [_p, _p, _i, _f, _f, _p]
        L.orc_euler_angles.argtypes = [_p, _p]
        L.orc_pen_walk.argtypes = [_p, _i, _p, _i, _p, _p, _p, _f, _f, _f, _p, _p, _p]

    def set_closest_point_mode(self, mode):
        """1 / "svd_fp32" (default, = the HIP path's default): the reference's fp32 cv::solve (DECOMP_SVD) restated from OpenCV 2.4
        (util.cpp:1183-1226, 1467-1497); 0 / "closed_form": exact fp64 closed form (the opt-in deviation of both)."""
        self.L.orc_set_closest_point_mode(1 if mode in (1, "svd_fp32") else 0)

    def reset_closest_point_mode(self):
        """Back to the default: the reference's arithmetic."""
        self.L.orc_set_closest_point_mode(1)

    def intersection_point(self, v1, p1, v2, p2):
        out = np.zeros(3, np.float32)
        rc = self.L.orc_intersection_point(_ptr(_f32(v1)), _ptr(_f32(p1)), _ptr(_f32(v2)), _ptr(_f32(p2)), _ptr(out))
        return rc, out

    def solve_svd_f32(self, A, B):
        A = _f32(A); B = _f32(B).reshape(-1)
        X = np.zeros(A.shape[1], np.float32)
        self.L.orc_solve_svd_f32(_ptr(A), _ptr(B), _ptr(X), A.shape[0], A.shape[1])
        return X

    # -- A9 / A11 pieces (G10, G11) ---------------------------------------------
    def euler_angles(self, R):
        """pcl::getEulerAngles of M rotation matrices (M x 3 x 3) -> M x 3 (roll, pitch, yaw)."""
        R = _f32(R).reshape(-1, 9)
        out = np.zeros((len(R), 3), np.float32)
        for i in range(len(R)):
            self.L.orc_euler_angles(_ptr(R[i]), _ptr(out[i]))
        return out

    def cluster_transforms(self, t_xyz, euler, distance_threshold, g_angle):
        """ClusterTransformation (util.cpp:1245-1277): cluster index per candidate, in creation order."""
        t, e = _f32(t_xyz).reshape(-1, 3), _f32(euler).reshape(-1, 3)
        out = np.full(len(t), -1, np.int32)
        n = self.L.orc_cluster_transforms(_ptr(t), _ptr(e), len(t), distance_threshold, g_angle, _ptr(out))
        return out, n

    def pen_walk(self, pts_a, pts_b, plane_b, start, direc, length, search_radius, min_distance):
        """One walk of AreTwoPlanesPenetrable (util.cpp:1379-1405): (positive, negative, skipped steps)."""
        a, b = _f32(pts_a).reshape(-1, 3), _f32(pts_b).reshape(-1, 3)
        pos, neg, sk = C.c_int(), C.c_int(), C.c_int()
        self.L.orc_pen_walk(_ptr(a), len(a), _ptr(b), len(b), _ptr(_f32(plane_b)), _ptr(_f32(start)), _ptr(_f32(direc)),
                            length, search_radius, min_distance, C.byref(pos), C.byref(neg), C.byref(sk))
        return pos.value, neg.value, sk.value

    # -- A3 ------------------------------------------------------------------
    def score_plane(self, pos_nrm, shape_index, plane4, eps, cos_thresh):
        pn = _f32(pos_nrm)
        n = len(pn)
        si = _i32(shape_index) if shape_index is not None else None
        idx = np.empty(n, np.int32)
        cnt = C.c_int32()
        self.L.orc_score_plane(_ptr(pn), _ptr(si), n, _ptr(_f32(plane4)), eps, cos_thresh, _ptr(idx), C.byref(cnt))
        return idx[: cnt.value].copy()

    def plane_from_points(self, tri9):
        out = np.zeros(4, np.float32)
        ok = self.L.orc_plane_from_points(_ptr(_f32(tri9)), _ptr(out))
        return bool(ok), out

    def connected_component(self, pos_nrm, normal, point, indices, bitmap_eps, filtering=True):
        pn = _f32(pos_nrm)
        ind = _i32(indices)
        out = np.empty(len(ind), np.int32)
        k = self.L.orc_connected_component(_ptr(pn), len(pn), _ptr(_f32(normal)), _ptr(_f32(point)), _ptr(ind),
                                           len(ind), bitmap_eps, int(filtering), _ptr(out))
        return out[:k].copy()

    def ls_fit(self, pos_nrm, indices):
        pn = _f32(pos_nrm)
        ind = _i32(indices)
        out = np.zeros(7, np.float32)
        self.L.orc_ls_fit(_ptr(pn), len(pn), _ptr(ind), len(ind), _ptr(out))
        return out

    def weighted_score(self, pos_nrm, normal, point, indices, eps):
        pn = _f32(pos_nrm)
        ind = _i32(indices)
        return self.L.orc_weighted_score(_ptr(pn), len(pn), _ptr(_f32(normal)), _ptr(_f32(point)), _ptr(ind), len(ind), eps)

    def cloud_scale(self, pos_nrm):
        pn = _f32(pos_nrm)
        return self.L.orc_cloud_scale(_ptr(pn), len(pn))

    # -- A13 -----------------------------------------------------------------
    def average_spacing(self, pts, k=6, samples=10000):
        a = _f32(pts)
        return self.L.orc_average_spacing(_ptr(a), len(a), a.shape[1], k, samples)

    def voxel_downsample(self, pts, leaf, sort_mode=0):
        a = _f32(pts)
        out = np.empty((len(a), 3), np.float32)
        n = C.c_int32()
        self.L.orc_voxel_downsample(_ptr(a), len(a), a.shape[1], leaf, sort_mode, _ptr(out), C.byref(n))
        return out[: n.value].copy()

    def bounding_box(self, xyz, sum_mode=0):
        a = _f32(xyz)
        c = np.zeros(3, np.float32)
        whd = np.zeros(3, np.float64)
        corners = np.zeros((8, 3), np.float32)
        rc = self.L.orc_bounding_box_mode(_ptr(a), len(a), sum_mode, _ptr(c), _ptr(whd), _ptr(corners))
        return rc, c, whd, corners

    # -- A6/A7/A8 ------------------------------------------------------------
    def intersection_line(self, a4, b4):
        v = np.zeros(3, np.float32)
        p = np.zeros(3, np.float32)
        rc = self.L.orc_intersection_line(_ptr(_f32(a4)), _ptr(_f32(b4)), _ptr(v), _ptr(p))
        return rc, v, p

    def closest_points(self, u1, p1, u2, p2):
        q1 = np.zeros(3, np.float32)
        q2 = np.zeros(3, np.float32)
        ln = C.c_double()
        rc = self.L.orc_closest_points(_ptr(_f32(u1)), _ptr(_f32(p1)), _ptr(_f32(u2)), _ptr(_f32(p2)), _ptr(q1), _ptr(q2), C.byref(ln))
        return rc, q1, q2, ln.value

    def match_descriptors(self, qry, tgt, radius=0.04):
        q = _f32(qry).reshape(-1, 8)
        t = _f32(tgt).reshape(-1, 8)
        off = np.zeros(len(q) + 1, np.int64)
        total = self.L.orc_match_descriptors(_ptr(q), len(q), _ptr(t), len(t), radius, _ptr(off), None, None, 0)
        nbr = np.empty(max(total, 1), np.int32)
        d2 = np.empty(max(total, 1), np.float64)
        self.L.orc_match_descriptors(_ptr(q), len(q), _ptr(t), len(t), radius, _ptr(off), _ptr(nbr), _ptr(d2), total)
        return off, nbr[:total], d2[:total]

    def umeyama3(self, src, dst):
        R = np.zeros((3, 3), np.float32)
        self.L.orc_umeyama3(_ptr(_f32(src)), _ptr(_f32(dst)), _ptr(R))
        return R

    def selfadjoint_eig3(self, cov):
        ev = np.zeros(3, np.float32)
        E = np.zeros((3, 3), np.float32)
        self.L.orc_selfadjoint_eig3(_ptr(_f32(cov)), _ptr(ev), _ptr(E))
        return ev, E

    # -- A12 -----------------------------------------------------------------
    def overlap_count(self, src_ds, tgt_ds, T, center, src_radius, inlier_dist):
        s = _f32(src_ds)
        t = _f32(tgt_ds)
        return self.L.orc_overlap_count(_ptr(s), len(s), _ptr(t), len(t), _ptr(_f32(T)), _ptr(_f32(center)),
                                        src_radius, inlier_dist)

    # -- whole deterministic stage ---------------------------------------------
    def registration(self, tgt, src, tgt_planes, src_planes, voxel_sort_mode=0, max_candidates=200, pen_stride=1,
                     unoriented_normals=False):
        """tgt/src: N x 6 float32. *_planes: (coef P x 4, offsets P+1, idx).  Returns (ok, T, dump dict).
        pen_stride > 1: sampled run for stress configurations (orc_registration_sampled) -- only every pen_stride-th
        candidate goes through the penetration test (pen_flags -1 elsewhere) and the run ends after that stage."""
        tgt = _f32(tgt)
        src = _f32(src)
        if unoriented_normals:
            tgt_planes = mirror_planes(tgt_planes)
        tc, to, ti = _f32(tgt_planes[0]), _i32(tgt_planes[1]), _i32(tgt_planes[2])
        sc, so, si = _f32(src_planes[0]), _i32(src_planes[1]), _i32(src_planes[2])
        T = np.zeros((4, 4), np.float32)
        h = self.L.orc_reg_create()
        try:
            ok = self.L.orc_registration_sampled(h, _ptr(tgt), len(tgt), _ptr(src), len(src), _ptr(tc), _ptr(to), _ptr(ti),
                                                 len(tc), _ptr(sc), _ptr(so), _ptr(si), len(sc), voxel_sort_mode,
                                                 max_candidates, pen_stride, _ptr(T))
            dump = _read_dump(self.L.orc_dump_get, h, ORC_DUMP_FIELDS)
        finally:
            self.L.orc_reg_destroy(h)
        return bool(ok), T, dump


# name -> dtype of every intermediate the oracle (and the product, under the same names) can dump
ORC_DUMP_FIELDS = {
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
    "scores": np.float32, "best_index": np.int32, "timing": np.float64, "timing_names": np.uint8,
}


def _read_dump(getter, h, fields):
    out = {}
    for name, dt in fields.items():
        ptr = C.c_void_p()
        nb = C.c_int64()
        if getter(h, name.encode(), C.byref(ptr), C.byref(nb)) != 0:
            continue
        if nb.value == 0 or not ptr.value:
            out[name] = np.zeros(0, dt)
            continue
        buf = (C.c_char * nb.value).from_address(ptr.value)
        out[name] = np.frombuffer(bytes(buf), dtype=dt).copy()
    if "timing_names" in out:
        out["timing_names"] = bytes(out["timing_names"]).decode().strip(";").split(";")
    return out


class Reference:
    """The compilable pieces of the real reference (oracle/ref/ref_shim.cpp over
    /root/reference/code/3rd_party/{ransac,ann_1.1.2,flann,eigen-3.4.0})."""

    def __init__(self, path=REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make ref` where /root/reference exists")
        L = self.L = C.CDLL(path)
        L.ref_cloud_scale.restype = _f
        L.ref_cloud_scale.argtypes = [_p, _i]
        L.ref_ransac_detect.argtypes = [_p, _i, C.c_uint, _f, _f, _f, _f, C.c_long, _p, _p, _p, _i, _p]
        L.ref_score_kat.argtypes = [_p, _i, _p, _p, _i, _f, _f, _p, _p, _p, _p, _p, _p]
        L.ref_connected_component.argtypes = [_p, _i, _p, _p, _p, _i, _f, _i, _p]
        L.ref_ls_fit.argtypes = [_p, _i, _p, _i, _p]
        L.ref_weighted_score.restype = _f
        L.ref_weighted_score.argtypes = [_p, _i, _p, _p, _p, _i, _f]
        L.ref_ann_radius_match.restype = C.c_long
        L.ref_ann_radius_match.argtypes = [_p, _i, _p, _i, _f, _p, _p, _p, C.c_long]
        L.ref_flann_build.restype = _p
        L.ref_flann_build.argtypes = [_p, _i]
        L.ref_flann_free.argtypes = [_p]
        L.ref_flann_radius.argtypes = [_p, _p, C.c_double, C.c_uint, _p, _p, _i]
        L.ref_flann_knn.argtypes = [_p, _p, _i, _p, _p]
        L.ref_overlap_count.argtypes = [_p, _i, _p, _i, _p, _f, _f]
        L.ref_umeyama3.argtypes = [_p, _p, _p]
        L.ref_selfadjoint_eig3.argtypes = [_p, _p, _p]
        L.ref_inverse4.argtypes = [_p, _p]
        L.ref_affine3.argtypes = [_p, _p, _p, _p]
        L.ref_format_matrix4.argtypes = [_p, C.c_char_p, _i]
        if hasattr(L, "ref_cluster_transforms"):
            L.ref_cluster_transforms.argtypes = [_p, _p, _i, _f, _f, _p]
            L.ref_pen_walk.argtypes = [_p, _i, _p, _i, _p, _p, _p, _f, _f, _f, _p, _p, _p]

    def cluster_transforms(self, t_xyz, euler, distance_threshold, g_angle):
        """pcl::ConditionalEuclideanClustering::segment + EnforceSimilarity over FLANN (ref_shim.cpp, G10)."""
        t, e = _f32(t_xyz).reshape(-1, 3), _f32(euler).reshape(-1, 3)
        out = np.full(len(t), -1, np.int32)
        n = self.L.ref_cluster_transforms(_ptr(t), _ptr(e), len(t), distance_threshold, g_angle, _ptr(out))
        return out, n

    def pen_walk(self, pts_a, pts_b, plane_b, start, direc, length, search_radius, min_distance):
        """One walk of AreTwoPlanesPenetrable over FLANN kd-trees (ref_shim.cpp, G11)."""
        a, b = _f32(pts_a).reshape(-1, 3), _f32(pts_b).reshape(-1, 3)
        pos, neg, sk = C.c_int(), C.c_int(), C.c_int()
        self.L.ref_pen_walk(_ptr(a), len(a), _ptr(b), len(b), _ptr(_f32(plane_b)), _ptr(_f32(start)), _ptr(_f32(direc)),
                            length, search_radius, min_distance, C.byref(pos), C.byref(neg), C.byref(sk))
        return pos.value, neg.value, sk.value

    def cloud_scale(self, pos_nrm):
        pn = _f32(pos_nrm)
        return self.L.ref_cloud_scale(_ptr(pn), len(pn))

    def ransac_detect(self, pos_nrm, min_support, dist_rel=0.005, bitmap_rel=0.02, normal_thresh=0.8,
                      overlook=0.001, fake_time=-1, max_planes=256):
        pn = _f32(pos_nrm)
        n = len(pn)
        planes = np.zeros((max_planes, 4), np.float32)
        offs = np.zeros(max_planes + 1, np.int32)
        idx = np.zeros(n, np.int32)
        rem = C.c_int()
        p = self.L.ref_ransac_detect(_ptr(pn), n, min_support, dist_rel, bitmap_rel, normal_thresh, overlook,
                                     fake_time, _ptr(planes), _ptr(offs), _ptr(idx), max_planes, C.byref(rem))
        return planes[:p].copy(), offs[: p + 1].copy(), idx[: offs[p]].copy()

    def score_kat(self, pos_nrm, shape_index_by_orig, tri, eps, normal_thresh):
        pn = _f32(pos_nrm)
        n = len(pn)
        tri = _f32(tri).reshape(-1, 9)
        h = len(tri)
        reordered = np.zeros((n, 6), np.float32)
        orig = np.zeros(n, np.int32)
        planes = np.zeros((h, 4), np.float32)
        ok = np.zeros(h, np.int32)
        counts = np.zeros(h, np.int32)
        lists = np.zeros((h, n), np.int32)
        self.L.ref_score_kat(_ptr(pn), n, _ptr(_i32(shape_index_by_orig)), _ptr(tri), h, eps, normal_thresh,
                             _ptr(reordered), _ptr(orig), _ptr(planes), _ptr(ok), _ptr(counts), _ptr(lists))
        return reordered, orig, planes, ok, counts, [lists[j, : counts[j]].copy() for j in range(h)]

    def connected_component(self, pos_nrm, normal, point, indices, bitmap_eps, filtering=True):
        pn = _f32(pos_nrm)
        ind = _i32(indices)
        out = np.empty(len(ind), np.int32)
        k = self.L.ref_connected_component(_ptr(pn), len(pn), _ptr(_f32(normal)), _ptr(_f32(point)), _ptr(ind),
                                           len(ind), bitmap_eps, int(filtering), _ptr(out))
        return out[:k].copy()

    def ls_fit(self, pos_nrm, indices):
        pn = _f32(pos_nrm)
        ind = _i32(indices)
        out = np.zeros(7, np.float32)
        self.L.ref_ls_fit(_ptr(pn), len(pn), _ptr(ind), len(ind), _ptr(out))
        return out

    def weighted_score(self, pos_nrm, normal, point, indices, eps):
        pn = _f32(pos_nrm)
        ind = _i32(indices)
        return self.L.ref_weighted_score(_ptr(pn), len(pn), _ptr(_f32(normal)), _ptr(_f32(point)), _ptr(ind), len(ind), eps)

    def ann_radius_match(self, qry, tgt, radius=0.04):
        q = _f32(qry).reshape(-1, 8)
        t = _f32(tgt).reshape(-1, 8)
        off = np.zeros(len(q) + 1, np.int64)
        total = self.L.ref_ann_radius_match(_ptr(t), len(t), _ptr(q), len(q), radius, _ptr(off), None, None, 0)
        nbr = np.empty(max(total, 1), np.int32)
        d = np.empty(max(total, 1), np.float32)
        self.L.ref_ann_radius_match(_ptr(t), len(t), _ptr(q), len(q), radius, _ptr(off), _ptr(nbr), _ptr(d), total)
        return off, nbr[:total], d[:total]

    def overlap_count(self, query_xyz, dest_xyz, center, query_radius, inlier_dist):
        q = _f32(query_xyz)
        d = _f32(dest_xyz)
        return self.L.ref_overlap_count(_ptr(q), len(q), _ptr(d), len(d), _ptr(_f32(center)), query_radius, inlier_dist)

    def knn_d2(self, cloud_xyz, queries, k):
        c = _f32(cloud_xyz)
        q = _f32(queries)
        h = self.L.ref_flann_build(_ptr(c), len(c))
        idx = np.zeros((len(q), k), np.int32)
        d = np.zeros((len(q), k), np.float32)
        for i in range(len(q)):
            self.L.ref_flann_knn(h, _ptr(q[i]), k, _ptr(idx[i]), _ptr(d[i]))
        self.L.ref_flann_free(h)
        return idx, d

    def radius_sets(self, cloud_xyz, queries, radius, max_nn=0):
        c = _f32(cloud_xyz)
        q = _f32(queries)
        h = self.L.ref_flann_build(_ptr(c), len(c))
        out = []
        buf = np.zeros(len(c), np.int32)
        for i in range(len(q)):
            k = self.L.ref_flann_radius(h, _ptr(q[i]), float(radius), max_nn, _ptr(buf), None, len(c))
            out.append(buf[:k].copy())
        self.L.ref_flann_free(h)
        return out

    def umeyama3(self, src, dst):
        T = np.zeros((4, 4), np.float32)
        self.L.ref_umeyama3(_ptr(_f32(src)), _ptr(_f32(dst)), _ptr(T))
        return T

    def selfadjoint_eig3(self, cov):
        ev = np.zeros(3, np.float32)
        E = np.zeros((3, 3), np.float32)
        self.L.ref_selfadjoint_eig3(_ptr(_f32(cov)), _ptr(ev), _ptr(E))
        return ev, E

    def inverse4(self, m):
        out = np.zeros((4, 4), np.float32)
        self.L.ref_inverse4(_ptr(_f32(m)), _ptr(out))
        return out

    def affine3(self, R, v, t):
        out = np.zeros(3, np.float32)
        self.L.ref_affine3(_ptr(_f32(R)), _ptr(_f32(v)), _ptr(_f32(t)), _ptr(out))
        return out

    def format_matrix4(self, m):
        buf = C.create_string_buffer(1024)
        n = self.L.ref_format_matrix4(_ptr(_f32(m)), buf, 1024)
        return buf.value.decode() if n >= 0 else None


def have_reference():
    return os.path.exists(REF_SO)
