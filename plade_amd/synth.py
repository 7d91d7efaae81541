"""Deterministic synthetic indoor scan pairs (SURVEY.md section 8d -- the reference
ships no generator, this is the spec): a 10 x 8 x 3 m box room (6 faces) plus 8
axis-aligned furniture boxes with 3 visible faces each (~30 planes, each >= ~1 %
of the points), position noise N(0, 5 mm) along the normal, oriented normals with
N(0, 0.02) jitter, 3 % uniform outliers.  Target = whole scene (seed A); source =
the same scene re-sampled (seed B), cropped to an oblique half-space and moved by
a seeded ground-truth SE(3).  Scene centred on the origin to keep fp32 noise small.
"""
import numpy as np

ROOM = np.array([10.0, 8.0, 3.0])


def _furniture(rng, n_boxes, ROOM=ROOM):
    """Non-overlapping boxes standing on the floor; returns (lo, hi) corner arrays."""
    boxes = []
    tries = 0
    while len(boxes) < n_boxes and tries < 20000:
        tries += 1
        zmax = 2.45 if ROOM[2] == 3.0 else ROOM[2] - 0.55
        size = np.array([rng.uniform(1.2, 2.0), rng.uniform(1.1, 1.8), rng.uniform(0.8, zmax)])
        # keep every face >= 0.5 m from the parallel room face and from the other boxes' parallel
        # faces: Schnabel's global scoring uses 3 eps = 15 cm here and would merge closer coplanar-ish
        # faces into one shape (in the reference as much as in this implementation)
        lo_xy = np.array([rng.uniform(-ROOM[0] / 2 + 0.5, ROOM[0] / 2 - 0.5 - size[0]),
                          rng.uniform(-ROOM[1] / 2 + 0.5, ROOM[1] / 2 - 0.5 - size[1])])
        lo = np.array([lo_xy[0], lo_xy[1], -ROOM[2] / 2])
        hi = lo + size
        ok = True
        for (l2, h2) in boxes:
            if np.all(lo[:2] < h2[:2] + 0.2) and np.all(hi[:2] > l2[:2] - 0.2):
                ok = False
                break
            if abs(hi[2] - h2[2]) < 0.2 or min(abs(lo[0] - l2[0]), abs(lo[0] - h2[0]), abs(hi[0] - l2[0]), abs(hi[0] - h2[0])) < 0.2 \
                    or min(abs(lo[1] - l2[1]), abs(lo[1] - h2[1]), abs(hi[1] - l2[1]), abs(hi[1] - h2[1])) < 0.2:
                ok = False
                break
        if hi[2] > ROOM[2] / 2 - 0.5:
            ok = False
        if ok:
            boxes.append((lo, hi))
    return boxes


def _tilted_furniture(rng, n_boxes, ROOM):
    """BASELINE configs[4]: ~100 planes need pieces in general position (random yaw, up to 20 degrees of pitch / roll), so
    that no two faces are parallel: axis-aligned furniture in a large hall merges into a few dozen planes, because
    Schnabel's global scoring takes everything within 3 eps = 1.5 % of the hall's width of a plane, and its connected-
    component step (bitmap pixel = 2 % of the width, closing, 8-connectivity) joins what is less than ~3 pixels apart.
    The pieces therefore sit on a jittered grid in two layers (standing / suspended), ~10 % of the hall's width in size
    and >= 7 % of it apart.  Returns a list of (centre, R (3x3, columns = box axes), half sizes)."""
    W = max(ROOM[0], ROOM[1])
    per_layer = (n_boxes + 1) // 2
    nx = max(1, int(round(np.sqrt(per_layer * ROOM[0] / ROOM[1]))))
    ny = (per_layer + nx - 1) // nx
    cell = np.array([ROOM[0] / nx, ROOM[1] / ny])
    boxes = []
    for k in range(n_boxes):
        layer, q = k // per_layer, k % per_layer
        ix, iy = q % nx, q // nx
        half = np.array([rng.uniform(1.1, 1.4), rng.uniform(0.9, 1.2), rng.uniform(0.8, 1.1)]) * (W / 32.0)
        yaw, pitch, roll = rng.uniform(0, 2 * np.pi), rng.uniform(-0.35, 0.35), rng.uniform(-0.35, 0.35)
        cz, sz, cy, sy, cx, sx = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        Rm = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
              @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
        jit = (rng.random(2) - 0.5) * 0.08 * cell
        c = np.array([-ROOM[0] / 2 + (ix + 0.5) * cell[0] + jit[0], -ROOM[1] / 2 + (iy + 0.5) * cell[1] + jit[1],
                      (-0.25 if layer == 0 else 0.25) * ROOM[2] + rng.uniform(-0.02, 0.02) * ROOM[2]])
        boxes.append((c, Rm, half))
    return boxes


# BASELINE configs[4] ("10M-pt dense scan pair, ~100 planes"): a 32 x 28 x 12 m hall with 32 pieces in general position,
# 6 + 3 x 32 = 102 faces.  Outliers 0.05 % instead of 3 %: uniform volumetric outliers at this point count put more than one
# normal-compatible outlier into every pixel of the connected-component bitmap (pixel = 2 % of the hall's width, slab
# = 6 eps = 3 % of it), which joins every pair of faces a common plane can be laid through into one shape -- in the
# reference's RANSAC as much as in this one -- and about 40 shapes are what is left of the 102 faces.
CONFIG4 = dict(n_boxes=32, room=(32.0, 28.0, 12.0), tilted=True, keep=0.85, outliers=0.0005)


def _faces(scene_seed, n_boxes, ROOM=ROOM, tilted=False):
    """List of faces: (origin, edge_u, edge_v, normal, weight)."""
    rng = np.random.default_rng(scene_seed)
    if tilted:
        h = ROOM / 2
        faces = []
        for ax in range(3):
            u, v = [a for a in range(3) if a != ax]
            for sgn in (-1, 1):
                o = -h.copy()
                o[ax] = sgn * h[ax]
                eu = np.zeros(3); eu[u] = ROOM[u]
                ev = np.zeros(3); ev[v] = ROOM[v]
                nrm = np.zeros(3); nrm[ax] = -sgn
                faces.append((o, eu, ev, nrm, ROOM[u] * ROOM[v]))
        for (c, Rm, half) in _tilted_furniture(rng, n_boxes, ROOM):
            signs = [rng.choice([-1, 1]) for _ in range(3)]
            signs[2] = 1                              # the top is always visible
            for ax in range(3):
                u, v = [a for a in range(3) if a != ax]
                nrm = signs[ax] * Rm[:, ax]
                o = c + signs[ax] * half[ax] * Rm[:, ax] - half[u] * Rm[:, u] - half[v] * Rm[:, v]
                faces.append((o, 2 * half[u] * Rm[:, u], 2 * half[v] * Rm[:, v], nrm, None))
        room_area = sum(f[4] for f in faces[:6])
        n_f = len(faces) - 6
        # the hall's faces share 40 % of the plane points by area, the furniture faces 60 % equally
        return [(o, eu, ev, nrm, (0.40 * w / room_area) if i < 6 else 0.60 / max(n_f, 1)) for i, (o, eu, ev, nrm, w) in enumerate(faces)]
    h = ROOM / 2
    faces = []
    # room faces, normals pointing to the interior
    for ax in range(3):
        u, v = [a for a in range(3) if a != ax]
        for sgn in (-1, 1):
            o = -h.copy()
            o[ax] = sgn * h[ax]
            eu = np.zeros(3); eu[u] = ROOM[u]
            ev = np.zeros(3); ev[v] = ROOM[v]
            nrm = np.zeros(3); nrm[ax] = -sgn
            faces.append((o, eu, ev, nrm, ROOM[u] * ROOM[v]))
    # furniture: top + two sides chosen per box, normals pointing out of the box (into the room)
    for (lo, hi) in _furniture(rng, n_boxes, ROOM):
        size = hi - lo
        sx = rng.choice([-1, 1]); sy = rng.choice([-1, 1])
        # top
        o = np.array([lo[0], lo[1], hi[2]])
        faces.append((o, np.array([size[0], 0, 0]), np.array([0, size[1], 0]), np.array([0, 0, 1.0]), None))
        # x side
        x = hi[0] if sx > 0 else lo[0]
        faces.append((np.array([x, lo[1], lo[2]]), np.array([0, size[1], 0]), np.array([0, 0, size[2]]),
                      np.array([float(sx), 0, 0]), None))
        y = hi[1] if sy > 0 else lo[1]
        faces.append((np.array([lo[0], y, lo[2]]), np.array([size[0], 0, 0]), np.array([0, 0, size[2]]),
                      np.array([0, float(sy), 0]), None))
    room_area = sum(f[4] for f in faces[:6])
    n_f = len(faces) - 6
    # furniture faces share 30 % of the plane points equally (each >= 1 % for <= 24 faces x 1.25 %)
    out = []
    for i, (o, eu, ev, nrm, w) in enumerate(faces):
        if i < 6:
            wt = 0.70 * w / room_area
        else:
            wt = 0.30 / max(n_f, 1)
        out.append((o, eu, ev, nrm, wt))
    return out


def sample_scene(n, scene_seed=0, sample_seed=1, n_boxes=8, noise=0.005, normal_jitter=0.02, outliers=0.03,
                 return_labels=False, room=None, tilted=False):
    """N x 6 float32 cloud; with return_labels also the generating face id per point (-1 = outlier).
    room: (W, D, H) in metres, default 10 x 8 x 3 (a larger room takes more furniture: BASELINE configs[4])."""
    ROOM = np.asarray(room, float) if room is not None else globals()["ROOM"]
    faces = _faces(scene_seed, n_boxes, ROOM, tilted)
    rng = np.random.default_rng(sample_seed)
    n_out = int(round(n * outliers))
    n_in = n - n_out
    w = np.array([f[4] for f in faces])
    w = w / w.sum()
    counts = np.floor(w * n_in).astype(int)
    counts[0] += n_in - counts.sum()
    pts = np.empty((n, 3)); nrm = np.empty((n, 3))
    labels = np.full(n, -1, np.int32)
    k = 0
    for fi, ((o, eu, ev, fn, _), c) in enumerate(zip(faces, counts)):
        labels[k:k + c] = fi
        a = rng.random(c)[:, None]; b = rng.random(c)[:, None]
        p = o + a * eu + b * ev + fn * rng.normal(0, noise, c)[:, None]
        nn = fn + rng.normal(0, normal_jitter, (c, 3))
        nn /= np.linalg.norm(nn, axis=1, keepdims=True)
        pts[k:k + c] = p; nrm[k:k + c] = nn
        k += c
    pts[k:] = (rng.random((n_out, 3)) - 0.5) * ROOM
    nn = rng.normal(size=(n_out, 3)); nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    nrm[k:] = nn
    perm = rng.permutation(n)
    cloud = np.concatenate([pts[perm], nrm[perm]], axis=1).astype(np.float32)
    if return_labels:
        return cloud, labels[perm]
    return cloud


def random_se3(seed, max_t=5.0):
    rng = np.random.default_rng(seed)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = rng.uniform(-max_t, max_t, 3)
    return T


def planes_from_labels(cloud, labels, min_points=50):
    """Ground-truth plane sets for the planes-given boundary (plade.h:74): per generating face an
    LS plane (unit n oriented like the mean point normal, d = -n.mean) and its point index list.
    Returns (coef P x 4 float32, offsets P+1 int32, idx int32)."""
    coef, offs, idx = [], [0], []
    for f in np.unique(labels[labels >= 0]):
        ids = np.nonzero(labels == f)[0].astype(np.int32)
        if len(ids) < min_points:
            continue
        p = cloud[ids, :3].astype(np.float64)
        c = p.mean(0)
        _, _, vt = np.linalg.svd(p - c, full_matrices=False)
        nrm = vt[2]
        if cloud[ids, 3:].astype(np.float64).mean(0) @ nrm < 0:
            nrm = -nrm
        coef.append([nrm[0], nrm[1], nrm[2], -(nrm @ c)])
        idx.append(ids)
        offs.append(offs[-1] + len(ids))
    return (np.asarray(coef, np.float32), np.asarray(offs, np.int32),
            np.concatenate(idx).astype(np.int32) if idx else np.zeros(0, np.int32))


def make_pair(n, seed=0, n_boxes=8, keep=0.6, return_labels=False, room=None, tilted=False, outliers=0.03):
    """Returns (target N x 6, source ~N x 6, T_gt 4x4) with T_gt mapping source -> target
    (+ the per-point face labels of both clouds when return_labels)."""
    target, tl = sample_scene(n, scene_seed=1000 + seed, sample_seed=2 * seed + 1, n_boxes=n_boxes, return_labels=True,
                              room=room, tilted=tilted, outliers=outliers)
    full, fl = sample_scene(int(n / keep), scene_seed=1000 + seed, sample_seed=2 * seed + 2, n_boxes=n_boxes,
                            return_labels=True, room=room, tilted=tilted, outliers=outliers)
    rng = np.random.default_rng(5000 + seed)
    d = np.array([1.0, 0.35 * rng.uniform(-1, 1), 0.0]); d /= np.linalg.norm(d)
    proj = full[:, :3] @ d
    cut = np.quantile(proj, keep)
    src_scene = full[proj <= cut]
    sl = fl[proj <= cut]
    T = random_se3(9000 + seed)  # source -> target
    Ti = np.linalg.inv(T)
    p = src_scene[:, :3].astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]
    nn = src_scene[:, 3:].astype(np.float64) @ Ti[:3, :3].T
    source = np.concatenate([p, nn], axis=1).astype(np.float32)
    if return_labels:
        return target, source, T.astype(np.float64), tl, sl
    return target, source, T.astype(np.float64)
