"""tests/scenes.py -- synthetic meshes / cameras shared by the tests, the golden generator and bench.

Own code (no reference import).  The camera follows the reference's projection chain so that the
rasteriser sees realistic inputs: v_cam = v @ R + t; x = pp_x + X f_x / Z; y = -(pp_y + Y f_y / Z);
z = Z + 2.732051 (SoftRas' default look_at eye)  -- SURVEY.md Appendix A.1.
"""
import math

import numpy as np

LOOK_AT_Z = 1.0 / math.tan(math.radians(30.0)) + 1.0  # 2.7320508


def octahedron():
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float64)
    f = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4],
                  [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.int64)
    return v, f


def icosphere(subdiv=3):
    """Unit icosphere: subdiv 3 -> 642 verts / 1280 faces, subdiv 4 -> 2562 / 5120."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t],
         [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]]
    v = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    f = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4],
         [11, 10, 2], [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8],
         [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = nf
    return np.asarray(v, np.float64), np.asarray(f, np.int64)


def bottle_like(subdiv=3):
    """An icosphere squashed into an elongated, slightly waisted body (642/1280 at subdiv 3):
    same topology as the reference's bottle prior, different geometry."""
    v, f = icosphere(subdiv)
    v = v.copy()
    r = 0.42 + 0.10 * np.cos(2.2 * v[:, 1])
    v[:, 0] *= r
    v[:, 2] *= r
    return v, f


def random_rotations(n, rng):
    out = []
    for _ in range(n):
        q, r = np.linalg.qr(rng.standard_normal((3, 3)))
        q = q * np.sign(np.diag(r))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        out.append(q)
    return np.stack(out)


def camera_batch(n, rng, depth=5.0, foc=5.9, pp_range=0.2):
    """rotation [n,3,3], translation [n,1,3], foc [n,2], pp [n,2] in NDC units."""
    rot = random_rotations(n, rng)
    trans = np.zeros((n, 1, 3))
    trans[:, 0, 2] = depth + rng.uniform(-0.3, 0.3, n)
    trans[:, 0, :2] = rng.uniform(-0.1, 0.1, (n, 2))
    f = foc * rng.uniform(0.9, 1.1, (n, 2))
    pp = rng.uniform(-pp_range, pp_range, (n, 2))
    return rot, trans, f, pp


def project(verts, rot, trans, foc, pp):
    """verts [n,V,3] (or [V,3]) -> SoftRas-space vertices [n,V,3] float32 (x, y up, z + eye)."""
    if verts.ndim == 2:
        verts = np.broadcast_to(verts, (rot.shape[0],) + verts.shape)
    cam = verts @ rot + trans
    x = pp[:, None, 0] + cam[..., 0] * foc[:, None, 0] / cam[..., 2]
    y = -(pp[:, None, 1] + cam[..., 1] * foc[:, None, 1] / cam[..., 2])
    z = cam[..., 2] + LOOK_AT_Z
    return np.stack([x, y, z], -1).astype(np.float32)


def face_gather(per_vertex, faces):
    """[n,V,C] , [F,3] -> [n,F,3,C]"""
    return np.ascontiguousarray(per_vertex[:, faces])


def raster_inputs(verts, faces, n, seed, tex="rand"):
    """Convenience: (face_vertices [n,F,3,3], face_textures [n,F,3,3]) float32."""
    rng = np.random.default_rng(seed)
    rot, trans, foc, pp = camera_batch(n, rng)
    pv = project(verts, rot, trans, foc, pp)
    fv = face_gather(pv, faces)
    if tex == "rand":
        col = rng.uniform(0, 1, (n, verts.shape[0], 3)).astype(np.float32)
    elif tex == "depth":
        col = pv.copy()
        col[..., 2] -= LOOK_AT_Z
    else:
        col = np.broadcast_to(verts.astype(np.float32), (n,) + verts.shape).copy()
    return fv, face_gather(col, faces)
