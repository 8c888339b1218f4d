"""scp_amd/eval_nocs.py -- the NOCS-style 3-D box metrics of the test loop (CPU numpy; evaluation harness, not a GPU path).

  box_iou            exact IoU of two oriented boxes (the quantity objectron's dataset/iou.py computes with polygon clipping)
  get_best_iou       model/util/eval_utils.py:134-179: for y-symmetric categories (symmetry_idx 0) the best IoU over 18
                     rotations of the ground truth about its y axis, otherwise the plain IoU
Boxes are the 9x3 vertex arrays of objectron's `Box` (centre first, then corners in (x,y,z) sign order ---, --+, -+-, ...).

The intersection volume is computed here from the vertex set of the intersection polytope -- corners of one box inside the
other plus the crossings of one box's 12 edges with the other's 6 faces -- and scipy's ConvexHull, a different route to the
same number as objectron's Sutherland-Hodgman face clipping; pinned to the reference's values in tests/golden/posefit_metric.npz."""
import numpy as np
from scipy.spatial import ConvexHull

_EDGES = ((1, 5), (2, 6), (3, 7), (4, 8), (1, 3), (5, 7), (2, 4), (6, 8), (1, 2), (3, 4), (5, 6), (7, 8))
_EPS = 1e-9


def box_from_transformation(rotation, translation, scale):
    """objectron Box.from_transformation: unit box scaled, rotated (column-vector convention), translated -> [9,3]"""
    w, h, d = np.asarray(scale, np.float64) / 2.0
    local = np.array([[0, 0, 0], [-w, -h, -d], [-w, -h, d], [-w, h, -d], [-w, h, d], [w, -h, -d], [w, -h, d], [w, h, -d], [w, h, d]])
    return local @ np.asarray(rotation, np.float64).T + np.asarray(translation, np.float64).reshape(1, 3)


def box_fit(vertices):
    """objectron Box.fit: (rotation, translation, scale) of a 9x3 box by least squares"""
    v = np.asarray(vertices, np.float64)
    scale = np.array([np.mean([np.linalg.norm(v[a] - v[b]) for a, b in _EDGES[4 * k:4 * k + 4]]) for k in range(3)])
    local = box_from_transformation(np.eye(3), np.zeros(3), scale)
    sol, *_ = np.linalg.lstsq(np.concatenate((local, np.ones((9, 1))), 1), v, rcond=None)
    return sol[:3, :3].T, sol[3, :3], scale


def box_volume(vertices):
    v = np.asarray(vertices, np.float64)
    return abs(np.linalg.det(np.array([v[2] - v[1], v[3] - v[1], v[5] - v[1]])))


def _polytope_points(src, other):
    """vertices of (src box) ∩ (other box) contributed by `other`: its corners inside src and its edges' crossings of src's
    faces, found in src's local frame where src is the axis-aligned box |x_k| <= s_k / 2"""
    rot, trans, scale = box_fit(src)
    half = scale / 2.0
    local = (np.asarray(other, np.float64) - trans) @ rot          # R^T (p - t), row form
    tol = 1e-6          # plane thickness, the same absolute tolerance objectron's clipping uses (boxes are in metres)
    pts = [p for p in local[1:] if np.all(np.abs(p) <= half + tol)]
    for a, b in _EDGES:
        p, q = local[a], local[b]
        for axis in range(3):
            dpq = q[axis] - p[axis]
            if abs(dpq) < _EPS:
                continue
            for side in (-1.0, 1.0):
                t = (side * half[axis] - p[axis]) / dpq
                if 0.0 <= t <= 1.0:
                    x = p + t * (q - p)
                    o = [k for k in range(3) if k != axis]
                    if abs(x[o[0]]) <= half[o[0]] + tol and abs(x[o[1]]) <= half[o[1]] + tol:
                        pts.append(x)
    if not pts:
        return np.zeros((0, 3))
    return np.asarray(pts) @ rot.T + trans


def box_iou(box1, box2):
    pts = np.concatenate((_polytope_points(box1, box2), _polytope_points(box2, box1)), 0)
    if pts.shape[0] < 4:
        return 0.0
    try:
        inter = ConvexHull(pts).volume
    except Exception:          # degenerate (coplanar) contact: no volume; the reference's call site catches the same way
        return 0.0
    v1, v2 = box_volume(box1), box_volume(box2)
    return inter / (v1 + v2 - inter)


def _rodrigues(axis_angle):
    th = np.linalg.norm(axis_angle)
    if th < 1e-12:
        return np.eye(3)
    k = axis_angle / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def get_best_iou(symmetry_idx, box_vertices, rot_gt, trans_gt, scale_gt):
    """eval_utils.py:134-179 (angle_ratio = 0, so the viewpoint terms never influence the choice): best IoU"""
    rot_gt = np.asarray(rot_gt, np.float64)
    if symmetry_idx != 0:
        return box_iou(box_vertices, box_from_transformation(rot_gt, trans_gt, scale_gt))
    y_axis = rot_gt[:, 1].copy()
    best = 0.0
    for i in range(18):
        rot = _rodrigues(y_axis * (i * 2 * np.pi / 18)) @ rot_gt
        best = max(best, box_iou(box_vertices, box_from_transformation(rot, trans_gt, scale_gt)))
    return best
