"""scp_amd/tester.py -- the test-time step: eval-mode forward, batched pose fitting, pose error.

Mirrors model/tester.py: `define_model` (:76-94), `batch_reshape` (:96-123, same NDC conversion as the trainer),
the loop body of `test` (:178-183: `pred = model(data)`, `pred_fit = pose_fitting(data, pred)`), `pose_fitting`
(:324-427, scp_amd.pose_fit) and the degree / centimetre part of `eval_nocs` (:295-321 with
model/util/eval_utils.py:182-199 get_best_deg_cm) and its 3-D IoU rows (scp_amd.eval_nocs: exact oriented-box IoU and the
18-fold y-symmetry search of get_best_iou, pinned to the reference's values).  Not rebuilt: the cv2 / matplotlib
visualisation and the CUB keypoint-transfer evaluation."""
import numpy as np
import torch

from . import eval_nocs, pose_fit
from .model import MeshNet
from .trainer import Trainer, enable_gemm_tuning, freeze_batchnorm_affine


def get_base_rot(opts, device=None):
    """model/util/base_rot.py:10-18"""
    br = [float(x) for x in opts.base_rot]
    return torch.tensor(br, dtype=torch.float32, device=device).reshape(1, 3, 3)


def get_best_deg_cm(symmetry_idx, box_vertices, box_rotation, rot_gt, trans_gt, scale_gt):
    """eval_utils.py:182-199.  box_vertices [9,3] (centre first, then the 8 corners in objectron order), box_rotation
    [3,3] the fitted rotation; returns (angle error in degrees, translation error in cm)"""
    trans_error = 100 * np.linalg.norm(box_vertices[0] - trans_gt)
    if symmetry_idx == 0:
        # objectron Box.from_transformation: unit-box corners scaled, rotated, translated; vertices[3]-vertices[1]
        # is the box's y edge
        y_gt = rot_gt @ (np.array([0.0, 1.0, 0.0]) * scale_gt)
        y_pred = box_vertices[3] - box_vertices[1]
        angle = np.arccos(y_pred.dot(y_gt) / (np.linalg.norm(y_pred) * np.linalg.norm(y_gt)))
    else:
        r = box_rotation @ rot_gt.transpose()
        angle = np.arccos((np.trace(r) - 1) / 2)
    return angle * 180 / np.pi, trans_error


class Tester:
    deg_cm_thresh = [[5, 2], [5, 5], [10, 2], [10, 5]]                       # tester.py:154
    iou_thresh = [0.25, 0.5]                                                  # tester.py:152

    def __init__(self, opts, prior=None, device=None):
        self.opts = opts
        self.device = torch.device(device if device is not None else "cuda")
        self.prior = prior
        self.deg_cm_result, self.iou_result = [], []

    def define_model(self):
        torch.backends.cudnn.benchmark = True
        if self.device.type == "cuda":
            if self.device.index is not None:
                torch.cuda.set_device(self.device)      # the C-ABI launches use the current device's current stream
            enable_gemm_tuning()
        self.model = MeshNet(self.opts, self.prior)
        if self.opts.model_path:
            self.model.load_network(self.opts.model_path)
        freeze_batchnorm_affine(self.model)                                   # set_bn_eval, tester.py:67-73
        self.model = self.model.to(self.device)
        if self.device.type == "cuda":
            self.model.encoder.backbone.to(memory_format=torch.channels_last)
            self.model.encoder.featnet.to(memory_format=torch.channels_last)
        self.model.eval()
        self.fitter = pose_fit.PoseFitter(self.opts.img_size, get_base_rot(self.opts, self.device))
        return self.model

    batch_reshape = Trainer.batch_reshape

    @torch.no_grad()
    def predict(self, data):
        """one test batch: (pred, pred_fit) as in tester.py:180-183"""
        img, mask, depth, occ, center, length, foc, foc_crop, pp, pp_crop, indices, gt = data
        pred = self.model(data)
        pred_v, faces, tex, imatch, match, match_conf = pred[:6]
        pred_fit = self.fitter.pose_fitting(depth, mask, match, match_conf, foc_crop, pp_crop, pred_v)
        return pred, pred_fit

    def test(self, loader=None, log=print):
        """the loop of tester.py:126-203 for the NOCS-style pose metric: test_loader -> eval forward -> pose_fitting ->
        degree / centimetre hits; returns {"5deg2cm": .., "5deg5cm": .., "10deg2cm": .., "10deg5cm": .., "n": ..} (fractions),
        or only the predictions' count when opts.eval is off.  (3-D IoU rows of eval_nocs: see the module docstring.)"""
        opts = self.opts
        self.define_model()
        if loader is None:
            from .data import test_loader
            loader, self.dataset = test_loader(opts, self.device)
        self.deg_cm_result, self.iou_result, n = [], [], 0
        for i, batch in enumerate(loader):
            self.model.iters = i
            data = self.batch_reshape(batch)
            pred, pred_fit = self.predict(data)
            n += data[0].shape[0]
            if opts.eval:
                self.eval_deg_cm(pred_fit, (batch["rotation"], batch["translation"], batch["scale"]))
        out = {"n": n}
        if opts.eval and self.deg_cm_result:
            hits = np.array(self.deg_cm_result) * 1.0
            for j, (d, c) in enumerate(self.deg_cm_thresh):
                out["%ddeg%dcm" % (d, c)] = hits[:, j].sum() / hits.shape[0]
                log("%2ddeg*%dcm: %.4f" % (d, c, out["%ddeg%dcm" % (d, c)]))
            ious = np.array(self.iou_result) * 1.0
            for j, th in enumerate(self.iou_thresh):
                out["iou@%d" % round(100 * th)] = ious[:, j].sum() / ious.shape[0]
                log("iou@%d: %.4f" % (round(100 * th), out["iou@%d" % round(100 * th)]))
        return out

    def eval_deg_cm(self, pred_fit, gt):
        """tester.py:295-321 without the IoU rows: appends one [5deg2cm, 5deg5cm, 10deg2cm, 10deg5cm] hit list per image"""
        bbox, verts, rotation, translation = pred_fit
        rot_gt, trans_gt, scale_gt = (np.asarray(g.cpu() if torch.is_tensor(g) else g, np.float64) for g in gt)
        bbox, rotation = bbox.cpu().numpy().astype(np.float64), rotation.cpu().numpy().astype(np.float64)
        out = []
        for i in range(bbox.shape[0]):
            # objectron's Box(vertices).rotation is recovered from the vertices; for a box built as (corners * s) R + t in
            # row-vector form that is R^T in its column-vector convention
            ang, cm = get_best_deg_cm(self.opts.symmetry_idx, bbox[i], rotation[i].T, rot_gt[i], trans_gt[i], scale_gt[i])
            hits = [bool(ang < d and cm < c) for d, c in self.deg_cm_thresh]
            self.deg_cm_result.append(hits)
            iou = eval_nocs.get_best_iou(self.opts.symmetry_idx, bbox[i], rot_gt[i], trans_gt[i], scale_gt[i])
            self.iou_result.append([bool(iou >= th) for th in self.iou_thresh])
            out.append((ang, cm))
        return out
