"""tests/step_report.py -- prints every loss / pose / gradient deviation of the golden training step
on the current device (no asserts); used to read parity numbers off the GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import step_case  # noqa: E402

dev = "cuda" if torch.cuda.is_available() else "cpu"
if dev == "cpu":
    import oracle_backend

    class _MP:
        def setattr(self, o, n, v):
            setattr(o, n, v)
    oracle_backend.install(_MP())
model, data, d = step_case.build(dev)
if "--free-nn" in sys.argv:
    model.pretrain_corr_net.nn_override = None
total, aux = model(data)
total.mean().backward()
for k, v in aux.items():
    ref = float(d["aux_" + k])
    print("%-24s %.9g  ref %.9g  rel %.2e" % (k, float(v), ref, abs(float(v) - ref) / max(abs(ref), 1e-12)))
rot, trans = model.last_pose
print("rotation max abs diff %.3e   translation max abs diff %.3e" % (
    np.abs(rot.cpu().numpy() - d["rotation"]).max(), np.abs(trans.cpu().numpy() - d["translation"]).max()))
params = dict(model.named_parameters())
for key, pname in (("grad_mean_v", "mesh.mean_v"), ("grad_resnet_conv1", "encoder.backbone.resnet.conv1.weight"),
                   ("grad_featnet_proj", "encoder.featnet.proj.weight"),
                   ("grad_pose_trans", "encoder.pose_predictor.trans_pred_layer.weight"),
                   ("grad_shapenerf_fc_rgb", "encoder.shape_predictor.shapenerf.fc_rgb.weight"),
                   ("grad_mesh_stn_fc", "encoder.featnet_mesh.stn.fc.weight")):
    g = params[pname].grad.detach().double().cpu().numpy().ravel()
    r = d[key].astype(np.float64).ravel()
    print("%-24s rel L2 %.3e  cos %.8f" % (key, np.linalg.norm(g - r) / np.linalg.norm(r),
                                           g @ r / np.linalg.norm(g) / np.linalg.norm(r)))
bw, fw = model.pretrain_corr_net.last_nn
for name, got in (("bw", bw), ("fw", fw)):
    ref = d["nn_" + name].astype(np.int64)
    gap, top = d["nn_%s_gap" % name], d["nn_%s_top" % name]
    flips = (got.cpu().numpy() != ref) & (top > -1e4)
    print("mutual-NN %s: %d flips of %d live, max rel gap at a flip %.3e" % (
        name, flips.sum(), (top > -1e4).sum(), (gap[flips] / np.abs(top[flips])).max() if flips.any() else 0.0))
