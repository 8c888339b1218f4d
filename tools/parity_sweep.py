"""tools/parity_sweep.py -- which matrix-core switch moves which loss term?  

For several synthetic B=32 batches: the first training step's forward on the CPU oracle backend (bench.cpu_baseline's step), then
on the GPU with every combination of interest of the four switches
    SCP_CONV_GEMM (encoder convolutions forward / input gradient)    SCP_VIT_GEMM (ViT linears)    SCP_VIT_ATTN (ViT attention)
set to the fp32 matrix cores or to the split main loop, through bench.loss_delta (pinned and free-running legs; the CPU side's
discrete selections injected, so the ViT switches can only show up in the mutual-NN flip fraction).  The weight-gradient switch
(SCP_CONV_WGRAD) cannot move a forward loss and is not swept.  Prints one table per leg: rows = configuration, columns = loss terms,
entries = max over the batches of |gpu - cpu| / |cpu|.

    python tools/parity_sweep.py [n_seeds] > profiles/r04_parity_sweep.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402
from scp_amd import dino, fused_conv  # noqa: E402

CONFIGS = [("all fp32 cores", "fp32", "fp32", "fp32"), ("conv split only", "split", "fp32", "fp32"),
           ("ViT GEMM split only", "fp32", "split", "fp32"), ("ViT attention split only", "fp32", "fp32", "split"),
           ("all split (shipped default)", "split", "split", "split")]
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rows = {leg: {name: {} for name, *_ in CONFIGS} for leg in ("pinned", "pinned_geometry_only", "free_running")}
extra = {name: {"flips": 0.0, "pred_v": 0.0, "rotation": 0.0, "translation": 0.0} for name, *_ in CONFIGS}
for seed in range(n_seeds):
    _, ref = bench.cpu_baseline(batch_seed=100 + seed)
    for name, conv, gemm, attn in CONFIGS:
        fused_conv.CONV_MODE, dino.GEMM_MODE, dino.ATTN_MODE = conv, gemm, attn
        out = bench.loss_delta(ref, "cuda:0", batch_seed=100 + seed)
        for leg in rows:
            src = out["pinned"]["pinned_geometry_only"] if leg == "pinned_geometry_only" else out[leg]
            for k, v in src["rel"].items():
                rows[leg][name][k] = max(rows[leg][name].get(k, 0.0), v)
        e = extra[name]
        e["flips"] = max(e["flips"], out["mutual_nn_flip_fraction_before_injection"])
        for k in ("pred_v", "rotation", "translation"):
            e[k] = max(e[k], out["free_running"]["encoder_deviation_max_abs"][k])
    print("seed %d done" % (100 + seed), file=sys.stderr, flush=True)
fused_conv.CONV_MODE = dino.GEMM_MODE = dino.ATTN_MODE = "split"
band = bench.reference_band() or {}
terms = sorted(next(iter(rows["pinned"].values())))
for leg in ("pinned", "pinned_geometry_only", "free_running"):
    print("\n%s leg: max over %d batches (seeds 100..%d) of |gpu - cpu| / |cpu| per loss term, B=32, 642v/1280f" % (leg, n_seeds, 99 + n_seeds))
    print("%-30s" % "configuration" + "".join("%12s" % t.replace("_loss", "")[:11] for t in terms))
    for name, *_ in CONFIGS:
        print("%-30s" % name + "".join("%12.2e" % rows[leg][name][t] for t in terms))
    if leg == "free_running" and band:
        print("%-30s" % "reference's own band (B=32)" + "".join("%12.2e" % band.get(t, 1e-4) for t in terms))
print("\nencoder outputs, GPU vs CPU (max abs over the batches), and the GPU's own mutual-NN selections that differ from the CPU's")
print("%-30s%12s%12s%12s%12s" % ("configuration", "pred_v", "rotation", "translation", "nn flips"))
for name, *_ in CONFIGS:
    e = extra[name]
    print("%-30s%12.2e%12.2e%12.2e%12.2e" % (name, e["pred_v"], e["rotation"], e["translation"], e["flips"]))
