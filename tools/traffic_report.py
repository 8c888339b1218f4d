"""tools/traffic_report.py -- turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --steps K` (one counter per
pass, kernel-trace only) into profiles/r05_traffic.json: HBM bytes PER STEP of the hand-written kernel families, the number
bench.py reports as roofline.traffic.  FETCH_SIZE is reported in KB and, on gfx950, counts half of the bytes of wide coalesced
streams (MI355X_MICROARCH.md): bytes = 2 * FETCH_SIZE_KB * 1024 + WRITE_SIZE_KB * 1024.
usage: python tools/traffic_report.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps_in_run> <size tag> <out.json>"""
import collections
import csv
import json
import sys

FAMILIES = {"vit_gemm": "vit_gemm_kernel", "vit_attention": "vit_attention_", "vit_qkv_split": "qkv_split_kernel", "conv_splitk_fold": "conv_splitk_fold_kernel", "raster_backward": "raster_backward_kernel<1, 1>",
            "raster_forward": "raster_forward_kernel", "corr_fused": "fvm_", "conv_igemm": "conv_igemm_kernel",
            "conv_wgrad": "conv_wgrad_kernel", "wgrad_fold": "wgrad_fold_kernel", "mutual_nn_fused": "mutual_nn_fused_kernel",
            # round 5: the keys bench.py's roofline.others looks up (per C-ABI call; one call of each per step)
            "raster_forward_softtex": "raster_forward_kernel<1, 1, false>", "fvm_forward": "fvm_forward_kernel",
            "fvm_backward": "fvm_backward_", "project_vertices": "project_", "gradclip": "gradclip_"}


def totals(path, counter):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            for fam, pat in FAMILIES.items():
                if pat in r["Kernel_Name"]:
                    agg[fam] += float(r["Counter_Value"])
    return agg


fetch, write = totals(sys.argv[1], "FETCH_SIZE"), totals(sys.argv[2], "WRITE_SIZE")
steps, tag = float(sys.argv[3]), sys.argv[4]
out = {}
for fam in FAMILIES:
    if fam in fetch or fam in write:
        out[fam] = {"bytes": (2 * fetch.get(fam, 0.0) + write.get(fam, 0.0)) * 1024.0 / steps, "size": tag,
                    "fetch_kb_per_step": fetch.get(fam, 0.0) / steps, "write_kb_per_step": write.get(fam, 0.0) / steps,
                    "note": "per training step; FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE"}
json.dump(out, open(sys.argv[5], "w"), indent=1)
print(json.dumps(out, indent=1))
