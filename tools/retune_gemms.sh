#!/bin/bash
# tools/retune_gemms.sh -- regenerate self-corr-pose_amd/tuning/gemm_gfx950.csv on an MI355X: runs the training step in fp32
# and in configs[4] precision with TunableOp tuning on, then merges what the two processes selected (one line per GEMM
# signature; Validator lines kept once).  Output: gpurun_out/gemm_gfx950.csv -- copy it over the shipped file.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SCP_GEMM_TUNING=online SCP_GEMM_TUNING_OUT=$PWD/gpurun_out/tune_f32.csv python bench.py --steps 3 --warmup 6 --no-cpu-baseline > /dev/null
SCP_GEMM_TUNING=online SCP_GEMM_TUNING_OUT=$PWD/gpurun_out/tune_bf16.csv python bench.py --steps 3 --warmup 6 --no-cpu-baseline --mixed-bf16 > /dev/null
SCP_GEMM_TUNING=online SCP_GEMM_TUNING_OUT=$PWD/gpurun_out/tune_test.csv python bench.py --steps 3 --warmup 3 --no-cpu-baseline --workload posefit > /dev/null || true
python3 - <<'PY'
import os
seen, out = set(), []
for name in ("tune_f32.csv", "tune_bf16.csv", "tune_test.csv"):
    path = os.path.join("gpurun_out", name)
    if not os.path.exists(path):
        continue
    for line in open(path):
        key = ",".join(line.split(",")[:2])
        if line.strip() and key not in seen:
            seen.add(key)
            out.append(line if line.endswith("\n") else line + "\n")
out.sort(key=lambda l: (not l.startswith("Validator"),))
open("gpurun_out/gemm_gfx950.csv", "w").writelines(out)
print("merged", len(out), "lines")
PY
