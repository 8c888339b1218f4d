#!/bin/bash
# tools/r06/categories.sh -- BASELINE configs[4] "all 5 categories" as measured lines: bench.py once per Wild6D
# category preset with ITS shape prior (482-995 vertices) at the headline geometry (B = 32, 256 x 256, fp32) and in configs[4]'s
# geometry + precision (512 x 512, B = 8, mixed bf16), plus the 2562-vertex stress mesh; one JSON line each -> gpurun_out/r06cat/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06cat; mkdir -p $O; cd $R
for c in bottle bowl camera laptop mug; do
    timeout 300 python bench.py --category $c --steps 20 --warmup 5 --no-isolated > $O/cat_${c}_b32_fp32.json 2> $O/cat_${c}_b32_fp32.err
    timeout 300 python bench.py --category $c --high-res --mixed-bf16 --steps 20 --warmup 5 --no-isolated > $O/cat_${c}_512_bf16.json 2> $O/cat_${c}_512_bf16.err
done
timeout 300 python bench.py --high-res --mixed-bf16 --steps 20 --warmup 5 --no-isolated > $O/stress2562_512_bf16.json 2> $O/stress2562_512_bf16.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06cat/*.json"))):
    try:
        d = json.load(open(f)); print("%-32s %7.2f it/s %7.2f ms  %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["metric"][:90]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
