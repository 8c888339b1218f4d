set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05; cd $R
timeout 1200 python -m pytest tests/test_project.py tests/test_render_golden.py tests/test_step_gpu.py tests/test_losses_golden.py tests/test_fused_losses.py -q -m gpu > gpurun_out/r05/pytest_project.txt 2>&1
tail -8 gpurun_out/r05/pytest_project.txt
SCP_STREAMS=serial timeout 600 python tools/op_census.py > gpurun_out/r05/op_census2.txt 2>&1
grep -A22 'launches by stage' gpurun_out/r05/op_census2.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r05/bench_proj.json 2>gpurun_out/r05/bench_proj.err
SCP_STREAMS=serial timeout 600 python bench.py --no-cpu-baseline --no-isolated > gpurun_out/r05/bench_proj_serial.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_proj","bench_proj_serial"):
    l=[x for x in open("gpurun_out/r05/%s.json"%f) if x.startswith("{")]
    d=json.loads(l[-1]); print(f, d["value"], d["ms_per_step"], d["config"]["vit_lookahead"]["unpipelined_ms_per_step"])
    for k,v in d["roofline"].get("others",{}).items(): print("   ",k, v.get("avg_launch_ms"), v.get("frac"), (v.get("valu") or {}).get("frac"))
PY
