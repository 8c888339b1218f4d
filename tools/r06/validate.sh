#!/bin/bash
# tools/r06/validate.sh -- round 6 closing run on the GPU box: the driver's exact suite command (twice more), smoke(), the default bench line
# with CPU baseline + loss delta, `python bench.py --gpus 2` launching its own ranks (gloo, both ranks on the one device), kernel trace of
# the step, HBM traffic (two --pmc passes)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06v; mkdir -p $O; cd $R
for k in b c; do
    timeout 900 python -m pytest tests/ -x -q -m gpu --durations=12 > $O/suite_$k.txt 2>&1; echo "suite_$k rc=$? $(grep -E 'passed|failed' $O/suite_$k.txt | tail -1)"
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python -c "
import json; d=json.load(open('$O/bench_n1.json')); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'parity_ok', d.get('loss_delta',{}).get('parity_ok'), 'frac', r['frac'], 'traffic', r.get('traffic'), 'cpu', d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)"
SCP_SINGLE_DEVICE=1 SCP_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline --no-isolated > $O/bench_n2_gloo_one_device.json 2> $O/bench_n2.err
python -c "
import json; d=json.load(open('$O/bench_n2_gloo_one_device.json')); print('n2 launch path:', d['n_gpus'], d['value'], d['config']['buckets_launched_inside_backward'], d['config']['rccl_ranks'])" || tail -5 $O/bench_n2.err
timeout 300 bash tools/step_trace.sh r06v/trace > $O/trace_stdout.txt 2>&1; tail -4 $O/trace_stdout.txt | cut -c1-200
TRAFFIC_OUT=r06v/r06_traffic.json timeout 900 bash tools/traffic_run.sh > $O/traffic_stdout.txt 2> $O/traffic_stderr.txt; tail -2 $O/traffic_stderr.txt
python -c "
import json; t=json.load(open('$O/r06_traffic.json'))
for k in ('vit_gemm','vit_attention','fvm_forward','fvm_backward','raster_forward','raster_backward','conv_igemm'): print(k, round(t[k]['bytes']/1e6,1), 'MB/step') if k in t else None"
