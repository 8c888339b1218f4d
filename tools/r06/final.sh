#!/bin/bash
# tools/r06/final.sh -- round 6, last GPU call: the driver's exact suite command twice on the final tree, the default bench line (with CPU
# baseline, loss delta and the round's traffic figures), then configs[4] per category
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06w; mkdir -p $O; cd $R
for k in d e; do
    timeout 900 python -m pytest tests/ -x -q -m gpu --durations=12 > $O/suite_$k.txt 2>&1; echo "suite_$k rc=$? $(grep -E ' passed| failed' $O/suite_$k.txt | tail -1)"
done
timeout 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python -c "
import json; d=json.load(open('$O/bench_n1.json')); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'parity_ok', d.get('loss_delta',{}).get('parity_ok'), 'frac', r['frac'], 'traffic/launch', r.get('traffic'), 'cpu', d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)"
bash tools/r06/categories.sh 2>&1 | tail -14
