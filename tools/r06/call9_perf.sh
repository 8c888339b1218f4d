#!/bin/bash
# tools/r06/call9_perf.sh -- round 6: what the kernel changes of the round are worth (pair-queue forward rasteriser, XCD-aware a7 kernels,
# FlatAdamW default): isolated kernel times A/B, the step time A/B, then the default bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R
timeout 200 python tools/softras_microbench.py > $O/softras_pair_queue.txt 2>&1; tail -12 $O/softras_pair_queue.txt | cut -c1-200
SCP_RASTER_FWD=legacy timeout 200 python tools/softras_microbench.py > $O/softras_per_face.txt 2>&1; tail -12 $O/softras_per_face.txt | cut -c1-200
timeout 200 python tools/fvm_bench.py > $O/fvm_bench.txt 2>&1; tail -3 $O/fvm_bench.txt | cut -c1-250
for tag in default three_side_streams legacy_raster torch_adamw; do
    case $tag in default) e="SCP_DUMMY=1";; three_side_streams) e="SCP_TEXTURE_STREAM=1";; legacy_raster) e="SCP_RASTER_FWD=legacy";; torch_adamw) e="SCP_ADAMW=torch";; esac
    env $e timeout 300 python bench.py --no-cpu-baseline --no-isolated --steps 40 --warmup 10 > $O/bench_$tag.json 2> $O/bench_$tag.err
    python -c "
import json; d=json.load(open('$O/bench_$tag.json')); r=d['roofline']; o=r['others']
print('$tag', round(d['value'],2), 'it/s', round(d['ms_per_step'],2), 'ms; gemm frac', round(r['frac'],3), '; raster fwd ms', o['raster_forward']['avg_launch_ms'] if 'raster_forward' in o else None, '; fvm bwd ms', o.get('fvm_backward',{}).get('avg_launch_ms'))" 2>&1 | tail -1
done
timeout 900 python -m pytest tests/ -x -q -m gpu --durations=45 > $O/suite_a.txt 2>&1; echo "suite_a rc=$?"; tail -60 $O/suite_a.txt | cut -c1-200
