#!/bin/bash
# tools/traffic_run.sh -- on the GPU box: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only) over a short
# bench run, then tools/traffic_report.py -> gpurun_out/r05_traffic.json (copy to profiles/).  Total steps in the run =
# INIT 3 + warmup 1 + 2 headline steps + 2 instrumented steps = 8.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated --no-lookahead > /dev/null 2>&1
done
f=$(ls /tmp/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1); w=$(ls /tmp/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1)
cd $R && mkdir -p gpurun_out && python tools/traffic_report.py $f $w auto B32_S256_V642 gpurun_out/${TRAFFIC_OUT:-r06_traffic.json}
