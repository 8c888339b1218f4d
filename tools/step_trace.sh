#!/bin/bash
# tools/step_trace.sh [tag] -- on the GPU box: rocprofv3 kernel trace of a short bench run WITH the ViT look-ahead (the headline
# schedule); kernel stats of the timed window,
# device-idle analysis (tools/profile_summary.py).  Writes gpurun_out/<tag>_{kernel_stats_timed_window.csv,idle_gaps.txt}
tag=${1:-step}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_$tag
rocprofv3 --kernel-trace -d /tmp/trace_$tag --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-isolated > $R/gpurun_out/${tag}_bench_under_rocprof.json 2>/dev/null
f=$(ls /tmp/trace_$tag/*/*kernel_trace.csv | head -1)
cd $R
# window = after the (INIT 3 + warmup 2 + 1)-th optimizer launch (adamw_flat_kernel: one launch per step since round 6)
python tools/profile_summary.py --trace $f adamw_flat_kernel 6 60 > gpurun_out/${tag}_kernel_stats_timed_window.csv
# every kernel of the window (the short ones on the step's serial tail -- clip, optimizer, weight planes -- are below the top 60)
python tools/profile_summary.py --trace $f adamw_flat_kernel 6 1000 > gpurun_out/${tag}_kernel_stats_timed_window_all.csv
python tools/profile_summary.py --gaps $f adamw_flat_kernel 6 30 > gpurun_out/${tag}_idle_gaps.txt
python tools/profile_summary.py --timeline $f adamw_flat_kernel 7 500 > gpurun_out/${tag}_timeline.txt
head -4 gpurun_out/${tag}_kernel_stats_timed_window.csv; head -12 gpurun_out/${tag}_idle_gaps.txt
