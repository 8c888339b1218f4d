#!/bin/bash
# tools/graph_timeline.sh [tag] -- on the GPU box: the training step captured in a HIP graph (tools/graph_probe.py) and replayed under
# rocprofv3 --kernel-trace.  Replay takes the host out of the picture (a traced eager step is host-bound: 50 ms instead of 40), so
# the per-family residency timeline (tools/profile_summary.py --timeline) shows what the DEVICE does with the step.
tag=${1:-graph}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_$tag
rocprofv3 --kernel-trace -d /tmp/trace_$tag --output-format csv -- python $R/tools/graph_probe.py > $R/gpurun_out/${tag}_probe.log 2>&1
f=$(ls /tmp/trace_$tag/*/*kernel_trace.csv | head -1)
cd $R
tail -4 gpurun_out/${tag}_probe.log
python tools/profile_summary.py --timeline $f multi_tensor_apply_kernel 40 250 > gpurun_out/${tag}_timeline.txt
python tools/profile_summary.py --chain $f multi_tensor_apply_kernel 40 > gpurun_out/${tag}_middle_chain.txt
python tools/profile_summary.py --trace $f multi_tensor_apply_kernel 150 60 > gpurun_out/${tag}_kernel_stats_replay_window.csv
head -3 gpurun_out/${tag}_kernel_stats_replay_window.csv
