#!/bin/bash
# tools/host_enqueue8.sh -- do eight host loops contend?  (VERDICT r3 item 8 i.)  The single-GPU box has the 8-GPU node's host: 128
# cores.  Runs tools/host_enqueue.py once alone, then EIGHT copies at the same time, each pinned to its own 16 cores (what a rank gets
# on the 8-GPU node), all on the one visible GPU -- the device is then 8x oversubscribed, so only the HOST figures mean anything: the
# time to enqueue one step on this process's empty queue (Python, autograd, ctypes, HIP runtime, KFD ioctls, page-table work), which
# is what eight ranks on one host would compete for.  Writes gpurun_out/r04_host_enqueue8.txt.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
out=gpurun_out/r04_host_enqueue8.txt
echo "# tools/host_enqueue8.sh: host time to enqueue one B=32 step on an empty queue (eager launches), $(nproc) host cores" > $out
echo "## one process, unpinned" >> $out
python tools/host_enqueue.py 2>/dev/null | grep -E "empty queue|host enqueue per step" >> $out
echo "## one process pinned to 16 cores" >> $out
taskset -c 0-15 python tools/host_enqueue.py 2>/dev/null | grep -E "empty queue|host enqueue per step" >> $out
echo "## eight processes at once, 16 cores each, one shared GPU (device 8x oversubscribed: host figures only)" >> $out
pids=()
for i in 0 1 2 3 4 5 6 7; do
  lo=$((16 * i)); hi=$((16 * i + 15))
  taskset -c $lo-$hi python tools/host_enqueue.py > /tmp/he8_$i.txt 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for i in 0 1 2 3 4 5 6 7; do echo "rank-like process $i: $(grep 'empty queue' /tmp/he8_$i.txt)" >> $out; done
cat $out
