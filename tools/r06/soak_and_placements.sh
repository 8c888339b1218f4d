#!/bin/bash
# tools/r06/soak_and_placements.sh -- final evidence on HEAD's defaults: 300-step soak (memory flat, losses finite, step time stable) and the
# default schedule at 12 stream placements x 2 fresh processes (tools/r06/hang_repro.py), single- and multi-GPU schedule
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06s; mkdir -p $O; cd $R
timeout 300 python tools/soak.py > $O/soak.txt 2>&1; tail -4 $O/soak.txt | cut -c1-200
export SCP_DEVICE_TIMEOUT_S=10 SCP_REPRO_ITERS=8
ok=0; hang=0
for p in 0 1 2 3 4 5 6 7 33 34 35 36; do for k in 1 2; do
    line=$(timeout 100 python tools/r06/hang_repro.py $p steps final 2>/dev/null | tail -1); case "$line" in OK*) ok=$((ok+1));; *) hang=$((hang+1)); echo "$line" | cut -c1-300;; esac
done; done
echo "default schedule over 12 placements x 2: ok $ok, not ok $hang" | tee $O/placements.txt
ok=0; hang=0
for p in 0 2 3 35; do for k in 1 2; do
    line=$(SCP_FORCE_COLLECTIVES=1 timeout 100 python tools/r06/hang_repro.py $p steps final_dist 2>/dev/null | tail -1); case "$line" in OK*) ok=$((ok+1));; *) hang=$((hang+1)); echo "$line" | cut -c1-300;; esac
done; done
echo "default schedule + communication stream (1-rank RCCL) over 4 placements x 2: ok $ok, not ok $hang" | tee -a $O/placements.txt
